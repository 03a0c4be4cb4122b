// lsm2d_capi.hip -- host side of the C ABI declared in include/lsm2d.h (liblsm2d_hip.so).
// Owns the device buffers, packs kernel arguments, launches the gfx950 kernels of lsm2d_kernels.h on
// the context's HIP stream.  No CPU fallback: without a usable HIP device every entry point fails.
#include "lsm2d.h"

namespace lsm2d { static constexpr int LSM2D_RUNNING = -99; }
using lsm2d::LSM2D_RUNNING;
#include "lsm2d_kernels.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <new>
#include <string>
#include <chrono>
#include <vector>
#include <algorithm>
#include <thread>
#include <atomic>

using namespace lsm2d;

static thread_local std::string g_last_error;

struct lsm2d_context {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t last_ev0 = nullptr, last_ev1 = nullptr;      // the events lsm2d_last_kernel_ms reads: the current lane's, or those of the batch most recently waited for
  bool have_timing = false;
  std::string last_error;
  // pinned host staging + device scratch, grown on demand
  void* h_stage = nullptr; size_t h_stage_bytes = 0; void* h_stage_dev = nullptr;
  void* d_scratch = nullptr; size_t d_scratch_bytes = 0;
  void* d_split = nullptr; size_t d_split_bytes = 0;      // workspace of the split aligner path
  void* d_kd_work = nullptr; size_t d_kd_work_bytes = 0;  // working set of the KD-tree build (ping-pong copies, queues, counters): kept, grown on demand
  void* h_flag = nullptr;                                 // 256 pinned bytes of its own for small read-backs inside a call (the KD-tree build's level counts)
  struct BeamDirs { int n_beams; float angle_min, angle_max; float2* d_dir; };
  std::vector<BeamDirs> beam_dirs;                        // (cos, sin) per beam of the sensors seen so far (lsm2d_preprocess_scan_into)
  int max_dyn_lds = 0;
  unsigned long long sync_epoch = 1;   // bumped by every stream_sync()
  int align_path = 0;          // 0 auto, 1 fused, 2 split, 3 slice pair
  int kernel_timing = 0;       // record HIP events around the hot-path launches (lsm2d_last_kernel_ms).  Off by default: two timed events per
                               // operation cost the live tracker 30 us of its 165 us step (they are API calls AND pipeline drains)
  int last_align_path = 0;     // what the most recent lsm2d_align_batch used (1, 2 or 3)
  int zero_copy_max = 256;     // largest batch whose arguments and results travel through pinned host memory directly ("zero_copy_max" option, A/B knob)
  int find_path = 0;           // 0 auto (point-query finder calls with more queries than one trip of a workgroup: many workgroups), 1 one workgroup always
  int grid_big_threshold = 16384;   // clouds of at least this many points get their search grid built by the chip-wide kernels (k_grid_big_*)
  int distmap_build = 0;       // 0 auto (scatter build when it packs), 1 gather build always (the two agree bit for bit: tests)
  int balance = 1;             // culled batches of more than 256 alignments: place them on the chip by estimated work (k_cull_estimate / balance_order); 0: workgroup b = alignment b
  int n_cu = 0;                // compute units of the device (hipDeviceProp_t.multiProcessorCount)
  int cull_est_um = 0, cull_est_urad = 40000;      // margins of the work estimate's chunk test ("cull_est_um", "cull_est_urad"; placement only)
  int results_to_host = 1;     // batches that travel by copies: poses, information matrices, statuses, iteration counts and clock stamps written straight to pinned host memory (0: to the device and copied; A/B knob)
  int two_stage = 0;           // 1: ... in TWO launches: iteration 0 first (k_first_iteration), the rest placed by the length of iteration 1's unit lists.  Measured on configs[1]: the second
                               // launch 0.698 ms with a tail of 7 % instead of 10, but the first costs 95 us (every workgroup in the same phase at the same time: nothing overlaps) and the ordering 16:
                               // 0.861 vs 0.836 ms per step.  Off; kept as an A/B knob with its bit-identity test
  int balance_notes = 1;       // ... group the workgroup ids by the CU the previous launch of the same shape ran them on (0: assume b, b + n_cu, ...; A/B knob)
  int32_t* d_wg_place = nullptr; unsigned long long wg_place_shape = 0;      // the notes (one int per workgroup) and the launch shape they belong to
  uint32_t* d_xcd = nullptr; size_t d_xcd_bytes = 0;                         // the XCD window's counters (AlignArgs::xcd_sync), cleared per launch
  int32_t* d_order = nullptr;                                                // [4096] the placement the latest estimate made: kept for the next run of the SAME batch (order_valid / order_key / order_poses)
  int proj_modes = 1;          // projective batches against map-sized clouds: the instantiation with the culled stream only (0: the shared one; A/B knob)
  int kd_modes = 1;            // KD-tree batches: the instantiations with one form of the descent only (0: the shared one; A/B knob)
  int nn_lds_only = 1;         // grid NN with every alignment's tables staged in LDS: the instantiation without the search in global memory (0: the shared one; A/B knob)
  int nn_qcache = 1;           // grid NN over a map-sized fixed cloud: cache every query's cell ranges in LDS between iterations (0: off; A/B knob)
  int cull_block = 0;          // steps per unit of the culled stream (0: automatic, ~1/25 of a chunk; even; tuning knob)
  int sum_order = 0;           // 0: H, b and the chi^2 sums are formed in trees (a thread's pairs, 64 lanes, 8 waves: the fast order); 1: pair after pair in the reference's order
                               // (nicp_post.m:69-90: ascending column / moving index) -- bitwise the sequential fp32 CPU restatement the tests check against; k_align_seq, k_split_finish<true>, k_linearize_seq
  int cull = 1;                // k_align, projective slices: exact culling of the moving cloud against the fixed canvas (0: off; results do not depend on it)
  int cull_keep = 1;           // ... the culled stream's unit lists are kept across iterations while the estimate stays within the margins they were built with (0: rebuilt every iteration; A/B knob)
  int cull_margin_um = 10000;  // the translation margin in micrometres (10 mm) and
  int cull_margin_urad = 2000; // the rotation margin in microradians (2 mrad): tuning knobs, results do not depend on them
  int kd_wg_max_points = 16384; // KD-tree build: clouds of at most this many points are built by ONE launch, a workgroup per cloud walking the levels itself (k_kd_build_wg); 0: the level loop for all (A/B knob; same trees)
  int kd_wide_min_points = 1024;   // KD-tree build of larger clouds: levels whose evenly split nodes would hold at least this many points run a workgroup per node (0: a wave per node always; A/B knob)
  int grid_big_cells_x10 = 50;     // exact NN grid over a map-sized cloud: cells per side = this / 10 x sqrt(points); role B / NN on configs[1]: 2.0 2.08 ms, 3.0 1.85, 4.0 1.67, 5.0 1.66, 6.0 (rounds 2-3) 1.72, 8.0 1.73 (tuning knob)
  int kd_scan_max_clouds = 8;      // KD-tree build of a set of at most this many clouds of <= 1280 points each: the latency form with the working set in LDS (0: never; A/B knob)
  int kd_chain = 1;            // KD-tree build: how a node's sequential sums run -- 1 systolic DPP pass (default), 0 one v_readlane + add per value (same bits: tests)
  int kd_lds_nodes = 1536;     // KD-tree finder inside k_align: nodes of the fixed cloud's tree staged in LDS (0: none; results do not depend on it; 512 / 1024 / 1536: 0.830 / 0.804 / 0.778 ms on configs[1] role B)
  int clock_stride = 0;               // 0: ~32 stamped workgroups per launch; > 0: every clock_stride-th ("clock_stride" option, diagnostics)
  long long last_clock_khz = 0;       // in-kernel clock of the most recent timed k_align launch (median over the stamped workgroups), 0 = none
  long long last_wg_lifetime_ns = 0;  // median lifetime of its stamped workgroups
  int xcd_lockstep = 0;        // big-map batches of one dispatch round: the workgroups of an XCD walk the map in step, pass by pass ("xcd_lockstep": 0 free-running, 1 nobody starts
                               // a pass before everybody on its XCD has finished the previous one, 2 .. : the one before that, ...)
  int uploads = 0;             // host-to-device cloud uploads queued so far ("uploads", read-only: the adapters' upload-once test reads it)
  long long last_h2d_bytes = 0; // bytes the most recent cloud upload moved over the host link
  int experiments =
#ifdef LSM2D_EXPERIMENTS
      1;
#else
      0;
#endif
  int lane_streams = 1;        // asynchronously begun batches launch on their lane's own stream (lane_stream); experiments build: 0 = in order on the context's stream, as first built
  int estimate_reuse = 1;      // a prepared batch run again with unchanged start poses keeps its placement (no k_cull_estimate launch); experiments build: 0 switches that off
  int last_xcd_lockstep = 0;   // what the latest aligner call ran with (0: free-running)
  int last_cull_estimate = 0;  // what the latest aligner call did about the placement's estimate ("last_cull_estimate")
  bool order_valid = false; unsigned long long order_key = 0; std::vector<float> order_poses;      // the placement d_order holds: which batch it was made for
  int last_query_cull = 0;     // the latest aligner call ran its point-query finder with the exact culling of the queries (k_align, tiles of 64 moving points)
  long long last_kd_levels = 0, last_kd_nodes = 0;      // shape of the most recently built KD-tree set (levels of the deepest tree, nodes in all of them)
  std::vector<lsm2d_cloudset*> live_sets;      // lsm2d_destroy orphans what is left (a set destroyed after its context must not touch it)
  // ---- batches in flight (lsm2d_align_batch_begin / _wait, round 5).  Everything a batch keeps between its launch and its results -- the pinned staging buffer its
  // results land in, the device scratch its arguments and statistics live in, the placement the estimate made for it, its timing events -- forms a LANE; the
  // context has two: the members above (h_stage, d_scratch, d_order, order_*, ev0, ev1) are the CURRENT lane's, `parked` holds the other one's.  begin() works
  // on the current lane, marks it busy and swaps: whatever is called next (the refill and the begin of the FOLLOWING batch) finds a free lane; wait() frees
  // the lane its batch was begun on.  At most two batches are in flight.
  struct Lane {
    void* h_stage = nullptr; size_t h_stage_bytes = 0; void* h_stage_dev = nullptr; void* d_scratch = nullptr; size_t d_scratch_bytes = 0;
    int32_t* d_order = nullptr; bool order_valid = false; unsigned long long order_key = 0; std::vector<float> order_poses;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr; bool busy = false; int id = 1;
    std::vector<unsigned char> inputs_shadow; bool inputs_valid = false;
  } parked;
  int lane_id = 0; bool lane_busy = false; hipEvent_t ev_done = nullptr;      // the current lane's id / busy flag / "this batch's last operation has run"
  // what the lane's device scratch holds as a batch's INPUT block (start poses, index arrays), byte for byte, while nothing else has used the scratch since: a batch
  // that comes again with the same inputs needs no upload either (ensure_scratch invalidates it: every other user of the scratch comes through there)
  std::vector<unsigned char> inputs_shadow; bool inputs_valid = false;
  int inflight = 0;                        // batches begun and not yet waited for
  // the SECOND stream: while a batch is in flight, what the next one needs ahead of its k_align -- its scans' preprocessing (lsm2d_preprocess_scans_refill), its
  // start poses' upload, its placement's estimate -- is queued here, so the chip runs it in the slots the launch in flight leaves free (its tail), and k_align
  // on the first stream waits for an event behind it
  hipStream_t stream_b = nullptr; hipEvent_t ev_b = nullptr, ev_a_est = nullptr; bool b_dirty = false, a_est_recorded = false, b_recorded = false;
  hipStream_t stream_c = nullptr; hipEvent_t ev_c = nullptr; bool c_dirty = false;      // lsm2d_preprocess_scans_refill while a batch is in flight: a stream of its own (refill_stream)
  hipStream_t stream_h = nullptr; hipEvent_t ev_h = nullptr;                            // ... and one for its host-to-device copy (the copy engine's; nothing it waits for)
  // a batch begun asynchronously launches on ITS LANE's stream: two batches in flight are two streams, and the second one's workgroups fill the slots the first
  // one's tail leaves free instead of waiting for its last workgroup (lane_stream)
  hipStream_t k_stream[2] = {nullptr, nullptr}; hipEvent_t ev_main = nullptr;
};
static void swap_lanes(lsm2d_context* c) {
  lsm2d_context::Lane& p = c->parked;
  std::swap(c->h_stage, p.h_stage); std::swap(c->h_stage_bytes, p.h_stage_bytes); std::swap(c->h_stage_dev, p.h_stage_dev);
  std::swap(c->d_scratch, p.d_scratch); std::swap(c->d_scratch_bytes, p.d_scratch_bytes);
  std::swap(c->d_order, p.d_order); std::swap(c->order_valid, p.order_valid); std::swap(c->order_key, p.order_key); c->order_poses.swap(p.order_poses);
  std::swap(c->ev0, p.ev0); std::swap(c->ev1, p.ev1); std::swap(c->ev_done, p.ev_done); std::swap(c->lane_busy, p.busy); std::swap(c->lane_id, p.id);
  c->inputs_shadow.swap(p.inputs_shadow); std::swap(c->inputs_valid, p.inputs_valid);
}
// Side streams, created on first use with the highest priority the device has: what they carry is short, and the launch in flight holds every wave slot of the
// chip -- the slots that come free at its end should go to the next batches' pre-kernels first, not to the 1000 long-lived workgroups of the launch queued behind it.
static bool make_side_stream(hipStream_t* st, hipEvent_t* ev) {
  int prio_least = 0, prio_greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void) hipGetLastError(); prio_greatest = 0; }
  if (hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_greatest) != hipSuccess) { (void) hipGetLastError(); *st = nullptr; return false; }
  if (hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); (void) hipStreamDestroy(*st); *st = nullptr; *ev = nullptr; return false; }
  return true;
}
// the stream a batch's PRE-kernels go to (its start poses, its estimate): the second one while another batch is in flight, else the context's own
static hipStream_t pre_stream(lsm2d_context* ctx) {
  if (ctx->inflight <= 0) return ctx->stream;
  if (!ctx->stream_b && !make_side_stream(&ctx->stream_b, &ctx->ev_b)) return ctx->stream;
  ctx->b_dirty = true;
  return ctx->stream_b;
}
// ... and the stream lsm2d_preprocess_scans_refill goes to while a batch is in flight: a THIRD one.  On the second stream a refill queued a step ahead of its
// batch would sit between two estimates, and every estimate behind a preprocessing launch that the launch in flight starves of wave slots: the chain
// preprocessing -> estimate -> k_align ran in the gap between two launches however early the host queued it (rocprofv3 trace).
static hipStream_t refill_stream(lsm2d_context* ctx) {
  if (ctx->inflight <= 0) return ctx->stream;
  if (!ctx->stream_c && !make_side_stream(&ctx->stream_c, &ctx->ev_c)) return ctx->stream;
  return ctx->stream_c;
}
// The refill's host-to-device copy depends on nothing the device does (the batch that read the set's previous contents has been waited for: the caller's side of
// the contract), but in order on the refill stream it sat BETWEEN two preprocessing launches: 82 us of copy after the previous launch had ended, the next one
// starting just as the next k_align took every wave slot -- a steady state in which every preprocessing launch finished after the launch it was meant to hide
// under (trace in DESIGN.md section 5).  On a stream of its own the copy runs when the host queues it.
// (Only for a refill queued a step AHEAD -- two batches in flight, the order lsm2d.h recommends.  With one in flight the refill belongs to the very next begin():
// there the early copy only moves the preprocessing launch into the start of the running k_align, and a fast host falls into a rhythm of one overlapped and one
// fully serial step -- 0.81 against 0.75 ms per step from C++, stream_ab_r05.txt.)
static hipStream_t refill_copy_stream(lsm2d_context* ctx, hipStream_t refill) {
  if (refill == ctx->stream || ctx->inflight < 2) return refill;
  if (!ctx->stream_h && !make_side_stream(&ctx->stream_h, &ctx->ev_h)) return refill;
  return ctx->stream_h;
}
// The stream an asynchronously begun batch's OWN operations go to (its memsets, k_align, its results' copies, its events): one per lane.  In order on the
// context's stream the younger batch's 1000 workgroups waited for the older launch's LAST workgroup while a tenth of the chip's slot-time stood empty in its tail
// (the mean workgroup ends 10 % before its launch); on two streams they start as slots come free.  The context's own stream keeps the synchronous calls and
// everything that prepares sets; an event recorded on it at begin() orders the lane's stream (and the pre-kernels' stream) behind what it holds.
static hipStream_t lane_stream(lsm2d_context* ctx) {
  if (!ctx->lane_streams) return ctx->stream;
  hipStream_t& st = ctx->k_stream[ctx->lane_id & 1];
  if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void) hipGetLastError(); st = nullptr; return ctx->stream; }
  if (!ctx->ev_main && hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); ctx->ev_main = nullptr; return ctx->stream; }
  return st;
}
// what was queued on the refill stream comes before whatever `st` (and the context's own stream) is given next
static hipError_t join_refill_stream(lsm2d_context* ctx, hipStream_t st) {
  if (!ctx->c_dirty || !ctx->stream_c) return hipSuccess;
  hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->ev_c, 0);
  if (e == hipSuccess && st != ctx->stream) e = hipStreamWaitEvent(st, ctx->ev_c, 0);
  ctx->c_dirty = false;
  return e;
}
// k_align (first stream) must see what the second stream was given for it
static hipError_t join_pre_stream(lsm2d_context* ctx, hipStream_t ks) {
  if (!ctx->b_dirty || !ctx->stream_b) return hipSuccess;
  hipError_t e = hipEventRecord(ctx->ev_b, ctx->stream_b);
  if (e == hipSuccess) { ctx->b_recorded = true; e = hipStreamWaitEvent(ks, ctx->ev_b, 0); }
  ctx->b_dirty = false;
  return e;
}

// every wait for the context's stream goes through here: the epoch lets a set know that a transfer it queued from its pinned
// staging buffer has certainly run (some wait happened since) without an event of its own
static hipError_t stream_sync(lsm2d_context* ctx) {
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  ++ctx->sync_epoch;
  return e;
}

// The aligner kernels write their results straight to pinned host memory when a call carries few alignments, each alignment's
// status word last and behind a system-scope release.  Polling those words gets the pose to the caller a few microseconds earlier
// than waking up from hipStreamSynchronize -- the live tracker's next launches (the merger) are waiting for exactly that.  The
// stream is in-order, so once the last alignment has reported, everything queued before the launch has run as well: the epoch
// moves as for a stream wait.  A launch that does not report within the spin budget falls back to the real wait (which also
// surfaces a device error).
static constexpr int32_t kStatusNotWritten = -1;
// (done != nullptr: the batch was begun asynchronously and launched on its lane's stream -- the fallback waits for the event recorded behind THAT launch;
// the context's own stream may be idle while the kernel is still writing: round-5 advisor)
static hipError_t wait_for_statuses(lsm2d_context* ctx, const int32_t* st, int n, hipEvent_t done = nullptr) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) {
    unsigned spins = 0;
    while (__atomic_load_n(&st[i], __ATOMIC_ACQUIRE) == kStatusNotWritten) {
      __builtin_ia32_pause();
      if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return done ? hipEventSynchronize(done) : stream_sync(ctx);
    }
  }
  ++ctx->sync_epoch;
  return hipSuccess;
}

struct GridCache {     // one search grid per (cloud set, max_distance), built on first use
  float max_distance = 0.0f;
  void* d_block = nullptr;      // ONE allocation (hipMalloc costs ~0.1 ms a call); the pointers below are views into it
  GridMeta* d_meta = nullptr; int32_t* d_cell_start = nullptr; int32_t* d_cursor = nullptr;
  int32_t* d_sorted_idx = nullptr; float2* d_sorted_xy = nullptr; float2* d_sorted_nrm = nullptr;
};

struct DistCache {     // one distance map per (cloud set, max_distance, resolution)
  float max_distance = 0.0f, resolution = 0.0f;
  DistMeta* d_meta = nullptr; int32_t* d_parent = nullptr;
};

struct KdCache {       // one KD-tree per cloud of the set, per (max_leaf_range, min_leaf_points)
  float max_leaf_range = 0.0f; int min_leaf_points = 0;
  void* d_block = nullptr;      // one allocation; the pointers below are views into it
  KdMeta* d_meta = nullptr; KdNode* d_nodes = nullptr; float2* d_leaf_xy = nullptr; int32_t* d_leaf_idx = nullptr; float2* d_leaf_nrm = nullptr;
  int levels = 0; long long total_nodes = 0; int max_nodes_per_cloud = 0;
  // a reserved single-cloud set (the live tracker's scan, refilled every step) keeps the allocation when its contents change: the next reset() rebuilds
  // into it -- a hipMalloc costs ~0.1 ms, more than the build of a scan's tree itself
  bool valid = true; size_t block_bytes = 0;
};

static unsigned long long next_cloudset_uid() { static std::atomic<unsigned long long> n{1}; return n.fetch_add(1, std::memory_order_relaxed); }
struct lsm2d_cloudset {
  lsm2d_context* ctx = nullptr;
  const unsigned long long uid = next_cloudset_uid();      // never reused: what a context remembers about a batch (the kept placement) names its sets by uid + version
  mutable unsigned long long version = 0;                   // bumped whenever the contents change (cloudset_drop_grids)
  mutable std::vector<GridCache> grids;
  mutable std::vector<DistCache> dists;
  mutable std::vector<KdCache> kds;
  // lane-chunked copy of xy for k_align's streaming pass (built on first use, dropped when the contents change)
  mutable float4* d_lane_xy = nullptr; mutable long long* d_lane_start = nullptr; mutable int32_t* d_lane_T = nullptr;
  mutable float4* d_lane_bounds = nullptr;      // bounding circle of every thread's chunk of every cloud (k_lane_bounds): what the culling tests
  mutable float4* d_block_bounds = nullptr;     // ... and of every block of every chunk (k_block_bounds): the block-level test of the unit lists
  mutable int32_t block_stride = kCullBlocks;   // blocks per chunk in d_block_bounds: kCullBlocksMax for a set that holds a map-sized cloud (cull_blocks_for)
  mutable float4* d_aos = nullptr;              // (x, y, nx, ny) rows of the whole set (k_aos_rows): one gather per z-buffer winner in k_align's bin walk
  hipEvent_t ev_stage = nullptr; bool stage_on_side = false;      // behind the set's latest copy OUT of h_upload queued on a side stream (the refill's pageable path): what acquire_upload_stage waits for
  hipEvent_t ev_prep = nullptr;                 // behind the set's latest preprocessing launch on the refill stream: its NEXT refill's copy (another stream) overwrites what that launch reads
  mutable float4* d_tile_bounds = nullptr; mutable int32_t* d_tile_start = nullptr;      // bounding circles of the tiles of 64 points (k_tile_bounds): the point-query finders' culling
  int32_t n_clouds = 0;
  mutable int64_t total = 0;  // logical points
  int64_t padded_total = 0;   // device points incl. even-alignment padding
  int64_t capacity = 0;       // > 0: a reserved single growable cloud (lsm2d_cloudset_create_reserved)
  float2* d_xy = nullptr; float2* d_nrm = nullptr;
  int32_t* d_start = nullptr; int32_t* d_count = nullptr;
  float* d_ranges = nullptr;      // lsm2d_preprocess_scans_refill: the device copy of the ranges the set was last refilled from
  std::vector<int32_t> h_start; mutable std::vector<int32_t> h_count;
  // Asynchronous clip / merge leave the size of a reserved set known to the device only: h_count[0] is then an UPPER BOUND
  // and count_pending is set; kernels read d_count, and whatever needs the exact number calls resolve_count() (one sync).
  mutable bool count_pending = false;
  // per-set pinned staging for lsm2d_cloudset_upload, so an upload does not have to wait for the stream
  void* h_upload = nullptr; size_t h_upload_bytes = 0; void* h_upload_dev = nullptr;
  mutable unsigned long long staged_epoch = 0;       // ctx->sync_epoch when the last transfer out of h_upload was queued (0: none pending)
  // lsm2d_cloudset_upload of a scan-sized set only fills h_upload: the unpacking into d_xy / d_nrm / d_count is queued by the first
  // consumer (flush_pending) -- or done by the aligner kernel itself in its prologue (single-alignment calls: one launch less per scan)
  mutable bool unpack_pending = false;
  // lsm2d_preprocess_scan_into likewise only stages the ranges (unless kernel timing is on): the preprocessing launch is queued by the
  // first reader -- an aligner call that reads several such sets queues them as ONE launch, one workgroup per scan (k_preprocess_multi)
  mutable bool prep_pending = false;
  mutable PrepArgs prep_args;
};

#define HIPCHK(ctx, call)                                                                           \
  do {                                                                                              \
    hipError_t e__ = (call);                                                                        \
    if (e__ != hipSuccess) {                                                                        \
      char buf__[512];                                                                              \
      snprintf(buf__, sizeof buf__, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      g_last_error = buf__;                                                                         \
      if (ctx) (ctx)->last_error = buf__;                                                           \
      return e__ == hipErrorOutOfMemory ? LSM2D_OUT_OF_MEMORY : LSM2D_DEVICE_ERROR;                 \
    }                                                                                               \
  } while (0)

// scope guard for device temporaries: every early return (HIPCHK) releases them
struct DevTmp {
  void* p = nullptr;
  ~DevTmp() { if (p) (void) hipFree(p); }
  void* release() { void* q = p; p = nullptr; return q; }
};

static int fail(lsm2d_context* ctx, int code, const char* msg) {
  g_last_error = msg;
  if (ctx) ctx->last_error = msg;
  return code;
}

// ---- the instantiations of k_align a batch can be launched as
typedef void (*AlignKernel)(const AlignArgs);
enum : unsigned { kFProj = 1, kFNN = 2, kFDist = 4, kFKd = 8 };
struct AlignVariant { unsigned finders; int mode; AlignKernel fn; };
static const AlignVariant kAlignVariants[] = {
  {kFProj, 5, k_align<true, false, false, false, 5>}, {kFProj, 0, k_align<true, false, false>},
#ifdef LSM2D_EXPERIMENTS
  {kFProj, 6, k_align<true, false, false, false, 6>},      // 5 with the XCD lockstep ("xcd_lockstep": measured, 2x slower on configs[4] for 65 % less fabric traffic: DESIGN App. A)
#endif
  {kFNN, 1, k_align<false, true, false, false, 1>},   {kFNN, 2, k_align<false, true, false, false, 2>}, {kFNN, 0, k_align<false, true, false>},
  {kFDist, 0, k_align<false, false, true>},
  {kFKd, 3, k_align<false, false, false, true, 3>},   {kFKd, 4, k_align<false, false, false, true, 4>}, {kFKd, 0, k_align<false, false, false, true>},
};
// ... and with "sum_order" 1 (k_align_seq): one instantiation per finder kind (mode 0: whatever the alignment needs, decided at run time) plus the culled
// projective stream -- the specialised modes of the table above are speed, not results, and this mode is bought for its bits
static const AlignVariant kAlignVariantsSeq[] = {
  {kFProj, 5, k_align_seq<true, false, false, false, 5>}, {kFProj, 0, k_align_seq<true, false, false>},
  {kFNN, 0, k_align_seq<false, true, false>}, {kFDist, 0, k_align_seq<false, false, true>}, {kFKd, 0, k_align_seq<false, false, false, true>},
  {kFProj | kFNN | kFDist, 0, k_align_seq<true, true, true>}, {kFProj | kFNN | kFDist | kFKd, 0, k_align_seq<true, true, true, true>},
};

extern "C" int lsm2d_version(void) { return LSM2D_VERSION; }

extern "C" const char* lsm2d_status_string(int s) {
  switch (s) {
    case LSM2D_SUCCESS: return "Success";
    case LSM2D_NOT_ENOUGH_CORRESPONDENCES: return "NotEnoughCorrespondences";
    case LSM2D_NOT_ENOUGH_INLIERS: return "NotEnoughInliers";
    case LSM2D_SINGULAR_H: return "SingularH";
    case LSM2D_BAD_ARGUMENT: return "BadArgument";
    case LSM2D_DEVICE_ERROR: return "DeviceError";
    case LSM2D_OUT_OF_MEMORY: return "OutOfMemory";
    case LSM2D_CAPACITY_EXCEEDED: return "CapacityExceeded";
    case LSM2D_NO_DEVICE: return "NoDevice";
    default: return "Unknown";
  }
}

extern "C" const char* lsm2d_last_error(const lsm2d_context* ctx) {
  return ctx ? ctx->last_error.c_str() : g_last_error.c_str();
}

extern "C" int lsm2d_create(int device_id, void* hip_stream, lsm2d_context** out) {
  if (!out) return LSM2D_BAD_ARGUMENT;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(nullptr, LSM2D_NO_DEVICE, "no HIP device visible");
  if (device_id < 0 || device_id >= n) return fail(nullptr, LSM2D_BAD_ARGUMENT, "device_id out of range");
  lsm2d_context* c = new (std::nothrow) lsm2d_context;
  if (!c) return LSM2D_OUT_OF_MEMORY;
  c->device = device_id;
  lsm2d_context* ctx = c;
  hipError_t e = hipSetDevice(device_id);
  if (e == hipSuccess && hip_stream) { c->stream = (hipStream_t) hip_stream; c->owns_stream = false; }
  else if (e == hipSuccess) { e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); c->owns_stream = true; }
  if (e == hipSuccess) e = hipEventCreate(&c->ev0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev1);
  if (e == hipSuccess) e = hipEventCreate(&c->parked.ev0);
  if (e == hipSuccess) e = hipEventCreate(&c->parked.ev1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->parked.ev_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_a_est, hipEventDisableTiming);
  if (e == hipSuccess) e = hipHostMalloc(&c->h_flag, 256, hipHostMallocDefault);
  if (e != hipSuccess) { g_last_error = hipGetErrorString(e); delete c; return LSM2D_DEVICE_ERROR; }
  // allow the big-canvas configurations to use the whole 160 KiB LDS of a CDNA4 CU
  hipDeviceProp_t prop;
  HIPCHK(ctx, hipGetDeviceProperties(&prop, device_id));
  c->max_dyn_lds = (int) prop.sharedMemPerBlock;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  (void) hipFuncSetAttribute((const void*) k_align<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_align<true, false, false, false, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
#ifdef LSM2D_EXPERIMENTS
  (void) hipFuncSetAttribute((const void*) k_align<true, false, false, false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
#endif
  (void) hipFuncSetAttribute((const void*) k_align<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_align<true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  for (const AlignVariant& v : kAlignVariantsSeq) (void) hipFuncSetAttribute((const void*) v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_align_pair, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_cull_estimate, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
#ifdef LSM2D_EXPERIMENTS
  (void) hipFuncSetAttribute((const void*) k_first_iteration, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_balance_only, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
#endif
  (void) hipFuncSetAttribute((const void*) k_kd_build_scan<1>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_kd_build_scan<0>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_kd_build_scan_multi<1>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_kd_build_scan_multi<0>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_find_projective, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_project_canvas, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_project_split, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_clip_small, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_merge_small, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_merge_multi, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_split_project<true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipFuncSetAttribute((const void*) k_split_project<false>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
  (void) hipGetLastError();
  *out = c;
  return LSM2D_SUCCESS;
}

extern "C" void lsm2d_destroy(lsm2d_context* c) {
  if (!c) return;
  (void) hipSetDevice(c->device);
  // every stream the context ever launched on comes to rest BEFORE anything is freed (batches may still be in flight on the lanes' streams; a lsm2d_pending
  // that has not been waited for is the caller's to delete -- include/lsm2d.h: wait for every begun batch before lsm2d_destroy)
  (void) hipStreamSynchronize(c->stream); ++c->sync_epoch;
  if (c->stream_b) (void) hipStreamSynchronize(c->stream_b);
  if (c->stream_c) (void) hipStreamSynchronize(c->stream_c);
  if (c->stream_h) (void) hipStreamSynchronize(c->stream_h);
  for (hipStream_t st : c->k_stream) if (st) (void) hipStreamSynchronize(st);
  for (lsm2d_cloudset* cs : c->live_sets) cs->ctx = nullptr;        // still owned by the caller: destroy them any time, use them no more
  if (c->h_stage) (void) hipHostFree(c->h_stage);
  if (c->h_flag) (void) hipHostFree(c->h_flag);
  if (c->d_scratch) (void) hipFree(c->d_scratch);
  if (c->d_split) (void) hipFree(c->d_split);
  if (c->d_kd_work) (void) hipFree(c->d_kd_work);
  if (c->d_wg_place) (void) hipFree(c->d_wg_place);
  if (c->d_order) (void) hipFree(c->d_order);
  if (c->parked.h_stage) (void) hipHostFree(c->parked.h_stage);
  if (c->parked.d_scratch) (void) hipFree(c->parked.d_scratch);
  if (c->parked.d_order) (void) hipFree(c->parked.d_order);
  for (hipEvent_t e : {c->parked.ev0, c->parked.ev1, c->parked.ev_done, c->ev_done, c->ev_b, c->ev_c, c->ev_h, c->ev_main, c->ev_a_est}) if (e) (void) hipEventDestroy(e);
  if (c->stream_b) (void) hipStreamDestroy(c->stream_b);
  if (c->stream_c) (void) hipStreamDestroy(c->stream_c);
  if (c->stream_h) (void) hipStreamDestroy(c->stream_h);
  for (hipStream_t st : c->k_stream) if (st) (void) hipStreamDestroy(st);
  if (c->d_xcd) (void) hipFree(c->d_xcd);
  for (auto& bd : c->beam_dirs) if (bd.d_dir) (void) hipFree(bd.d_dir);
  if (c->ev0) (void) hipEventDestroy(c->ev0);
  if (c->ev1) (void) hipEventDestroy(c->ev1);
  if (c->owns_stream && c->stream) (void) hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int lsm2d_synchronize(lsm2d_context* ctx) {
  if (!ctx) return LSM2D_BAD_ARGUMENT;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, stream_sync(ctx));
  if (ctx->stream_b) HIPCHK(ctx, hipStreamSynchronize(ctx->stream_b));      // (the side streams of the streamed pipeline: "everything" includes them)
  if (ctx->stream_h) HIPCHK(ctx, hipStreamSynchronize(ctx->stream_h));
  if (ctx->stream_c) HIPCHK(ctx, hipStreamSynchronize(ctx->stream_c));
  for (hipStream_t st : ctx->k_stream) if (st) HIPCHK(ctx, hipStreamSynchronize(st));
  return LSM2D_SUCCESS;
}

// ---- options ---------------------------------------------------------------------------------------------------------------------------
// ONE table (round 5; 39 strcmp lines before).  Public keys are the ones include/lsm2d.h documents: what a caller may legitimately choose.  The A/B knobs of
// measured-and-rejected or always-on alternatives (DESIGN App. A) exist only in a -DLSM2D_EXPERIMENTS build of the library: the shipped one answers
// "unknown option" to them, and "experiments" reads 0 there.
namespace {
enum : int { kOptBool = 1, kOptEven = 2, kOptResetsNotes = 4, kOptReadOnly = 8, kOptExperiment = 16 };
struct OptionDesc { const char* key; int lsm2d_context::* field; long long lo, hi; int flags; };
struct OptionDescLL { const char* key; long long lsm2d_context::* field; };
#ifdef LSM2D_EXPERIMENTS
constexpr int kCullMax = 2;
#else
constexpr int kCullMax = 1;
#endif
const OptionDesc kOptions[] = {
  // ---- public (include/lsm2d.h)
  {"kernel_timing",      &lsm2d_context::kernel_timing,      0, 1,          kOptBool},
  {"clock_stride",       &lsm2d_context::clock_stride,       0, 0x7fffffff, 0},
  {"align_path",         &lsm2d_context::align_path,         0, 3,          0},
  {"find_path",          &lsm2d_context::find_path,          0, 1,          0},
  {"zero_copy_max",      &lsm2d_context::zero_copy_max,      0, 65536,      0},
  {"cull",               &lsm2d_context::cull,               0, kCullMax,   0},
  {"balance",            &lsm2d_context::balance,            0, 1,          0},
  {"cull_margin_um",     &lsm2d_context::cull_margin_um,     0, 1000000,    0},
  {"cull_margin_urad",   &lsm2d_context::cull_margin_urad,   0, 50000,      0},      // (the kept lists' proof compares |sin dth| with the margin: asin(x) - x stays below the test's 0.05-column slack up to here)
  {"grid_big_threshold", &lsm2d_context::grid_big_threshold, 1, 0x7fffffff, 0},
  {"distmap_build",      &lsm2d_context::distmap_build,      0, 1,          0},
  {"kd_lds_nodes",       &lsm2d_context::kd_lds_nodes,       0, 4096,       0},
  {"sum_order",          &lsm2d_context::sum_order,          0, 1,          0},
  // ---- read-only
  {"last_align_path",    &lsm2d_context::last_align_path,    0, 0, kOptReadOnly},
  {"last_query_cull",    &lsm2d_context::last_query_cull,    0, 0, kOptReadOnly},
  {"max_dyn_lds",        &lsm2d_context::max_dyn_lds,        0, 0, kOptReadOnly},      // bytes of LDS one workgroup may ask for
  {"uploads",            &lsm2d_context::uploads,            0, 0, kOptReadOnly},      // host-to-device cloud uploads this context has queued so far (lsm2d_cloudset_create / _upload)
  {"last_cull_estimate", &lsm2d_context::last_cull_estimate, 0, 0, kOptReadOnly},      // 1: the latest aligner call launched the placement's estimate; 0: it reused the order of an unchanged prepared batch, or needed none
  {"last_xcd_lockstep",  &lsm2d_context::last_xcd_lockstep,  0, 0, kOptReadOnly},
  {"experiments",        &lsm2d_context::experiments,        0, 0, kOptReadOnly},
#ifdef LSM2D_EXPERIMENTS
  // ---- A/B knobs of the experiments build (the GPU suite with LSM2D_EXPERIMENTS=1; each one's measurement: DESIGN App. A)
  {"cull_est_um",        &lsm2d_context::cull_est_um,        0, 1000000, kOptExperiment},
  {"cull_est_urad",      &lsm2d_context::cull_est_urad,      0, 1000000, kOptExperiment},
  {"results_to_host",    &lsm2d_context::results_to_host,    0, 1,       kOptExperiment},
  {"two_stage",          &lsm2d_context::two_stage,          0, 1,       kOptExperiment | kOptResetsNotes},
  {"balance_notes",      &lsm2d_context::balance_notes,      0, 1,       kOptExperiment | kOptResetsNotes},
  {"cull_keep",          &lsm2d_context::cull_keep,          0, 1,       kOptExperiment | kOptBool},
  {"nn_qcache",          &lsm2d_context::nn_qcache,          0, 1,       kOptExperiment | kOptBool},
  {"nn_lds_only",        &lsm2d_context::nn_lds_only,        0, 1,       kOptExperiment | kOptBool},
  {"kd_modes",           &lsm2d_context::kd_modes,           0, 1,       kOptExperiment | kOptBool},
  {"proj_modes",         &lsm2d_context::proj_modes,         0, 1,       kOptExperiment | kOptBool},
  {"cull_block",         &lsm2d_context::cull_block,         0, 4096,    kOptExperiment | kOptEven},
  {"kd_chain",           &lsm2d_context::kd_chain,           0, 1,       kOptExperiment},
  {"grid_big_cells_x10", &lsm2d_context::grid_big_cells_x10, 5, 400,     kOptExperiment},
  {"kd_scan_max_clouds", &lsm2d_context::kd_scan_max_clouds, 0, 0x7fffffff, kOptExperiment},
  {"kd_wide_min_points", &lsm2d_context::kd_wide_min_points, 0, 0x7fffffff, kOptExperiment},
  {"kd_wg_max_points",   &lsm2d_context::kd_wg_max_points,   0, 1 << 20, kOptExperiment},
  {"estimate_reuse",     &lsm2d_context::estimate_reuse,     0, 1,       kOptExperiment},
  {"lane_streams",       &lsm2d_context::lane_streams,       0, 1,       kOptExperiment},
  {"xcd_lockstep",       &lsm2d_context::xcd_lockstep,       0, 64,      kOptExperiment},
#endif
};
const OptionDescLL kOptionsLL[] = {      // read-only, 64-bit
  {"last_kd_levels", &lsm2d_context::last_kd_levels}, {"last_kd_nodes", &lsm2d_context::last_kd_nodes},
  {"last_kernel_clock_khz", &lsm2d_context::last_clock_khz}, {"last_workgroup_lifetime_ns", &lsm2d_context::last_wg_lifetime_ns},
  {"last_h2d_bytes", &lsm2d_context::last_h2d_bytes},
};
}  // namespace

extern "C" int lsm2d_set_option(lsm2d_context* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return LSM2D_BAD_ARGUMENT;
  for (const OptionDesc& o : kOptions) {
    if (strcmp(key, o.key)) continue;
    if (o.flags & kOptReadOnly) return fail(ctx, LSM2D_BAD_ARGUMENT, "set_option: this key is read-only");
    if (o.flags & kOptBool) value = value != 0;
    if (value < o.lo || value > o.hi || ((o.flags & kOptEven) && (value & 1))) {
      char buf[160]; snprintf(buf, sizeof buf, "set_option: %s must be %sin %lld .. %lld", o.key, (o.flags & kOptEven) ? "even and " : "", o.lo, o.hi);
      return fail(ctx, LSM2D_BAD_ARGUMENT, buf);
    }
    ctx->*(o.field) = (int) value;
    if (o.flags & kOptResetsNotes) ctx->wg_place_shape = 0;
    if (o.field == &lsm2d_context::kernel_timing && !value) ctx->have_timing = false;
    ctx->order_valid = false;      // whatever changed may change what a batch launches: the next call makes its placement afresh
    return LSM2D_SUCCESS;
  }
  return fail(ctx, LSM2D_BAD_ARGUMENT, "unknown option");
}

extern "C" int lsm2d_get_option(lsm2d_context* ctx, const char* key, int64_t* out_value) {
  if (!ctx || !key || !out_value) return LSM2D_BAD_ARGUMENT;
  for (const OptionDesc& o : kOptions) if (!strcmp(key, o.key)) { *out_value = ctx->*(o.field); return LSM2D_SUCCESS; }
  for (const OptionDescLL& o : kOptionsLL) if (!strcmp(key, o.key)) { *out_value = ctx->*(o.field); return LSM2D_SUCCESS; }
  return fail(ctx, LSM2D_BAD_ARGUMENT, "unknown option");
}

extern "C" int lsm2d_last_kernel_ms(lsm2d_context* ctx, float* out_ms) {
  if (!ctx || !out_ms) return LSM2D_BAD_ARGUMENT;
  if (!ctx->have_timing) return fail(ctx, LSM2D_BAD_ARGUMENT, "no timed launch yet");
  hipEvent_t e0 = ctx->last_ev0 ? ctx->last_ev0 : ctx->ev0, e1 = ctx->last_ev1 ? ctx->last_ev1 : ctx->ev1;
  HIPCHK(ctx, hipEventSynchronize(e1));
  HIPCHK(ctx, hipEventElapsedTime(out_ms, e0, e1));
  return LSM2D_SUCCESS;
}

// a timed launch outside the batch entry points recorded the CURRENT lane's events: they are what lsm2d_last_kernel_ms reads next (round-5 advisor: after an
// asynchronous begin had swapped lanes the call kept answering with the old batch's events)
static void note_timed(lsm2d_context* ctx, bool timed) { ctx->have_timing = timed; ctx->last_ev0 = ctx->ev0; ctx->last_ev1 = ctx->ev1; }

static bool valid_cloud_index_fwd(const lsm2d_cloudset* cs, int32_t i);
// what a caller's buffer is to the runtime (entry points that take bulk input accept all three)
enum class PtrKind { pageable, pinned, device };
static PtrKind pointer_kind(const void* p, int* device = nullptr) {
  hipPointerAttribute_t at; memset(&at, 0, sizeof at);
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void) hipGetLastError(); return PtrKind::pageable; }     // plain malloc memory: an error on some runtimes
  if (device) *device = at.device;
  if (at.type == hipMemoryTypeDevice) return PtrKind::device;
  if (at.type == hipMemoryTypeHost) return PtrKind::pinned;
  return PtrKind::pageable;                      // unregistered; managed memory is treated as host memory the runtime can page
}

// (both lanes taken by batches in flight: the "current" staging buffers are the older batch's -- its arguments, its results.  Every entry point that stages
// anything comes through here and fails loudly instead of writing over them; lsm2d_align_batch_begin says the same before it gets here)
static const char* const kBothLanesBusy = "two batches are in flight on this context and its staging buffers are theirs: wait for the older one first (lsm2d_align_batch_wait)";
static int ensure_stage(lsm2d_context* ctx, size_t bytes) {
  if (ctx->lane_busy) return fail(ctx, LSM2D_BAD_ARGUMENT, kBothLanesBusy);
  if (bytes <= ctx->h_stage_bytes) return LSM2D_SUCCESS;
  if (ctx->h_stage) { HIPCHK(ctx, stream_sync(ctx)); HIPCHK(ctx, hipHostFree(ctx->h_stage)); ctx->h_stage = nullptr; ctx->h_stage_bytes = 0; }
  size_t cap = bytes + bytes / 2 + 4096;
  HIPCHK(ctx, hipHostMalloc(&ctx->h_stage, cap, hipHostMallocCoherent | hipHostMallocMapped));
  HIPCHK(ctx, hipHostGetDevicePointer(&ctx->h_stage_dev, ctx->h_stage, 0));      // looked up once: kernels read / write the buffer directly
  ctx->h_stage_bytes = cap;
  return LSM2D_SUCCESS;
}
// device-side address of the pinned staging buffer: kernels with small outputs write them there directly (no device-to-host copy)
static int stage_device_view(lsm2d_context* ctx, char** out) {
  *out = (char*) ctx->h_stage_dev;
  return LSM2D_SUCCESS;
}

static int ensure_scratch(lsm2d_context* ctx, size_t bytes) {
  if (ctx->lane_busy) return fail(ctx, LSM2D_BAD_ARGUMENT, kBothLanesBusy);
  ctx->inputs_valid = false;      // (whoever asks is about to write the scratch; lsm2d_align_batch looks at the flag before it asks)
  if (bytes <= ctx->d_scratch_bytes) return LSM2D_SUCCESS;
  // (a NEW allocation holds nobody's input block: the shadow goes with the old one -- round-5 advisor: a batch run again with more outputs, e.g. statistics,
  // grew the scratch, compared equal against the shadow and skipped the upload into memory that had never seen it)
  ctx->inputs_shadow.clear();
  if (ctx->d_scratch) { HIPCHK(ctx, stream_sync(ctx)); HIPCHK(ctx, hipFree(ctx->d_scratch)); ctx->d_scratch = nullptr; ctx->d_scratch_bytes = 0; }
  size_t cap = bytes + bytes / 2 + 4096;
  HIPCHK(ctx, hipMalloc(&ctx->d_scratch, cap));
  ctx->d_scratch_bytes = cap;
  return LSM2D_SUCCESS;
}

// ---- cloud sets -----------------------------------------------------------------------------------
static int cloudset_layout(lsm2d_cloudset* cs, const int32_t* offsets, int32_t n_clouds, int64_t total) {
  cs->n_clouds = n_clouds; cs->total = total;
  cs->h_start.resize(n_clouds); cs->h_count.resize(n_clouds);
  int64_t p = 0;
  for (int32_t c = 0; c < n_clouds; ++c) {
    const int64_t b = offsets ? offsets[c] : 0, e = offsets ? offsets[c + 1] : total;
    if (b < 0 || e < b || e > total) return LSM2D_BAD_ARGUMENT;
    cs->h_start[c] = (int32_t) p; cs->h_count[c] = (int32_t) (e - b);
    p += (e - b); p += (p & 1);        // every cloud starts on an even point index (16-byte aligned xy)
    if (p > 0x7fffffff - 2) return LSM2D_CAPACITY_EXCEEDED;
  }
  if (offsets && (offsets[0] != 0 || offsets[n_clouds] != total)) return LSM2D_BAD_ARGUMENT;
  cs->padded_total = p + 2;            // slack so the last lane's 16-byte load stays in bounds
  return LSM2D_SUCCESS;
}

static int cloudset_alloc(lsm2d_context* ctx, lsm2d_cloudset* cs) {
  HIPCHK(ctx, hipMalloc((void**) &cs->d_xy, sizeof(float2) * (size_t) cs->padded_total));
  HIPCHK(ctx, hipMalloc((void**) &cs->d_nrm, sizeof(float2) * (size_t) cs->padded_total));
  HIPCHK(ctx, hipMalloc((void**) &cs->d_start, sizeof(int32_t) * (size_t) cs->n_clouds));
  HIPCHK(ctx, hipMalloc((void**) &cs->d_count, sizeof(int32_t) * (size_t) cs->n_clouds));
  HIPCHK(ctx, hipMemsetAsync(cs->d_xy, 0, sizeof(float2) * (size_t) cs->padded_total, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(cs->d_nrm, 0, sizeof(float2) * (size_t) cs->padded_total, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(cs->d_start, cs->h_start.data(), sizeof(int32_t) * (size_t) cs->n_clouds, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(cs->d_count, cs->h_count.data(), sizeof(int32_t) * (size_t) cs->n_clouds, hipMemcpyHostToDevice, ctx->stream));
  return LSM2D_SUCCESS;
}

static int cloudset_create_impl(lsm2d_context* ctx, const void* points, bool on_device, const int32_t* offsets,
                                int32_t n_clouds, int64_t total, lsm2d_cloudset** out) {
  if (!ctx || !out || n_clouds < 1 || total < 0 || (total > 0 && !points) || (n_clouds > 1 && !offsets))
    return fail(ctx, LSM2D_BAD_ARGUMENT, "cloudset_create: bad argument");
  *out = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  lsm2d_cloudset* cs = new (std::nothrow) lsm2d_cloudset;
  if (!cs) return LSM2D_OUT_OF_MEMORY;
  cs->ctx = ctx; ctx->live_sets.push_back(cs);
  int rc = cloudset_layout(cs, offsets, n_clouds, total);
  if (rc == LSM2D_SUCCESS) rc = cloudset_alloc(ctx, cs);
  if (rc != LSM2D_SUCCESS) { lsm2d_cloudset_destroy(cs); return rc; }
  if (total > 0) {
    // stage the AoS points (and the logical offsets) on the device, then split / pad with one kernel
    const size_t pts_bytes = sizeof(float4) * (size_t) total, off_bytes = sizeof(int32_t) * (size_t) (n_clouds + 1);
    void* d_src = nullptr; int32_t* d_off = nullptr;
    std::vector<int32_t> one = {0, (int32_t) total};
    const int32_t* h_off = offsets ? offsets : one.data();
    hipError_t e = hipSuccess;
    if (!on_device) {
      ++ctx->uploads; ctx->last_h2d_bytes = (long long) pts_bytes;
      e = hipMalloc(&d_src, pts_bytes);
      if (e == hipSuccess) e = hipMemcpyAsync(d_src, points, pts_bytes, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipMalloc((void**) &d_off, off_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, h_off, off_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      const int block = 256;
      long long blocks = (total + block - 1) / block; if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(k_repack_cloud, dim3((unsigned) blocks), dim3(block), 0, ctx->stream,
                         (const float4*) (on_device ? points : d_src), d_off, cs->d_start, n_clouds, (long long) total, cs->d_xy, cs->d_nrm);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = stream_sync(ctx);
    if (d_src) (void) hipFree(d_src);
    if (d_off) (void) hipFree(d_off);
    if (e != hipSuccess) { lsm2d_cloudset_destroy(cs); HIPCHK(ctx, e); }
  } else {
    HIPCHK(ctx, stream_sync(ctx));
  }
  *out = cs;
  return LSM2D_SUCCESS;
}

extern "C" int lsm2d_cloudset_create(lsm2d_context* ctx, const float* pts, const int32_t* offsets, int32_t n_clouds,
                                     int64_t total, lsm2d_cloudset** out) {
  return cloudset_create_impl(ctx, pts, false, offsets, n_clouds, total, out);
}
extern "C" int lsm2d_cloudset_create_from_device(lsm2d_context* ctx, const void* d_pts, const int32_t* offsets,
                                                 int32_t n_clouds, int64_t total, lsm2d_cloudset** out) {
  return cloudset_create_impl(ctx, d_pts, true, offsets, n_clouds, total, out);
}
extern "C" void lsm2d_cloudset_destroy(lsm2d_cloudset* cs) {
  if (!cs) return;
  if (cs->ctx) (void) hipSetDevice(cs->ctx->device);
  if (cs->d_xy) (void) hipFree(cs->d_xy);
  if (cs->d_nrm) (void) hipFree(cs->d_nrm);
  if (cs->d_start) (void) hipFree(cs->d_start);
  if (cs->d_count) (void) hipFree(cs->d_count);
  if (cs->d_ranges) (void) hipFree(cs->d_ranges);
  if (cs->ev_prep) (void) hipEventDestroy(cs->ev_prep);
  if (cs->ev_stage) (void) hipEventDestroy(cs->ev_stage);
  for (auto& g : cs->grids) if (g.d_block) (void) hipFree(g.d_block);
  for (auto& d : cs->dists) { if (d.d_meta) (void) hipFree(d.d_meta); if (d.d_parent) (void) hipFree(d.d_parent); }
  for (auto& k : cs->kds) if (k.d_block) (void) hipFree(k.d_block);
  if (cs->d_lane_xy) (void) hipFree(cs->d_lane_xy);
  if (cs->d_lane_start) (void) hipFree(cs->d_lane_start);
  if (cs->d_lane_T) (void) hipFree(cs->d_lane_T);
  if (cs->d_lane_bounds) (void) hipFree(cs->d_lane_bounds);
  if (cs->d_block_bounds) (void) hipFree(cs->d_block_bounds);
  if (cs->d_aos) (void) hipFree(cs->d_aos);
  if (cs->d_tile_bounds) (void) hipFree(cs->d_tile_bounds);
  if (cs->d_tile_start) (void) hipFree(cs->d_tile_start);
  if (cs->ctx && cs->staged_epoch == cs->ctx->sync_epoch) (void) stream_sync(cs->ctx);      // a staged transfer may still be reading h_upload
  if (cs->h_upload) (void) hipHostFree(cs->h_upload);
  if (cs->ctx) { auto& v = cs->ctx->live_sets; for (size_t i = 0; i < v.size(); ++i) if (v[i] == cs) { v[i] = v.back(); v.pop_back(); break; } }
  delete cs;
}
static int flush_pending(const lsm2d_cloudset* cs);
static int resolve_count(const lsm2d_cloudset* cs) {
  if (!cs || !cs->count_pending) return LSM2D_SUCCESS;
  if (!cs->ctx) return LSM2D_BAD_ARGUMENT;      // its context is gone
  { const int rc0 = flush_pending(cs); if (rc0) return rc0; }      // a preprocessing launch still pending: its result is the count asked for
  lsm2d_context* ctx = cs->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, stream_sync(ctx));
  if (ctx->stream_b) HIPCHK(ctx, hipStreamSynchronize(ctx->stream_b));
  if (ctx->stream_h) HIPCHK(ctx, hipStreamSynchronize(ctx->stream_h));
  if (ctx->stream_c) HIPCHK(ctx, hipStreamSynchronize(ctx->stream_c));      // (a refill queued on the refill stream)
  if (cs->n_clouds > 1) {      // a refilled set of scans (lsm2d_preprocess_scans_refill): every cloud's size
    HIPCHK(ctx, hipMemcpy(cs->h_count.data(), cs->d_count, sizeof(int32_t) * (size_t) cs->n_clouds, hipMemcpyDeviceToHost));
    cs->total = 0; for (int c = 0; c < cs->n_clouds; ++c) cs->total += cs->h_count[c];
    cs->count_pending = false;
    return LSM2D_SUCCESS;
  }
  int32_t n = 0;
  HIPCHK(ctx, hipMemcpy(&n, cs->d_count, sizeof(int32_t), hipMemcpyDeviceToHost));
  cs->h_count[0] = n; cs->total = n; cs->count_pending = false;
  return LSM2D_SUCCESS;
}
// queues the unpacking of a set whose latest upload still sits in its pinned buffer; every reader of the device arrays calls it
static int flush_pending(const lsm2d_cloudset* cs) {
  if (cs && !cs->ctx) return LSM2D_BAD_ARGUMENT;      // its context is gone
  if (cs && cs->prep_pending) {
    lsm2d_context* ctx = cs->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_preprocess_scans, dim3(1), dim3(kPrepBlock), 0, ctx->stream, cs->prep_args);
    HIPCHK(ctx, hipGetLastError());
    cs->prep_pending = false; cs->staged_epoch = ctx->sync_epoch;
    return LSM2D_SUCCESS;
  }
  if (!cs || !cs->unpack_pending) return LSM2D_SUCCESS;
  lsm2d_context* ctx = cs->ctx;
  const int n = cs->h_count[0];
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_upload_unpack, dim3((unsigned) (n > 4096 ? 16 : (n + 255) / 256 > 0 ? (n + 255) / 256 : 1)), dim3(256), 0, ctx->stream,
                     (const float4*) cs->h_upload_dev, n, cs->d_xy, cs->d_nrm, cs->d_count);
  HIPCHK(ctx, hipGetLastError());
  cs->unpack_pending = false; cs->staged_epoch = ctx->sync_epoch;
  return LSM2D_SUCCESS;
}
// the sets an aligner call reads: pending preprocessing launches go out together, one workgroup per scan
static int flush_preprocessing_together(lsm2d_context* ctx, const lsm2d_cloudset* const* sets, int n_sets) {
  const lsm2d_cloudset* todo[kPrepMulti]; int nt = 0;
  for (int i = 0; i < n_sets; ++i) {
    const lsm2d_cloudset* cs = sets[i];
    if (!cs || !cs->prep_pending) continue;
    if (cs->ctx != ctx) return fail(ctx, LSM2D_BAD_ARGUMENT, "cloud set from another (or a destroyed) context");
    bool seen = false; for (int k = 0; k < nt; ++k) seen = seen || todo[k] == cs;
    if (seen) continue;
    if (nt == kPrepMulti) { const int rc = flush_pending(cs); if (rc) return rc; continue; }
    todo[nt++] = cs;
  }
  if (nt == 0) return LSM2D_SUCCESS;
  if (nt == 1) return flush_pending(todo[0]);
  PrepMultiArgs M;
  for (int k = 0; k < nt; ++k) M.a[k] = todo[k]->prep_args;
  for (int k = nt; k < kPrepMulti; ++k) M.a[k] = todo[0]->prep_args;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_preprocess_multi, dim3((unsigned) nt), dim3(kPrepBlock), 0, ctx->stream, M);
  HIPCHK(ctx, hipGetLastError());
  for (int k = 0; k < nt; ++k) { todo[k]->prep_pending = false; todo[k]->staged_epoch = ctx->sync_epoch; }
  return LSM2D_SUCCESS;
}
extern "C" int32_t lsm2d_cloudset_num_clouds(const lsm2d_cloudset* cs) { return cs ? cs->n_clouds : 0; }
extern "C" int64_t lsm2d_cloudset_num_points(const lsm2d_cloudset* cs) { return (cs && resolve_count(cs) == LSM2D_SUCCESS) ? cs->total : 0; }
extern "C" int64_t lsm2d_cloudset_cloud_size(const lsm2d_cloudset* cs, int32_t i) {
  return (cs && i >= 0 && i < cs->n_clouds && resolve_count(cs) == LSM2D_SUCCESS) ? cs->h_count[i] : -1;
}

extern "C" int32_t lsm2d_cloudset_cloud_sizes(const lsm2d_cloudset* cs, int32_t* out, int32_t capacity) {
  if (!cs || capacity < 0 || (capacity > 0 && !out)) return LSM2D_BAD_ARGUMENT;
  const int rc = resolve_count(cs); if (rc != LSM2D_SUCCESS) return rc;
  const int32_t n = capacity < cs->n_clouds ? capacity : cs->n_clouds;
  for (int32_t i = 0; i < n; ++i) out[i] = cs->h_count[(size_t) i];
  return n;
}

static void cloudset_drop_grids(const lsm2d_cloudset* cs) {     // the contents changed: cached NN grids are stale
  ++cs->version;
  for (auto& g : cs->grids) if (g.d_block) (void) hipFree(g.d_block);
  cs->grids.clear();
  if (cs->d_lane_xy) { (void) hipFree(cs->d_lane_xy); cs->d_lane_xy = nullptr; }
  if (cs->d_lane_start) { (void) hipFree(cs->d_lane_start); cs->d_lane_start = nullptr; }
  if (cs->d_lane_T) { (void) hipFree(cs->d_lane_T); cs->d_lane_T = nullptr; }
  if (cs->d_lane_bounds) { (void) hipFree(cs->d_lane_bounds); cs->d_lane_bounds = nullptr; }
  if (cs->d_block_bounds) { (void) hipFree(cs->d_block_bounds); cs->d_block_bounds = nullptr; }
  if (cs->d_aos) { (void) hipFree(cs->d_aos); cs->d_aos = nullptr; }
  if (cs->d_tile_bounds) { (void) hipFree(cs->d_tile_bounds); cs->d_tile_bounds = nullptr; }
  if (cs->d_tile_start) { (void) hipFree(cs->d_tile_start); cs->d_tile_start = nullptr; }
  for (auto& d : cs->dists) { if (d.d_meta) (void) hipFree(d.d_meta); if (d.d_parent) (void) hipFree(d.d_parent); }
  cs->dists.clear();
  if (cs->capacity > 0 && cs->n_clouds == 1 && cs->kds.size() == 1) cs->kds[0].valid = false;      // a reserved set: the allocation is recycled by the next build (KdCache::valid)
  else { for (auto& k : cs->kds) if (k.d_block) (void) hipFree(k.d_block); cs->kds.clear(); }
}

extern "C" int lsm2d_cloudset_create_reserved(lsm2d_context* ctx, int64_t capacity, lsm2d_cloudset** out) {
  if (!ctx || !out || capacity < 1 || capacity > 0x7ffffff0) return fail(ctx, LSM2D_BAD_ARGUMENT, "cloudset_create_reserved: bad argument");
  *out = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  lsm2d_cloudset* cs = new (std::nothrow) lsm2d_cloudset;
  if (!cs) return LSM2D_OUT_OF_MEMORY;
  cs->ctx = ctx; ctx->live_sets.push_back(cs); cs->n_clouds = 1; cs->total = 0; cs->capacity = capacity; cs->padded_total = capacity + (capacity & 1) + 2;
  cs->h_start.assign(1, 0); cs->h_count.assign(1, 0);
  int rc = cloudset_alloc(ctx, cs);
  if (rc == LSM2D_SUCCESS) { hipError_t e = stream_sync(ctx); if (e != hipSuccess) rc = LSM2D_DEVICE_ERROR; }
  if (rc != LSM2D_SUCCESS) { lsm2d_cloudset_destroy(cs); return rc; }
  *out = cs;
  return LSM2D_SUCCESS;
}

static int set_single_count(lsm2d_context* ctx, lsm2d_cloudset* cs, int32_t n) {
  cs->h_count[0] = n; cs->total = n;
  HIPCHK(ctx, hipMemcpyAsync(cs->d_count, cs->h_count.data(), sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  return LSM2D_SUCCESS;
}

// the set's own pinned staging buffer, free to be overwritten: waits for the set's previous staged transfer only
static int acquire_upload_stage(lsm2d_cloudset* cs, size_t need) {
  lsm2d_context* ctx = cs->ctx;
  // free to overwrite once the stream has been waited for since the last staged transfer was queued (the aligner call in between
  // does that); two uploads to the same set back to back wait here
  if (cs->staged_epoch == ctx->sync_epoch) HIPCHK(ctx, stream_sync(ctx));
  // ... and a copy queued on a side stream (the refill while a batch is in flight) is not covered by the context stream's epoch: its own event
  if (cs->stage_on_side) { HIPCHK(ctx, hipEventSynchronize(cs->ev_stage)); cs->stage_on_side = false; }
  if (need > cs->h_upload_bytes) {
    if (cs->h_upload) { HIPCHK(ctx, hipHostFree(cs->h_upload)); cs->h_upload = nullptr; cs->h_upload_bytes = 0; }
    const size_t want = cs->capacity > 0 ? sizeof(float) * 4 * (size_t) cs->capacity + 16 : need;
    HIPCHK(ctx, hipHostMalloc(&cs->h_upload, want > need ? want : need, hipHostMallocCoherent | hipHostMallocMapped));
    HIPCHK(ctx, hipHostGetDevicePointer(&cs->h_upload_dev, cs->h_upload, 0));
    cs->h_upload_bytes = want > need ? want : need;
  }
  return LSM2D_SUCCESS;
}

extern "C" int lsm2d_cloudset_upload(lsm2d_cloudset* cs, const float* pts, int64_t n) {
  if (!cs || !cs->ctx || cs->n_clouds != 1 || n < 0 || (n > 0 && !pts)) return fail(cs ? cs->ctx : nullptr, LSM2D_BAD_ARGUMENT, "cloudset_upload: bad argument");
  lsm2d_context* ctx = cs->ctx;
  const int64_t cap = cs->capacity > 0 ? cs->capacity : cs->padded_total - 2;
  if (n > cap) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "cloudset_upload: does not fit the allocation");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  cloudset_drop_grids(cs);
  ++ctx->uploads; ctx->last_h2d_bytes = (long long) (sizeof(float) * 4 * (size_t) n);
  // no allocation and no wait for the stream -- only for this set's PREVIOUS upload (an event), whose source the staging buffer
  // still is until it ran
  { const int rc0 = acquire_upload_stage(cs, sizeof(float) * 4 * (size_t) (n > 0 ? n : 1) + 16); if (rc0) return rc0; }
  cs->count_pending = false; cs->prep_pending = false;
  cs->h_count[0] = (int32_t) n; cs->total = n;
  if (n <= 16384) {
    // scan-sized: the points go into the pinned buffer as they are and ONE small kernel reads them over the bus, splits them into
    // the coordinate / normal arrays and sets the count (three host-to-device copies cost three times the API and launch overhead)
    // -- queued by the set's first reader (flush_pending), or done by the aligner kernel itself
    if (n) memcpy(cs->h_upload, pts, sizeof(float) * 4 * (size_t) n);
    cs->unpack_pending = true;
    return LSM2D_SUCCESS;
  } else {
    cs->unpack_pending = false;
    // split on the host, then plain async copies
    float2* hxy = (float2*) cs->h_upload; float2* hn = hxy + n;
    for (int64_t i = 0; i < n; ++i) { hxy[i] = make_float2(pts[4 * i], pts[4 * i + 1]); hn[i] = make_float2(pts[4 * i + 2], pts[4 * i + 3]); }
    if (n) {
      HIPCHK(ctx, hipMemcpyAsync(cs->d_xy, hxy, sizeof(float2) * (size_t) n, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(cs->d_nrm, hn, sizeof(float2) * (size_t) n, hipMemcpyHostToDevice, ctx->stream));
    }
    // the count travels in the same staging buffer (h_count may be rewritten by the host before the copy runs)
    int32_t* hcnt = (int32_t*) ((char*) cs->h_upload + cs->h_upload_bytes - sizeof(int32_t));
    *hcnt = (int32_t) n;
    HIPCHK(ctx, hipMemcpyAsync(cs->d_count, hcnt, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  }
  cs->staged_epoch = ctx->sync_epoch;
  return LSM2D_SUCCESS;
}

extern "C" int lsm2d_cloudset_download(const lsm2d_cloudset* cs, int32_t ci, float* out, int64_t capacity, int64_t* out_n) {
  if (!cs || !cs->ctx || !out_n || !valid_cloud_index_fwd(cs, ci)) return fail(cs ? cs->ctx : nullptr, LSM2D_BAD_ARGUMENT, "cloudset_download: bad argument");
  lsm2d_context* ctx = cs->ctx;
  { int rc0 = resolve_count(cs); if (rc0) return rc0; rc0 = flush_pending(cs); if (rc0) return rc0; }
  const int64_t n = cs->h_count[ci];
  *out_n = n;
  if (n > capacity || (n > 0 && !out)) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "cloudset_download: out buffer too small");
  if (n == 0) return LSM2D_SUCCESS;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = sizeof(float4) * (size_t) n;
  int rc = ensure_scratch(ctx, bytes); if (rc) return rc;
  rc = ensure_stage(ctx, bytes); if (rc) return rc;
  const int base = cs->h_start[ci];
  hipLaunchKernelGGL(k_pack_aos, dim3((unsigned) ((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const float2*) (cs->d_xy + base), (const float2*) (cs->d_nrm + base), (int) n, (float4*) ctx->d_scratch);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_stage, ctx->d_scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, stream_sync(ctx));
  memcpy(out, ctx->h_stage, bytes);
  return LSM2D_SUCCESS;
}

// ---- parameter packing -------------------------------------------------------------------------------
// Depths are compared through r2: sqrtf is correctly rounded and monotone, so
//   rmin <= sqrtf(r2)  <=>  r2 >= r2lo,  r2lo = smallest float whose sqrtf reaches rmin   (same for rmax).
// Ranges are clamped to [1e-15, 1e18] m so r2 stays a normal fp32 number (the oracle applies the same clamp).
static float r2_lower_threshold(float rmin) {
  float v = rmin * rmin;
  while (sqrtf(v) >= rmin) v = nextafterf(v, 0.0f);
  while (sqrtf(v) < rmin) v = nextafterf(v, INFINITY);
  return v;
}
static float r2_upper_threshold(float rmax) {
  float v = rmax * rmax;
  while (sqrtf(v) <= rmax) v = nextafterf(v, INFINITY);
  while (sqrtf(v) > rmax) v = nextafterf(v, 0.0f);
  return v;
}
static bool make_projk(const lsm2d_projector& p, ProjK* k) {
  if (p.canvas_cols <= 0 || !(p.angle_max > p.angle_min) || !(p.range_max >= p.range_min) || !(p.range_min >= 0.0f)) return false;
  k->cols = p.canvas_cols;
  k->K00 = (float) p.canvas_cols / (p.angle_max - p.angle_min);
  k->K01 = (float) p.canvas_cols * 0.5f + p.col_offset;
  k->rmin = fmaxf(p.range_min, 1e-15f); k->rmax = fminf(p.range_max, 1e18f); k->colsf = (float) p.canvas_cols;
  if (!(k->rmax >= k->rmin)) return false;
  k->r2lo = r2_lower_threshold(k->rmin); k->r2hi = r2_upper_threshold(k->rmax);
  // The stream's short quotient min(|x|,|y|) / r (div_by_depth) is exact for min >= 1e-12; below that its result is merely SOME value
  // t' with |t'| <= A = 2.1e-12 / r, r >= rmin.  That cannot move a column when (i) pi/2 - t' and pi - t' round back to pi/2 and pi
  // (A below a quarter ulp of pi/2) and (ii) K00 * (+-t') + K01 rounds to K01 (K00 A below a quarter ulp of K01): then the kernels
  // drop the guard branch in front of the quotient.  True for every sane projector (cols 1081, 2 pi: K00 A ~ 2e-9 vs 1.5e-5).
  const double A = 2.1e-12 / (0.7 * (double) k->rmin);
  k->tiny_ok = (A < 2.5e-8 && k->K01 >= 0.25f && (double) k->K00 * A < (double) k->K01 * 1.4e-8) ? 1 : 0;
  return true;
}
static CloudDev cloud_dev(const lsm2d_cloudset* cs, const int32_t* d_index) {
  CloudDev c; c.xy = cs->d_xy; c.nrm = cs->d_nrm; c.start = cs->d_start; c.count = cs->d_count; c.index = d_index; c.n_clouds = cs->n_clouds;
  c.lane_xy = cs->d_lane_xy; c.lane_start = cs->d_lane_start; c.lane_T = cs->d_lane_T; c.lane_bounds = cs->d_lane_bounds;
  c.block_bounds = cs->d_block_bounds; c.block_stride = cs->block_stride; c.aos = cs->d_aos;
  c.tile_bounds = cs->d_tile_bounds; c.tile_start = cs->d_tile_start;
  c.grid = GridDev{nullptr, nullptr, nullptr, nullptr, nullptr};
  c.dist = DistDev{nullptr, nullptr};
  c.kd = KdDev{nullptr, nullptr, nullptr, nullptr, nullptr};
  return c;
}
// NN finder: uniform grid over every cloud of the (fixed) set, cached per max_distance.  Replaces
// CorrespondenceFinderKDTree2D::reset() (registration/correspondence_finder_kd_tree_2d.cpp:31-38).
static int ensure_grid(lsm2d_context* ctx, const lsm2d_cloudset* cs, float max_distance, GridDev* out) {
  for (const auto& g : cs->grids)
    if (g.max_distance == max_distance) { *out = GridDev{g.d_meta, g.d_cell_start, g.d_sorted_idx, g.d_sorted_xy, g.d_sorted_nrm}; return LSM2D_SUCCESS; }
  const int nc = cs->n_clouds;
  std::vector<int32_t> cell_base(nc), gcap(nc);
  int64_t cells = 0;
  for (int c = 0; c < nc; ++c) {
    // cells per side: ~3 sqrt(n) for scan-sized clouds (their queries mostly find empty blocks; finer cells only add block
    // levels), ~6 sqrt(n) for map-sized ones (dense walls: the 3x3 block of a converged query holds 3x fewer candidates).
    // Measured on configs[1], role B: 3 sqrt(n) 3.12 ms, 6 sqrt(n) 2.95 ms, 12 sqrt(n) 3.24 ms (cell table out of L2).
    int cap = (int) ceil((cs->h_count[c] >= 16384 ? 0.1 * (double) ctx->grid_big_cells_x10 : 3.0) * sqrt((double) cs->h_count[c]));
    cap = cap < 16 ? 16 : (cap > 2048 ? 2048 : cap);       // 4096 buys 13 % on a 1M-point map for 4x the cell table: not taken
    gcap[c] = cap; cell_base[c] = (int32_t) cells; cells += (int64_t) cap * cap + 1;
    if (cells > 0x7fffffff) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "grid: too many cells");
  }
  GridCache g; g.max_distance = max_distance;
  DevTmp t_block;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t) 255; return o; };
  const size_t o_meta = take(sizeof(GridMeta) * (size_t) nc), o_start = take(sizeof(int32_t) * (size_t) cells), o_cursor = take(sizeof(int32_t) * (size_t) cells);
  const size_t o_sidx = take(sizeof(int32_t) * (size_t) cs->padded_total), o_sxy = take(sizeof(float2) * (size_t) cs->padded_total), o_snr = take(sizeof(float2) * (size_t) cs->padded_total);
  const size_t o_base = take(sizeof(int32_t) * (size_t) nc), o_gcap = take(sizeof(int32_t) * (size_t) nc), o_tiles = take(sizeof(int32_t) * 2048);
  HIPCHK(ctx, hipMalloc(&t_block.p, off));
  char* blk = (char*) t_block.p;
  g.d_meta = (GridMeta*) (blk + o_meta); g.d_cell_start = (int32_t*) (blk + o_start); g.d_cursor = (int32_t*) (blk + o_cursor);
  g.d_sorted_idx = (int32_t*) (blk + o_sidx); g.d_sorted_xy = (float2*) (blk + o_sxy); g.d_sorted_nrm = (float2*) (blk + o_snr);
  int32_t* d_base = (int32_t*) (blk + o_base); int32_t* d_gcap = (int32_t*) (blk + o_gcap); int32_t* d_tiles = (int32_t*) (blk + o_tiles);
  HIPCHK(ctx, hipMemcpyAsync(d_base, cell_base.data(), sizeof(int32_t) * (size_t) nc, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_gcap, gcap.data(), sizeof(int32_t) * (size_t) nc, hipMemcpyHostToDevice, ctx->stream));
  GridBuildArgs A;
  A.xy = cs->d_xy; A.start = cs->d_start; A.count = cs->d_count; A.n_clouds = nc; A.h_min = max_distance * 0.015625f;
  A.cell_base = d_base; A.gcap = d_gcap; A.meta = g.d_meta; A.cell_start = g.d_cell_start; A.cursor = g.d_cursor;
  A.sorted_idx = g.d_sorted_idx; A.sorted_xy = g.d_sorted_xy; A.nrm = cs->d_nrm; A.sorted_nrm = g.d_sorted_nrm;
  A.big_threshold = ctx->grid_big_threshold;
  hipLaunchKernelGGL(k_grid_build, dim3((unsigned) nc), dim3(1024), 0, ctx->stream, A);
  HIPCHK(ctx, hipGetLastError());
  // map-sized clouds: k_grid_build left them with their meta only; the counting sort runs over the whole chip, one cloud after the other
  for (int c = 0; c < nc; ++c) {
    if (cs->h_count[c] < ctx->grid_big_threshold) continue;
    const int64_t cells_max = (int64_t) gcap[c] * gcap[c] + 1;
    const int n_tiles = (int) ((cells_max + kGridTile - 1) / kGridTile);            // <= 1025: gcap <= 2048
    HIPCHK(ctx, hipMemsetAsync(g.d_cursor + cell_base[c], 0, sizeof(int32_t) * (size_t) cells_max, ctx->stream));
    GridBigArgs B;
    B.xy = cs->d_xy; B.start = cs->d_start; B.count = cs->d_count; B.cloud = c; B.meta = g.d_meta; B.cell_start = g.d_cell_start; B.cursor = g.d_cursor;
    B.tile_sums = d_tiles; B.sorted_idx = g.d_sorted_idx; B.sorted_xy = g.d_sorted_xy; B.nrm = cs->d_nrm; B.sorted_nrm = g.d_sorted_nrm;
    int pb = (cs->h_count[c] + 1023) / 1024; pb = pb < 1 ? 1 : (pb > 2048 ? 2048 : pb);
    hipLaunchKernelGGL(k_grid_big_hist, dim3((unsigned) pb), dim3(256), 0, ctx->stream, B);
    hipLaunchKernelGGL(k_grid_big_scan<0>, dim3((unsigned) n_tiles), dim3(1024), 0, ctx->stream, B);
    hipLaunchKernelGGL(k_grid_big_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, B.tile_sums, n_tiles);
    hipLaunchKernelGGL(k_grid_big_scan<1>, dim3((unsigned) n_tiles), dim3(1024), 0, ctx->stream, B);
    hipLaunchKernelGGL(k_grid_big_scatter, dim3((unsigned) pb), dim3(256), 0, ctx->stream, B);
    HIPCHK(ctx, hipGetLastError());
  }
  HIPCHK(ctx, stream_sync(ctx));      // host vectors above must outlive the copies
  g.d_block = t_block.release();      // owned by the cache from here on
  cs->grids.push_back(g);
  *out = GridDev{g.d_meta, g.d_cell_start, g.d_sorted_idx, g.d_sorted_xy, g.d_sorted_nrm};
  return LSM2D_SUCCESS;
}

// KD-tree finder: CorrespondenceFinderKDTree2D::reset() (registration/correspondence_finder_kd_tree_2d.cpp:31-38) for every cloud of the
// (fixed) set, cached per (max_leaf_range, min_leaf_points).  Built on the device, level by level (k_kd_level: one wave per node); the
// host only learns, after each level, how many nodes the next one has.
// ---- the latency form of the KD-tree build (k_kd_build_scan): a handful of scan-sized clouds, working set in LDS.  Nothing is uploaded or cleared ahead of
// the kernel (it writes the set's KdMeta itself), and the fixed sets of an aligner call's KD-tree slices -- the live tracker: one scan per laser -- are built
// side by side by ONE launch with ONE wait for the trees' sizes (kd_scan_launch).
static constexpr int kKdScanCap = 1280;      // points per cloud the LDS layout is sized for (58 KB)
struct KdScanPrep {
  const lsm2d_cloudset* cs = nullptr; KdCache kc; DevTmp block; KdBuildScanArgs W; std::vector<KdMeta> h_meta;
};
static void kd_defaults(float& max_leaf_range, int& min_leaf_points) {      // the class defaults (correspondence_finder_kd_tree_2d.h:26-33), as the oracle applies them
  if (!(max_leaf_range > 0.0f)) max_leaf_range = 1e-2f;
  if (min_leaf_points <= 0) min_leaf_points = 20;
}
static const KdCache* kd_cached(const lsm2d_cloudset* cs, float max_leaf_range, int min_leaf_points) {
  for (const auto& k : cs->kds) if (k.valid && k.max_leaf_range == max_leaf_range && k.min_leaf_points == min_leaf_points) return &k;
  return nullptr;
}
static bool kd_scan_eligible(const lsm2d_context* ctx, const lsm2d_cloudset* cs) {
  const int nc = cs->n_clouds;
  if (nc < 1 || nc > ctx->kd_scan_max_clouds || nc > 8 || cs->count_pending || (int) kd_scan_lds_bytes(kKdScanCap) > ctx->max_dyn_lds) return false;
  for (int c = 0; c < nc; ++c) if (cs->h_count[c] > kKdScanCap || cs->h_count[c] > ctx->kd_wg_max_points) return false;
  return true;
}
// the set's block (recycled for a reserved single-cloud set) and the kernel's arguments; no GPU work
static int kd_scan_prepare(lsm2d_context* ctx, const lsm2d_cloudset* cs, float max_leaf_range, int min_leaf_points, KdScanPrep& P) {
  const int nc = cs->n_clouds;
  P.cs = cs; P.kc = KdCache(); P.kc.max_leaf_range = max_leaf_range; P.kc.min_leaf_points = min_leaf_points;
  long long nodes = 0;
  for (int c = 0; c < nc; ++c) {
    P.W.node_base[c] = (int32_t) nodes;
    const long long room = cs->capacity > 0 ? cs->capacity : cs->h_count[c];      // (a reserved set: for whatever it may hold later -- its allocation is recycled)
    nodes += 2ll * room > 2 ? 2ll * room : 2;
  }
  const size_t np = (size_t) cs->padded_total;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t) 255; return o; };
  const size_t o_meta = take(sizeof(KdMeta) * (size_t) nc), o_nodes = take(sizeof(KdNode) * (size_t) nodes);
  const size_t o_lxy = take(sizeof(float2) * np), o_lidx = take(sizeof(int32_t) * np), o_lnr = take(sizeof(float2) * np);
  const bool recycle = cs->kds.size() == 1 && !cs->kds[0].valid && cs->kds[0].d_block && cs->kds[0].block_bytes >= off;
  P.kc.block_bytes = off;
  if (recycle) { P.block.p = cs->kds[0].d_block; P.kc.block_bytes = cs->kds[0].block_bytes; cs->kds.clear(); }      // (owned by the guard again until the build has succeeded)
  else {
    if (cs->kds.size() == 1 && !cs->kds[0].valid) { if (cs->kds[0].d_block) (void) hipFree(cs->kds[0].d_block); cs->kds.clear(); }
    HIPCHK(ctx, hipMalloc(&P.block.p, off));
  }
  char* blk = (char*) P.block.p;
  KdCache& kc = P.kc;
  kc.d_meta = (KdMeta*) (blk + o_meta); kc.d_nodes = (KdNode*) (blk + o_nodes);
  kc.d_leaf_xy = (float2*) (blk + o_lxy); kc.d_leaf_idx = (int32_t*) (blk + o_lidx); kc.d_leaf_nrm = (float2*) (blk + o_lnr);
  KdBuildArgs B;
  B.start = cs->d_start; B.meta = kc.d_meta; B.nodes = kc.d_nodes; B.n_nodes = nullptr;
  B.leaf_xy = kc.d_leaf_xy; B.leaf_idx = kc.d_leaf_idx; B.max_leaf_range = max_leaf_range; B.min_leaf_points = min_leaf_points;
  B.xy_in = nullptr; B.idx_in = nullptr; B.xy_out = nullptr; B.idx_out = nullptr; B.q_in = nullptr; B.q_out = nullptr; B.q_out_count = nullptr; B.n_items = 0; B.n_items_ptr = nullptr; B.local_io = 0; B.io_base = 0; B.io_node_base = 0;
  P.W.B = B; P.W.count = cs->d_count; P.W.xy0 = cs->d_xy; P.W.nrm0 = cs->d_nrm; P.W.leaf_nrm = kc.d_leaf_nrm; P.W.meta_rw = kc.d_meta; P.W.cap = kKdScanCap; P.W.n_clouds = nc;
  P.h_meta.resize((size_t) nc);
  return LSM2D_SUCCESS;
}
// one launch for all prepared sets, one wait for their trees' sizes; the caches are published to their sets
static int kd_scan_launch(lsm2d_context* ctx, KdScanPrep* P, int count) {
  if (count <= 0) return LSM2D_SUCCESS;
  const size_t lds = kd_scan_lds_bytes(kKdScanCap);
  if (count == 1) {
    if (ctx->kd_chain == 1) hipLaunchKernelGGL(k_kd_build_scan<1>, dim3((unsigned) P[0].W.n_clouds), dim3(kKdScanThreads), lds, ctx->stream, P[0].W);
    else hipLaunchKernelGGL(k_kd_build_scan<0>, dim3((unsigned) P[0].W.n_clouds), dim3(kKdScanThreads), lds, ctx->stream, P[0].W);
  } else {
    KdBuildScanMulti M; int widest = 1;
    for (int i = 0; i < kMaxSlices; ++i) { M.w[i] = P[i < count ? i : 0].W; if (i >= count) M.w[i].n_clouds = 0; }
    for (int i = 0; i < count; ++i) if (P[i].W.n_clouds > widest) widest = P[i].W.n_clouds;
    if (ctx->kd_chain == 1) hipLaunchKernelGGL(k_kd_build_scan_multi<1>, dim3((unsigned) widest, (unsigned) count), dim3(kKdScanThreads), lds, ctx->stream, M);
    else hipLaunchKernelGGL(k_kd_build_scan_multi<0>, dim3((unsigned) widest, (unsigned) count), dim3(kKdScanThreads), lds, ctx->stream, M);
  }
  HIPCHK(ctx, hipGetLastError());
  for (int i = 0; i < count; ++i)
    HIPCHK(ctx, hipMemcpyAsync(P[i].h_meta.data(), P[i].kc.d_meta, sizeof(KdMeta) * P[i].h_meta.size(), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, stream_sync(ctx));
  for (int i = 0; i < count; ++i) {
    KdCache& kc = P[i].kc;
    kc.levels = 0; kc.total_nodes = 0; kc.max_nodes_per_cloud = 0;
    for (const KdMeta& m : P[i].h_meta) { kc.total_nodes += m.n_nodes; if (m.n_nodes > kc.max_nodes_per_cloud) kc.max_nodes_per_cloud = m.n_nodes; if (m.pad0 > kc.levels) kc.levels = m.pad0; }
    ctx->last_kd_levels = kc.levels; ctx->last_kd_nodes = kc.total_nodes;
    kc.d_block = P[i].block.release();      // owned by the cache from here on
    P[i].cs->kds.push_back(kc);
  }
  return LSM2D_SUCCESS;
}

static int ensure_kdtree(lsm2d_context* ctx, const lsm2d_cloudset* cs, float max_leaf_range, int min_leaf_points, KdDev* out, const KdCache** out_cache = nullptr) {
  kd_defaults(max_leaf_range, min_leaf_points);
  if (const KdCache* k = kd_cached(cs, max_leaf_range, min_leaf_points)) {
    *out = KdDev{k->d_meta, k->d_nodes, k->d_leaf_xy, k->d_leaf_idx, k->d_leaf_nrm}; if (out_cache) *out_cache = k; return LSM2D_SUCCESS;
  }
  if (kd_scan_eligible(ctx, cs)) {      // a handful of scan-sized clouds: the latency form
    KdScanPrep P;
    int rc = kd_scan_prepare(ctx, cs, max_leaf_range, min_leaf_points, P); if (rc) return rc;
    rc = kd_scan_launch(ctx, &P, 1); if (rc) return rc;
    const KdCache& k = cs->kds.back();
    *out = KdDev{k.d_meta, k.d_nodes, k.d_leaf_xy, k.d_leaf_idx, k.d_leaf_nrm}; if (out_cache) *out_cache = &k;
    return LSM2D_SUCCESS;
  }
  const int nc = cs->n_clouds;
  // a cloud of n points has at most 2 n - 1 nodes (every split leaves both children non-empty); an empty cloud still has its root
  std::vector<KdMeta> meta((size_t) nc);
  long long nodes = 0;
  for (int c = 0; c < nc; ++c) {
    meta[(size_t) c].node_base = (int32_t) nodes; meta[(size_t) c].n_nodes = 0; meta[(size_t) c].pad0 = meta[(size_t) c].pad1 = 0;
    const long long room = cs->capacity > 0 ? cs->capacity : cs->h_count[c];      // (a reserved set: for whatever it may hold later -- its allocation is recycled)
    nodes += 2ll * room > 2 ? 2ll * room : 2;
    if (nodes > 0x7ffffff0ll) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "kdtree: too many nodes");
  }
  const size_t np = (size_t) cs->padded_total, nq = np / 2 + (size_t) nc + 2;
  constexpr int kMaxLevels = 8192;
  KdCache kc; kc.max_leaf_range = max_leaf_range; kc.min_leaf_points = min_leaf_points;
  DevTmp t_block;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t) 255; return o; };
  const size_t o_meta = take(sizeof(KdMeta) * (size_t) nc), o_nodes = take(sizeof(KdNode) * (size_t) nodes);
  const size_t o_lxy = take(sizeof(float2) * np), o_lidx = take(sizeof(int32_t) * np), o_lnr = take(sizeof(float2) * np);
  const bool recycle = cs->kds.size() == 1 && !cs->kds[0].valid && cs->kds[0].d_block && cs->kds[0].block_bytes >= off;
  kc.block_bytes = off;
  if (recycle) { t_block.p = cs->kds[0].d_block; kc.block_bytes = cs->kds[0].block_bytes; cs->kds.clear(); }      // (owned by the guard again until the build has succeeded)
  else {
    if (cs->kds.size() == 1 && !cs->kds[0].valid) { if (cs->kds[0].d_block) (void) hipFree(cs->kds[0].d_block); cs->kds.clear(); }
    HIPCHK(ctx, hipMalloc(&t_block.p, off));
  }
  char* blk = (char*) t_block.p;
  kc.d_meta = (KdMeta*) (blk + o_meta); kc.d_nodes = (KdNode*) (blk + o_nodes);
  kc.d_leaf_xy = (float2*) (blk + o_lxy); kc.d_leaf_idx = (int32_t*) (blk + o_lidx); kc.d_leaf_nrm = (float2*) (blk + o_lnr);
  // working set of the build: two ping-pong copies of (xy, idx), two queues, per-cloud node counters, one queue counter per level
  off = 0;
  const size_t w_xy0 = take(sizeof(float2) * np), w_xy1 = take(sizeof(float2) * np), w_ix0 = take(sizeof(int32_t) * np), w_ix1 = take(sizeof(int32_t) * np);
  const size_t w_q0 = take(sizeof(int4) * nq), w_q1 = take(sizeof(int4) * nq), w_nn = take(sizeof(int32_t) * (size_t) nc), w_cnt = take(sizeof(int32_t) * (size_t) (kMaxLevels + 1));
  if (off > ctx->d_kd_work_bytes) {
    if (ctx->d_kd_work) { HIPCHK(ctx, stream_sync(ctx)); HIPCHK(ctx, hipFree(ctx->d_kd_work)); ctx->d_kd_work = nullptr; ctx->d_kd_work_bytes = 0; }
    HIPCHK(ctx, hipMalloc(&ctx->d_kd_work, off + off / 2));
    ctx->d_kd_work_bytes = off + off / 2;
  }
  char* wk = (char*) ctx->d_kd_work;
  float2* xyb[2] = {(float2*) (wk + w_xy0), (float2*) (wk + w_xy1)}; int32_t* ixb[2] = {(int32_t*) (wk + w_ix0), (int32_t*) (wk + w_ix1)};
  int4* qb[2] = {(int4*) (wk + w_q0), (int4*) (wk + w_q1)};
  int32_t* d_nn = (int32_t*) (wk + w_nn); int32_t* d_cnt = (int32_t*) (wk + w_cnt);
  HIPCHK(ctx, hipMemcpyAsync(kc.d_meta, meta.data(), sizeof(KdMeta) * (size_t) nc, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * (size_t) (kMaxLevels + 1), ctx->stream));
  KdBuildArgs B;
  B.start = cs->d_start; B.meta = kc.d_meta; B.nodes = kc.d_nodes; B.n_nodes = d_nn;
  B.leaf_xy = kc.d_leaf_xy; B.leaf_idx = kc.d_leaf_idx; B.max_leaf_range = max_leaf_range; B.min_leaf_points = min_leaf_points;
  B.xy_in = nullptr; B.idx_in = nullptr; B.xy_out = nullptr; B.idx_out = nullptr; B.q_in = nullptr; B.q_out = nullptr; B.q_out_count = nullptr; B.n_items = 0; B.n_items_ptr = nullptr; B.local_io = 0; B.io_base = 0; B.io_node_base = 0;
  // Scan-sized clouds: ONE launch builds every such tree, a workgroup per cloud walking its levels itself (k_kd_build_wg) -- no host round trip per level.
  // Map-sized clouds keep the level loop (a level of theirs fills the chip): their roots are queued here, the host learns every level's node count.
  const int wg_max = ctx->kd_wg_max_points;
  std::vector<int4> roots; std::vector<int32_t> ones((size_t) nc, 1);
  int n_small = 0;
  for (int c = 0; c < nc; ++c) { if (cs->h_count[c] <= wg_max) ++n_small; else roots.push_back(make_int4(c, 0, 0, cs->h_count[c])); }
  HIPCHK(ctx, hipMemcpyAsync(d_nn, ones.data(), sizeof(int32_t) * (size_t) nc, hipMemcpyHostToDevice, ctx->stream));
  if (n_small > 0) {
    KdBuildWgArgs W; W.B = B; W.count = cs->d_count; W.xy0 = cs->d_xy; W.nrm0 = cs->d_nrm;
    W.xy_buf[0] = xyb[0]; W.xy_buf[1] = xyb[1]; W.idx_buf[0] = ixb[0]; W.idx_buf[1] = ixb[1]; W.q_buf[0] = qb[0]; W.q_buf[1] = qb[1];
    W.leaf_nrm = kc.d_leaf_nrm; W.meta_rw = kc.d_meta; W.max_points = wg_max;
    if (ctx->kd_chain == 1) hipLaunchKernelGGL(k_kd_build_wg<1>, dim3((unsigned) nc), dim3(256), 0, ctx->stream, W);
    else hipLaunchKernelGGL(k_kd_build_wg<0>, dim3((unsigned) nc), dim3(256), 0, ctx->stream, W);
    HIPCHK(ctx, hipGetLastError());
  }
  long long n_items = (long long) roots.size(); int level = 0;
  std::vector<KdMeta> h_meta((size_t) nc);
  if (n_items > 0) {
    // (the big clouds' items use the queue buffers from position 0: the small clouds' stretches -- start / 2 + c -- are theirs alone only while the
    // workgroup build runs, and the stream orders the two)
    HIPCHK(ctx, hipMemcpyAsync(qb[0], roots.data(), sizeof(int4) * roots.size(), hipMemcpyHostToDevice, ctx->stream));
    // Levels are launched back to back with an UPPER BOUND of their item count (a level cannot hold more items than roots x 2^level, nor more than
    // there are groups of max(min_leaf_points, 2) points); the kernel reads the true count where the previous level left it, waves beyond it leave at
    // once, a level with no items does nothing.  The host reads the counts after as many levels as a balanced tree would have, then every four levels.
    long long big_points = 0; for (const int4& r : roots) big_points += r.w;
    const long long per_item = min_leaf_points > 2 ? min_leaf_points : 2;
    const long long cap_items = big_points / per_item + (long long) roots.size();
    const int32_t n_roots = (int32_t) roots.size();
    HIPCHK(ctx, hipMemcpyAsync(d_cnt, &n_roots, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));      // (pageable source: copied before the call returns)
    int expect = 4; for (long long m = cap_items; m > 1; m >>= 1) ++expect;      // (a 100k-point map with 20-point leaves: 16 launches for its 15 levels)
    std::vector<int32_t> h_counts;
    bool more = true;
    while (more) {
      const int until = level == 0 ? expect : level + 4;
      for (; level < until; ++level) {
        if (level >= kMaxLevels) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "kdtree: deeper than 8192 levels");
        B.xy_in = level == 0 ? cs->d_xy : xyb[level & 1]; B.idx_in = level == 0 ? nullptr : ixb[level & 1];
        B.xy_out = xyb[(level + 1) & 1]; B.idx_out = ixb[(level + 1) & 1];
        B.q_in = qb[level & 1]; B.q_out = qb[(level + 1) & 1]; B.q_out_count = d_cnt + level + 1; B.n_items = 0; B.n_items_ptr = d_cnt + level;
        long long bound = level < 40 ? ((long long) n_roots << level) : cap_items; if (bound > cap_items) bound = cap_items;
        // the top levels of a map-sized cloud -- while an evenly split node would still hold "kd_wide_min_points" points --: a workgroup per node (kd_node_wide)
        const bool wide = ctx->kd_wide_min_points > 0 && level < 40 && (big_points / n_roots) >> level >= ctx->kd_wide_min_points;
        if (wide) {
          if (ctx->kd_chain == 1) hipLaunchKernelGGL(k_kd_level_wide<1>, dim3((unsigned) bound), dim3(256), 0, ctx->stream, B);
          else hipLaunchKernelGGL(k_kd_level_wide<0>, dim3((unsigned) bound), dim3(256), 0, ctx->stream, B);
        } else {
          const unsigned blocks = (unsigned) ((bound + 3) / 4);
          if (ctx->kd_chain == 1) hipLaunchKernelGGL(k_kd_level<1>, dim3(blocks), dim3(256), 0, ctx->stream, B);
          else hipLaunchKernelGGL(k_kd_level<0>, dim3(blocks), dim3(256), 0, ctx->stream, B);
        }
      }
      // the big clouds' sizes and leaf-order normals (the workgroup build wrote its own clouds') go out with every batch of levels -- they are right as soon
      // as the last level has run, which is nearly always the first batch -- and the counts and the sizes come back with ONE wait
      hipLaunchKernelGGL(k_kd_finish, dim3((unsigned) ((nc + 255) / 256)), dim3(256), 0, ctx->stream, (const int32_t*) d_nn, nc, kc.d_meta);
      for (int c = 0; c < nc; ++c) {
        if (cs->h_count[c] <= wg_max) continue;
        int gx = (cs->h_count[c] + 255) / 256; if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(k_kd_permute_normals, dim3((unsigned) gx, 1u), dim3(256), 0, ctx->stream, (const float2*) cs->d_nrm,
                           (const int32_t*) cs->d_start, (const int32_t*) cs->d_count, (const int32_t*) kc.d_leaf_idx, kc.d_leaf_nrm, c);
      }
      HIPCHK(ctx, hipGetLastError());
      h_counts.resize((size_t) level + 1);
      HIPCHK(ctx, hipMemcpyAsync(h_counts.data(), d_cnt, sizeof(int32_t) * ((size_t) level + 1), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(h_meta.data(), kc.d_meta, sizeof(KdMeta) * (size_t) nc, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, stream_sync(ctx));
      more = h_counts[(size_t) level] > 0;
    }
    level = 0; while ((size_t) level < h_counts.size() && h_counts[(size_t) level] > 0) ++level;      // levels that held items: what the level-by-level loop counted
  }
  else {      // what the host needs of the result: every tree's node count (and, of the workgroup builds, the depth): one read, one wait
    HIPCHK(ctx, hipMemcpyAsync(h_meta.data(), kc.d_meta, sizeof(KdMeta) * (size_t) nc, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, stream_sync(ctx));
  }
  kc.levels = level; kc.total_nodes = 0; kc.max_nodes_per_cloud = 0;
  for (int c = 0; c < nc; ++c) {
    const int nn = h_meta[(size_t) c].n_nodes;
    kc.total_nodes += nn; if (nn > kc.max_nodes_per_cloud) kc.max_nodes_per_cloud = nn;
    if (cs->h_count[c] <= wg_max && h_meta[(size_t) c].pad0 > kc.levels) kc.levels = h_meta[(size_t) c].pad0;
  }
  ctx->last_kd_levels = kc.levels; ctx->last_kd_nodes = kc.total_nodes;
  kc.d_block = t_block.release();      // owned by the cache from here on (the working set goes with its guard)
  cs->kds.push_back(kc);
  *out = KdDev{kc.d_meta, kc.d_nodes, kc.d_leaf_xy, kc.d_leaf_idx, kc.d_leaf_nrm};
  if (out_cache) *out_cache = &cs->kds.back();
  return LSM2D_SUCCESS;
}

static CloudDev cloud_dev_with_tiles(CloudDev c, const lsm2d_cloudset* cs) { c.tile_bounds = cs->d_tile_bounds; c.tile_start = cs->d_tile_start; return c; }

// lane-chunked copy of every cloud for k_align's projective streaming pass (project_cloud_lanes in lsm2d_device.h), with the bounding circles of
// every thread's chunk and of every block of it (the exact culling).  Everything is built into guarded temporaries and PUBLISHED to the set only
// after the last launch has finished (round-3 advisor: a failure half-way used to leave d_lane_xy set and never filled -- the next call streamed
// uninitialised memory).
static int ensure_lane_layout(lsm2d_context* ctx, const lsm2d_cloudset* cs) {
  if (cs->d_lane_xy || cs->count_pending) return LSM2D_SUCCESS;     // a size-pending set is a clipped scene: small, and building needs a sync
  const int nc = cs->n_clouds;
  bool any_big = false;                       // every cloud <= one pair per thread: the plain layout already is lane-chunked
  for (int c = 0; c < nc; ++c) if (((long long) cs->h_count[c] + 1) / 2 > kAlignBlock) { any_big = true; break; }
  if (!any_big) return LSM2D_SUCCESS;
  // project_cloud_lanes walks a cloud's rows with a 32-bit scalar byte offset: a cloud beyond 2^31 bytes of slots
  // (> 2.6e8 points) keeps the plain layout and project_cloud
  for (int c = 0; c < nc; ++c) if ((((long long) cs->h_count[c] + 1) / 2 + kAlignBlock) * (long long) sizeof(float4) >= (1ll << 31)) return LSM2D_SUCCESS;
  std::vector<long long> lstart((size_t) nc); std::vector<int32_t> lT((size_t) nc);
  long long slots = 0; int maxT = 1;
  for (int c = 0; c < nc; ++c) {
    const long long npairs = ((long long) cs->h_count[c] + 1) / 2;
    const int T = (int) ((npairs + kAlignBlock - 1) / kAlignBlock);
    lstart[c] = slots; lT[c] = T; slots += (long long) T * kAlignBlock; if (T > maxT) maxT = T;
  }
  if (slots == 0) slots = 1;
  slots += 2 * kAlignBlock;      // two spare rows behind the last cloud: project_cloud_units' look-ahead load may read one row past a cloud's last
  // block circles: 57 KB per cloud.  They are what the kept unit lists are built from -- worth it for a map or a few thousand big clouds, not for tens of
  // thousands of scan-sized moving clouds (round-4 advisor: several GB there): beyond 256 MB the set goes without them and its batches run the chunk-level
  // stream of the shared instantiation (proj_culled_for_all needs block_bounds)
  const int nbs = cull_blocks_for(maxT);      // 7 blocks per chunk, 14 when the set holds a map-sized cloud (lsm2d_kernels.h)
  const bool want_blocks = sizeof(float4) * (size_t) nc * nbs * kAlignBlock <= ((size_t) 256 << 20);
  DevTmp t_xy, t_bounds, t_blocks, t_start, t_T;
  HIPCHK(ctx, hipMalloc(&t_xy.p, sizeof(float4) * (size_t) slots));
  HIPCHK(ctx, hipMalloc(&t_bounds.p, sizeof(float4) * (size_t) nc * kAlignBlock));
  if (want_blocks) HIPCHK(ctx, hipMalloc(&t_blocks.p, sizeof(float4) * (size_t) nc * nbs * kAlignBlock));
  HIPCHK(ctx, hipMalloc(&t_start.p, sizeof(long long) * (size_t) nc));
  HIPCHK(ctx, hipMalloc(&t_T.p, sizeof(int32_t) * (size_t) nc));
  float4* d_xy = (float4*) t_xy.p; float4* d_bounds = (float4*) t_bounds.p; float4* d_blocks = (float4*) t_blocks.p;
  long long* d_start = (long long*) t_start.p; int32_t* d_T = (int32_t*) t_T.p;
  HIPCHK(ctx, hipMemsetAsync(d_xy + (slots - 2 * kAlignBlock), 0x7f, sizeof(float4) * 2 * kAlignBlock, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_start, lstart.data(), sizeof(long long) * (size_t) nc, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_T, lT.data(), sizeof(int32_t) * (size_t) nc, hipMemcpyHostToDevice, ctx->stream));
  long long per = (long long) maxT * kAlignBlock; int gx = (int) ((per + 255) / 256); if (gx > 2048) gx = 2048; if (gx < 1) gx = 1;
  for (int c0 = 0; c0 < nc; c0 += 32768) {         // gridDim.y is limited to 65535
    const int ny = nc - c0 < 32768 ? nc - c0 : 32768;
    hipLaunchKernelGGL(k_lane_layout, dim3((unsigned) gx, (unsigned) ny), dim3(256), 0, ctx->stream, (const float2*) cs->d_xy,
                       (const int32_t*) cs->d_start, (const int32_t*) cs->d_count, (const long long*) d_start,
                       (const int32_t*) d_T, (int) kAlignBlock, d_xy, c0);
    hipLaunchKernelGGL(k_lane_bounds, dim3((unsigned) (kAlignBlock / 4), (unsigned) ny), dim3(256), 0, ctx->stream, (const float2*) cs->d_xy,
                       (const int32_t*) cs->d_start, (const int32_t*) cs->d_count, (const int32_t*) d_T, (int) kAlignBlock, d_bounds, c0);
    if (want_blocks)
      hipLaunchKernelGGL(k_block_bounds, dim3((unsigned) (kAlignBlock * nbs / 4), (unsigned) ny), dim3(256), 0, ctx->stream, (const float2*) cs->d_xy,
                         (const int32_t*) cs->d_start, (const int32_t*) cs->d_count, (const int32_t*) d_T, (int) kAlignBlock, d_blocks, c0, nbs);
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, stream_sync(ctx));        // the host vectors above back the async copies; and only a finished build is published
  cs->d_lane_xy = (float4*) t_xy.release(); cs->d_lane_bounds = (float4*) t_bounds.release(); cs->d_block_bounds = (float4*) t_blocks.release();
  cs->d_lane_start = (long long*) t_start.release(); cs->d_lane_T = (int32_t*) t_T.release();
  cs->block_stride = nbs;
  return LSM2D_SUCCESS;
}

// (x, y, nx, ny) rows of the whole set: k_align's bin walk gathers a z-buffer winner's point and normal as one 16-byte row
static int ensure_aos(lsm2d_context* ctx, const lsm2d_cloudset* cs) {
  if (cs->d_aos || cs->count_pending || cs->unpack_pending || cs->prep_pending) return LSM2D_SUCCESS;      // (a set still changing on the device is gathered from its split arrays)
  DevTmp t;
  HIPCHK(ctx, hipMalloc(&t.p, sizeof(float4) * (size_t) cs->padded_total));
  long long blocks = (cs->padded_total + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_aos_rows, dim3((unsigned) blocks), dim3(256), 0, ctx->stream, (const float2*) cs->d_xy, (const float2*) cs->d_nrm, (long long) cs->padded_total, (float4*) t.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, stream_sync(ctx));
  cs->d_aos = (float4*) t.release();
  return LSM2D_SUCCESS;
}

// bounding circles of every cloud's tiles of 64 consecutive points: what k_align's point-query branch tests before it runs a tile's queries
static int ensure_tile_bounds(lsm2d_context* ctx, const lsm2d_cloudset* cs) {
  if (cs->d_tile_bounds || cs->count_pending) return LSM2D_SUCCESS;
  const int nc = cs->n_clouds;
  std::vector<int32_t> tstart((size_t) nc);
  long long tiles = 0; int max_tiles = 1;
  for (int c = 0; c < nc; ++c) {
    const int t = (cs->h_count[c] + 63) / 64;
    tstart[c] = (int32_t) tiles; tiles += t; if (t > max_tiles) max_tiles = t;
    if (tiles > 0x7fffffff) return LSM2D_SUCCESS;          // (no culling for such a set)
  }
  DevTmp t_bounds, t_start;      // published only when complete (see ensure_lane_layout)
  HIPCHK(ctx, hipMalloc(&t_bounds.p, sizeof(float4) * (size_t) (tiles > 0 ? tiles : 1)));
  HIPCHK(ctx, hipMalloc(&t_start.p, sizeof(int32_t) * (size_t) nc));
  HIPCHK(ctx, hipMemcpyAsync(t_start.p, tstart.data(), sizeof(int32_t) * (size_t) nc, hipMemcpyHostToDevice, ctx->stream));
  int gx = (max_tiles + 3) / 4; if (gx > 4096) gx = 4096; if (gx < 1) gx = 1;
  for (int c0 = 0; c0 < nc; c0 += 32768) {
    const int ny = nc - c0 < 32768 ? nc - c0 : 32768;
    hipLaunchKernelGGL(k_tile_bounds, dim3((unsigned) gx, (unsigned) ny), dim3(256), 0, ctx->stream, (const float2*) cs->d_xy, (const int32_t*) cs->d_start,
                       (const int32_t*) cs->d_count, (const int32_t*) t_start.p, (float4*) t_bounds.p, c0);
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, stream_sync(ctx));        // the host vector above backs the async copy
  cs->d_tile_bounds = (float4*) t_bounds.release(); cs->d_tile_start = (int32_t*) t_start.release();
  return LSM2D_SUCCESS;
}

// Distance-map finder: CorrespondenceFinderNN2D::reset() (registration/correspondence_finder_nn_2d.cpp:10-52,84-97) for every
// cloud of the (fixed) set, cached per (max_distance, resolution).
static int ensure_distmap(lsm2d_context* ctx, const lsm2d_cloudset* cs, float max_distance, float resolution, DistDev* out) {
  for (const auto& d : cs->dists)
    if (d.max_distance == max_distance && d.resolution == resolution) { *out = DistDev{d.d_meta, d.d_parent}; return LSM2D_SUCCESS; }
  if (!(resolution > 0.0f) || max_distance < 0.0f) return fail(ctx, LSM2D_BAD_ARGUMENT, "distmap: resolution must be > 0 and max_distance >= 0");
  const int nc = cs->n_clouds;
  DevTmp t_bbox;
  HIPCHK(ctx, hipMalloc(&t_bbox.p, sizeof(float4) * (size_t) nc));
  float4* d_bbox = (float4*) t_bbox.p;
  hipLaunchKernelGGL(k_cloud_bbox, dim3((unsigned) nc), dim3(256), 0, ctx->stream, (const float2*) cs->d_xy, (const int32_t*) cs->d_start,
                     (const int32_t*) cs->d_count, d_bbox);
  HIPCHK(ctx, hipGetLastError());
  std::vector<float4> bbox((size_t) nc);
  HIPCHK(ctx, hipMemcpyAsync(bbox.data(), d_bbox, sizeof(float4) * (size_t) nc, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, stream_sync(ctx));
  const float inv_res = 1.0f / resolution;
  const float mds_px = max_distance * max_distance * inv_res * inv_res;
  const int padding = (int) (sqrtf(mds_px) + 75.5f);
  const int R = (int) floor(sqrt((double) mds_px));
  std::vector<DistMeta> meta((size_t) nc);
  long long total = 0; int max_rows_cols = 0;
  for (int c = 0; c < nc; ++c) {
    const float lx = bbox[c].x, ly = bbox[c].y, ux = bbox[c].z, uy = bbox[c].w;
    const double rows = ceil((double) ((ux - lx) * inv_res + (float) padding)), cols = ceil((double) ((uy - ly) * inv_res + (float) padding));
    if (!(rows >= 1 && cols >= 1) || rows * cols > 400e6) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "distmap: grid too large");
    DistMeta& m = meta[c];
    m.lx = lx; m.ly = ly; m.inv_res = inv_res; m.half_pad = (float) padding * 0.5f; m.rows = (int) rows; m.cols = (int) cols; m.base = total;
    total += (long long) m.rows * m.cols;
    if (m.rows * (long long) m.cols > max_rows_cols) max_rows_cols = (int) (m.rows * (long long) m.cols > 0x7fffffff ? 0x7fffffff : m.rows * (long long) m.cols);
    if (total > (1ll << 33)) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "distmap: more than 8 Gi pixels in one set");
  }
  // scatter build (every point stamps its disc, k_distmap_stamp) whenever (d2, index) packs into 31 bits; else the gather build
  int max_pts = 1; for (int c = 0; c < nc; ++c) if (cs->h_count[c] > max_pts) max_pts = cs->h_count[c];
  int gbits = 1; while (gbits < 31 && (1ll << gbits) < (long long) max_pts) ++gbits;
  const bool scatter = ctx->distmap_build != 1 && R <= 511 && (((long long) R * R + 1) << gbits) <= (1ll << 31);
  for (auto& m : meta) { m.gbits = scatter ? gbits : 31; m.gmask = scatter ? (int32_t) ((1u << gbits) - 1u) : 0x7fffffff; }
  DistCache d; d.max_distance = max_distance; d.resolution = resolution;
  DevTmp t_dmeta, t_parent, t_goal;
  HIPCHK(ctx, hipMalloc(&t_dmeta.p, sizeof(DistMeta) * (size_t) nc));
  HIPCHK(ctx, hipMalloc(&t_parent.p, sizeof(int32_t) * (size_t) total));
  d.d_meta = (DistMeta*) t_dmeta.p; d.d_parent = (int32_t*) t_parent.p;
  HIPCHK(ctx, hipMemcpyAsync(d.d_meta, meta.data(), sizeof(DistMeta) * (size_t) nc, hipMemcpyHostToDevice, ctx->stream));
  if (scatter) {
    HIPCHK(ctx, hipMemsetAsync(d.d_parent, 0xff, sizeof(int32_t) * (size_t) total, ctx->stream));
    int sb = (max_pts + 3) / 4; if (sb > 16384) sb = 16384;
    for (int c0 = 0; c0 < nc; c0 += 32768) {         // gridDim.y is limited to 65535
      const int ny = nc - c0 < 32768 ? nc - c0 : 32768;
      hipLaunchKernelGGL(k_distmap_stamp, dim3((unsigned) sb, (unsigned) ny), dim3(256), 0, ctx->stream, (const float2*) cs->d_xy,
                         (const int32_t*) cs->d_start, (const int32_t*) cs->d_count, (const DistMeta*) d.d_meta, (uint32_t*) d.d_parent, mds_px, R, c0);
    }
  } else {
    HIPCHK(ctx, hipMalloc(&t_goal.p, sizeof(int32_t) * (size_t) total));
    int32_t* d_cellgoal = (int32_t*) t_goal.p;
    HIPCHK(ctx, hipMemsetAsync(d_cellgoal, 0x7f, sizeof(int32_t) * (size_t) total, ctx->stream));
    int gb = (max_pts + 255) / 256; if (gb > 1024) gb = 1024;
    int fb = (max_rows_cols + 255) / 256; if (fb > 4096) fb = 4096; if (fb < 1) fb = 1;
    for (int c0 = 0; c0 < nc; c0 += 32768) {
      const int ny = nc - c0 < 32768 ? nc - c0 : 32768;
      hipLaunchKernelGGL(k_distmap_goals, dim3((unsigned) gb, (unsigned) ny), dim3(256), 0, ctx->stream, (const float2*) cs->d_xy,
                         (const int32_t*) cs->d_start, (const int32_t*) cs->d_count, (const DistMeta*) d.d_meta, d_cellgoal, c0);
    }
    for (int c0 = 0; c0 < nc; c0 += 32768) {
      const int ny = nc - c0 < 32768 ? nc - c0 : 32768;
      hipLaunchKernelGGL(k_distmap_fill, dim3((unsigned) fb, (unsigned) ny), dim3(256), 0, ctx->stream, (const DistMeta*) d.d_meta,
                         (const int32_t*) d_cellgoal, d.d_parent, mds_px, R, c0);
    }
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, stream_sync(ctx));
  t_dmeta.release(); t_parent.release();          // owned by the cache from here on (t_goal is freed by its guard)
  cs->dists.push_back(d);
  *out = DistDev{d.d_meta, d.d_parent};
  return LSM2D_SUCCESS;
}

static Iso make_iso(const float pose[3]) { Iso T; sincos_fixed(pose[2], T.s, T.c); T.tx = pose[0]; T.ty = pose[1]; return T; }
static float wrap_host(float a) {
  while (a > 3.14159274101257324f) a -= 6.28318548202514648f;
  while (a <= -3.14159274101257324f) a += 6.28318548202514648f;
  return a;
}
static void inverse_host(const float a[3], float out[3]) {   // (R,t)^-1 = (R^T, -R^T t)
  float s, c; sincos_fixed(a[2], s, c);
  out[0] = -(fmaf(c, a[0], s * a[1]));
  out[1] = -(fmaf(-s, a[0], c * a[1]));
  out[2] = wrap_host(-a[2]);
}
static bool valid_cloud_index(const lsm2d_cloudset* cs, int32_t i) { return cs && i >= 0 && i < cs->n_clouds; }
static bool valid_cloud_index_fwd(const lsm2d_cloudset* cs, int32_t i) { return valid_cloud_index(cs, i); }
static void compose_host(const float a[3], const float b[3], float out[3]) {   // v2t(a) * v2t(b)
  float s, c; sincos_fixed(a[2], s, c);
  out[0] = fmaf(c, b[0], fmaf(-s, b[1], a[0]));
  out[1] = fmaf(s, b[0], fmaf(c, b[1], a[1]));
  out[2] = wrap_host(a[2] + b[2]);
}

// z-buffer of one cloud spread over many workgroups into a global canvas (d_canvas: cols u64 cells)
static int project_split(lsm2d_context* ctx, const float2* d_xy, int n, const Iso& T, const ProjK& P, u64* d_canvas) {
  HIPCHK(ctx, hipMemsetAsync(d_canvas, 0xFF, sizeof(u64) * (size_t) P.cols, ctx->stream));
  if (n <= 0) return LSM2D_SUCCESS;
  ProjectSplitArgs A; A.xy = d_xy; A.n = n; A.T = T; A.proj = P; A.gcanvas = d_canvas;
  int blocks = (n + 8191) / 8192; if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(k_project_split, dim3(blocks), dim3(512), sizeof(u64) * (size_t) P.cols, ctx->stream, A);
  HIPCHK(ctx, hipGetLastError());
  return LSM2D_SUCCESS;
}

// a BATCH of scans: the small form of the kernel (512 threads, 37 KB: the footprint of one k_align workgroup) whenever the beams fit it -- it runs beside a launch in
// flight in single slots as they come free, four to a CU when the chip is idle; a handful of scans keep one beam per thread (a scan's latency, the tracker's concern)
static void launch_preprocess_scans(const PrepArgs& A, int n_scans, hipStream_t st) {
  if (n_scans >= 8 && A.n_beams <= kPrepSmallBeams) hipLaunchKernelGGL(k_preprocess_scans_small, dim3((unsigned) n_scans), dim3(kPrepSmallBlock), 0, st, A);
  else hipLaunchKernelGGL(k_preprocess_scans, dim3((unsigned) n_scans), dim3(kPrepBlock), 0, st, A);
}
// ---- RawDataPreprocessorProjective2D, batched -------------------------------------------------------------------
extern "C" int lsm2d_preprocess_scans(lsm2d_context* ctx, const lsm2d_preprocessor* pp, const float* ranges, int32_t n_scans,
                                      lsm2d_cloudset** out) {
  if (!ctx || !pp || !out || n_scans < 1 || !ranges) return fail(ctx, LSM2D_BAD_ARGUMENT, "preprocess_scans: bad argument");
  *out = nullptr;
  const int nb = pp->n_beams;
  if (nb < 1 || nb > kPrepMaxBeams) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "preprocess_scans: n_beams must be in [1, 2048]");
  if (!(pp->angle_max > pp->angle_min) || pp->normal_min_points < 1 || !(pp->normal_point_distance >= 0.0f))
    return fail(ctx, LSM2D_BAD_ARGUMENT, "preprocess_scans: bad parameters");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int stride = nb + (nb & 1);
  if ((int64_t) stride * n_scans > 0x7ffffff0) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "preprocess_scans: too many points");
  lsm2d_cloudset* cs = new (std::nothrow) lsm2d_cloudset;
  if (!cs) return LSM2D_OUT_OF_MEMORY;
  cs->ctx = ctx; ctx->live_sets.push_back(cs); cs->n_clouds = n_scans; cs->padded_total = (int64_t) stride * n_scans + 2;
  cs->h_start.resize(n_scans); cs->h_count.assign(n_scans, 0);
  for (int c = 0; c < n_scans; ++c) cs->h_start[c] = c * stride;
  int rc = cloudset_alloc(ctx, cs);
  if (rc != LSM2D_SUCCESS) { lsm2d_cloudset_destroy(cs); return rc; }
  // where the ranges live: device memory is read in place, pinned (or registered) host memory is copied from directly, pageable
  // host memory goes through the context's pinned staging buffer (one more pass over it on the host)
  int rdev = -1;
  const PtrKind kind = pointer_kind(ranges, &rdev);
  if (kind == PtrKind::device && rdev != ctx->device) { lsm2d_cloudset_destroy(cs); return fail(ctx, LSM2D_BAD_ARGUMENT, "preprocess_scans: device-resident ranges must live on the context's device"); }
  // beam directions with the host libm (the oracle does the same): angle = (c - n/2) * sensor_res
  const size_t rbytes = sizeof(float) * (size_t) nb * (size_t) n_scans, dbytes = sizeof(float2) * (size_t) nb;
  const size_t o_rng = (dbytes + 255) & ~(size_t) 255;
  rc = ensure_scratch(ctx, o_rng + (kind == PtrKind::device ? 0 : rbytes)); if (rc) { lsm2d_cloudset_destroy(cs); return rc; }
  rc = ensure_stage(ctx, o_rng + (kind == PtrKind::pageable ? rbytes : 0)); if (rc) { lsm2d_cloudset_destroy(cs); return rc; }
  if (kind == PtrKind::pageable) memcpy((char*) ctx->h_stage + o_rng, ranges, rbytes);
  float2* hd = (float2*) ctx->h_stage;
  const float sensor_res = (pp->angle_max - pp->angle_min) / (float) nb, k01 = (float) nb * 0.5f;
  for (int c = 0; c < nb; ++c) { const float a = ((float) c - k01) * sensor_res; hd[c] = make_float2(cosf(a), sinf(a)); }
  PrepArgs A;
  A.ranges = kind == PtrKind::device ? ranges : (const float*) ((char*) ctx->d_scratch + o_rng); A.beam_dir = (const float2*) ctx->d_scratch;
  A.n_beams = nb; A.stride = stride; A.rmin = pp->range_min; A.rmax = pp->range_max;
  A.d2max = pp->normal_point_distance * pp->normal_point_distance; A.min_points = pp->normal_min_points;
  A.inv_res = pp->voxelize_resolution > 0.0f ? 1.0f / pp->voxelize_resolution : 0.0f;
  A.out_xy = cs->d_xy; A.out_nrm = cs->d_nrm; A.out_count = cs->d_count;
  hipError_t e = hipMemcpyAsync(ctx->d_scratch, ctx->h_stage, kind == PtrKind::pageable ? o_rng + rbytes : dbytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && kind == PtrKind::pinned) e = hipMemcpyAsync((char*) ctx->d_scratch + o_rng, ranges, rbytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && ctx->kernel_timing) e = hipEventRecord(ctx->ev0, ctx->stream);
  if (e == hipSuccess) { launch_preprocess_scans(A, n_scans, ctx->stream); e = hipGetLastError(); }
  if (e == hipSuccess && ctx->kernel_timing) e = hipEventRecord(ctx->ev1, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(cs->h_count.data(), cs->d_count, sizeof(int32_t) * (size_t) n_scans, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = stream_sync(ctx);
  if (e != hipSuccess) { lsm2d_cloudset_destroy(cs); HIPCHK(ctx, e); }
  note_timed(ctx, ctx->kernel_timing);
  cs->total = 0; for (int c = 0; c < n_scans; ++c) cs->total += cs->h_count[c];
  *out = cs;
  return LSM2D_SUCCESS;
}

// The same operation INTO an existing set of lsm2d_preprocess_scans (same number of scans and beams): no allocation, nothing waits -- the streaming form (a batch of
// fresh scans per step while the previous batch aligns).  The ranges are copied to a device buffer of the set's own (pinned host memory: an asynchronous DMA;
// pageable: staged through the set's pinned buffer; device memory: read in place) and preprocessed by one launch, both on the stream a batch's pre-kernels go to
// (pre_stream: the second stream while a batch is in flight).  The clouds' sizes stay on the device (the host keeps the upper bound n_beams per scan) until
// somebody asks; the set's AoS rows, if it has them, are rewritten by the same launch.
extern "C" int lsm2d_preprocess_scans_refill(lsm2d_context* ctx, const lsm2d_preprocessor* pp, const float* ranges, int32_t n_scans, lsm2d_cloudset* set) {
  if (!ctx || !pp || !ranges || !set || set->ctx != ctx || n_scans < 1) return fail(ctx, LSM2D_BAD_ARGUMENT, "preprocess_scans_refill: bad argument");
  const int nb = pp->n_beams;
  if (nb < 1 || nb > kPrepMaxBeams) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "preprocess_scans_refill: n_beams must be in [1, 2048]");
  if (!(pp->angle_max > pp->angle_min) || pp->normal_min_points < 1 || !(pp->normal_point_distance >= 0.0f))
    return fail(ctx, LSM2D_BAD_ARGUMENT, "preprocess_scans_refill: bad parameters");
  const int stride = nb + (nb & 1);
  if (set->n_clouds != n_scans || set->capacity > 0 || set->padded_total != (int64_t) stride * n_scans + 2 || (n_scans > 1 && set->h_start[1] != stride))
    return fail(ctx, LSM2D_BAD_ARGUMENT, "preprocess_scans_refill: the set must come from lsm2d_preprocess_scans with the same number of scans and beams");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rdev = -1;
  const PtrKind kind = pointer_kind(ranges, &rdev);
  if (kind == PtrKind::device && rdev != ctx->device) return fail(ctx, LSM2D_BAD_ARGUMENT, "preprocess_scans_refill: device-resident ranges must live on the context's device");
  const size_t rbytes = sizeof(float) * (size_t) nb * (size_t) n_scans;
  // everything derived from the old contents goes -- except the AoS rows, which this launch rewrites in place (hipFree is a device-wide wait)
  float4* keep_aos = set->d_aos; set->d_aos = nullptr;
  cloudset_drop_grids(set);
  set->d_aos = keep_aos;
  if (kind != PtrKind::device && !set->d_ranges) HIPCHK(ctx, hipMalloc((void**) &set->d_ranges, rbytes));
  // beam directions with the host libm (the oracle does the same), once per sensor geometry
  const float2* d_dir = nullptr;
  for (const auto& bd : ctx->beam_dirs) if (bd.n_beams == nb && bd.angle_min == pp->angle_min && bd.angle_max == pp->angle_max) d_dir = bd.d_dir;
  if (!d_dir) {
    std::vector<float2> hd((size_t) nb);
    const float sensor_res = (pp->angle_max - pp->angle_min) / (float) nb, k01 = (float) nb * 0.5f;
    for (int c = 0; c < nb; ++c) { const float a = ((float) c - k01) * sensor_res; hd[c] = make_float2(cosf(a), sinf(a)); }
    float2* d = nullptr;
    HIPCHK(ctx, hipMalloc((void**) &d, sizeof(float2) * (size_t) nb));
    hipError_t e = hipMemcpy(d, hd.data(), sizeof(float2) * (size_t) nb, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void) hipFree(d); HIPCHK(ctx, e); }
    ctx->beam_dirs.push_back({nb, pp->angle_min, pp->angle_max, d}); d_dir = d;
  }
  const hipStream_t pre = refill_stream(ctx), cpy = refill_copy_stream(ctx, pre);
  if (set->ev_prep) HIPCHK(ctx, hipStreamWaitEvent(cpy, set->ev_prep, 0));      // (the same set refilled twice in a row: its previous launch, on the refill stream, may still read d_ranges and write the clouds)
  if (kind == PtrKind::pageable) {
    const int rc = acquire_upload_stage(set, rbytes + 16); if (rc) return rc;
    memcpy(set->h_upload, ranges, rbytes);
    HIPCHK(ctx, hipMemcpyAsync(set->d_ranges, set->h_upload, rbytes, hipMemcpyHostToDevice, cpy));
    set->staged_epoch = ctx->sync_epoch;
    if (cpy != ctx->stream) {      // (round-5 advisor: the epoch above speaks for the context's stream only)
      if (!set->ev_stage && hipEventCreateWithFlags(&set->ev_stage, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); set->ev_stage = nullptr; }
      if (set->ev_stage) { HIPCHK(ctx, hipEventRecord(set->ev_stage, cpy)); set->stage_on_side = true; }
      else HIPCHK(ctx, hipStreamSynchronize(cpy));
    }
  }
  else if (kind == PtrKind::pinned) HIPCHK(ctx, hipMemcpyAsync(set->d_ranges, ranges, rbytes, hipMemcpyHostToDevice, cpy));
  if (cpy != pre && kind != PtrKind::device) { HIPCHK(ctx, hipEventRecord(ctx->ev_h, cpy)); HIPCHK(ctx, hipStreamWaitEvent(pre, ctx->ev_h, 0)); }
  ++ctx->uploads; ctx->last_h2d_bytes = kind == PtrKind::device ? 0 : (long long) rbytes;
  PrepArgs A;
  A.ranges = kind == PtrKind::device ? ranges : (const float*) set->d_ranges; A.beam_dir = d_dir;
  A.n_beams = nb; A.stride = stride; A.rmin = pp->range_min; A.rmax = pp->range_max;
  A.d2max = pp->normal_point_distance * pp->normal_point_distance; A.min_points = pp->normal_min_points;
  A.inv_res = pp->voxelize_resolution > 0.0f ? 1.0f / pp->voxelize_resolution : 0.0f;
  A.out_xy = set->d_xy; A.out_nrm = set->d_nrm; A.out_count = set->d_count; A.out_aos = set->d_aos;
  launch_preprocess_scans(A, n_scans, pre);
  HIPCHK(ctx, hipGetLastError());
  if (pre != ctx->stream) {
    HIPCHK(ctx, hipEventRecord(ctx->ev_c, pre)); ctx->c_dirty = true;      // the next aligner call waits for it (join_refill_stream)
    if (!set->ev_prep && hipEventCreateWithFlags(&set->ev_prep, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); set->ev_prep = nullptr; }
    if (set->ev_prep) HIPCHK(ctx, hipEventRecord(set->ev_prep, pre));
    else HIPCHK(ctx, hipStreamSynchronize(pre));      // (no event to be had: the only safe order left)
  }
  for (int c = 0; c < n_scans; ++c) set->h_count[c] = nb;      // upper bounds: the real sizes are on the device
  set->total = (int64_t) nb * n_scans; set->count_pending = true; set->unpack_pending = false; set->prep_pending = false;
  return LSM2D_SUCCESS;
}

// the live tracker's form: ONE scan into an existing reserved set, no allocation, no wait (size pending on the device)
extern "C" int lsm2d_preprocess_scan_into(lsm2d_context* ctx, const lsm2d_preprocessor* pp, const float* ranges, lsm2d_cloudset* out) {
  if (!ctx || !pp || !ranges || !out || out->ctx != ctx || out->n_clouds != 1) return fail(ctx, LSM2D_BAD_ARGUMENT, "preprocess_scan_into: bad argument");
  const int nb = pp->n_beams;
  if (nb < 1 || nb > kPrepMaxBeams) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "preprocess_scan_into: n_beams must be in [1, 2048]");
  if (!(pp->angle_max > pp->angle_min) || pp->normal_min_points < 1 || !(pp->normal_point_distance >= 0.0f))
    return fail(ctx, LSM2D_BAD_ARGUMENT, "preprocess_scan_into: bad parameters");
  const int64_t cap = out->capacity > 0 ? out->capacity : out->padded_total - 2;
  if (cap < nb) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "preprocess_scan_into: the set must have room for n_beams points");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  cloudset_drop_grids(out);
  // beam directions with the host libm (the oracle does the same), once per sensor geometry
  const float2* d_dir = nullptr;
  for (const auto& bd : ctx->beam_dirs) if (bd.n_beams == nb && bd.angle_min == pp->angle_min && bd.angle_max == pp->angle_max) d_dir = bd.d_dir;
  if (!d_dir) {
    std::vector<float2> hd((size_t) nb);
    const float sensor_res = (pp->angle_max - pp->angle_min) / (float) nb, k01 = (float) nb * 0.5f;
    for (int c = 0; c < nb; ++c) { const float a = ((float) c - k01) * sensor_res; hd[c] = make_float2(cosf(a), sinf(a)); }
    float2* d = nullptr;
    HIPCHK(ctx, hipMalloc((void**) &d, sizeof(float2) * (size_t) nb));
    hipError_t e = hipMemcpy(d, hd.data(), sizeof(float2) * (size_t) nb, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void) hipFree(d); HIPCHK(ctx, e); }
    ctx->beam_dirs.push_back({nb, pp->angle_min, pp->angle_max, d}); d_dir = d;
  }
  const size_t rbytes = sizeof(float) * (size_t) nb;
  out->unpack_pending = false;                  // an upload nobody read is simply replaced
  int rc = acquire_upload_stage(out, rbytes + 16); if (rc) return rc;
  memcpy(out->h_upload, ranges, rbytes);
  void* dev_view = out->h_upload_dev;          // the kernel reads the ranges straight from the pinned buffer: no copy, no extra launch
  PrepArgs A;
  A.ranges = (const float*) dev_view; A.beam_dir = d_dir;
  A.n_beams = nb; A.stride = nb + (nb & 1); A.rmin = pp->range_min; A.rmax = pp->range_max;
  A.d2max = pp->normal_point_distance * pp->normal_point_distance; A.min_points = pp->normal_min_points;
  A.inv_res = pp->voxelize_resolution > 0.0f ? 1.0f / pp->voxelize_resolution : 0.0f;
  A.out_xy = out->d_xy; A.out_nrm = out->d_nrm; A.out_count = out->d_count;
  out->h_count[0] = nb; out->total = nb; out->count_pending = true;          // at most one point per beam
  if (!ctx->kernel_timing) {                    // the launch is queued by the set's first reader (flush_pending / flush_preprocessing_together)
    out->prep_pending = true; out->prep_args = A;
    return LSM2D_SUCCESS;
  }
  out->prep_pending = false;
  HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(k_preprocess_scans, dim3(1), dim3(kPrepBlock), 0, ctx->stream, A);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  out->staged_epoch = ctx->sync_epoch;                    // the staging buffer is free again once the kernel has run
  note_timed(ctx, true);
  return LSM2D_SUCCESS;
}

// ---- SceneClipperProjective2D ------------------------------------------------------------------------------
// vox_res > 0: the voxelize_resolution branch (mapping/scene_clipper_projective_2d.cpp:36-48) -- the clip kernels leave the cloud in the
// SENSOR frame and k_voxelize_clipped voxelises it and moves it to the robot frame
static int clip_scene_impl(lsm2d_context* ctx, const lsm2d_projector* pr, const lsm2d_cloudset* scene, int32_t si,
                           const float robot_in_local_map[3], const float sensor_in_robot[3], float vox_res, lsm2d_cloudset* clipped,
                           int32_t* out_n, int32_t* out_src) {
  if (!ctx || !pr || !robot_in_local_map || !sensor_in_robot || !clipped || !valid_cloud_index(scene, si) || clipped->n_clouds != 1 ||
      clipped == scene || (!out_n && out_src))
    return fail(ctx, LSM2D_BAD_ARGUMENT, "clip_scene: bad argument");
  const bool vox = vox_res > 0.0f;
  if (vox && out_src) return fail(ctx, LSM2D_BAD_ARGUMENT, "clip_scene: a voxelised cloud has no source indices");
  if (vox && pr->canvas_cols > kVoxMax) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "clip_scene: voxelisation takes at most 2048 columns");
  // out_n == NULL: asynchronous -- nothing comes back, the clipped set's size stays on the device until somebody asks
  if (scene->count_pending && scene->h_count[si] > 32768) { const int rc0 = resolve_count(scene); if (rc0) return rc0; }
  ProjK P;
  if (!make_projk(*pr, &P)) return fail(ctx, LSM2D_BAD_ARGUMENT, "clip_scene: bad projector");
  const int64_t cap = clipped->capacity > 0 ? clipped->capacity : clipped->padded_total - 2;
  if (cap < P.cols) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "clip_scene: clipped set smaller than canvas_cols");
  if ((int) (sizeof(u64) * (size_t) P.cols) > ctx->max_dyn_lds) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "clip_scene: canvas does not fit LDS");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  cloudset_drop_grids(clipped);
  { const int rc0 = flush_pending(scene); if (rc0) return rc0; }
  clipped->unpack_pending = false; clipped->prep_pending = false;      // whatever was staged for the output set is replaced
  const size_t cols = (size_t) P.cols, o_src = cols * 8, o_cnt = o_src + cols * 4, bytes = o_cnt + 16;
  int rc = ensure_scratch(ctx, bytes); if (rc) return rc;
  rc = ensure_stage(ctx, bytes); if (rc) return rc;
  float cam[3], cam_inv[3];
  compose_host(robot_in_local_map, sensor_in_robot, cam); inverse_host(cam, cam_inv);
  const Iso T = make_iso(cam_inv);
  u64* d_canvas = (u64*) ctx->d_scratch;
  if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  const bool small = scene->h_count[si] <= 32768;            // one workgroup, LDS canvas, one launch
  if (!small) { rc = project_split(ctx, scene->d_xy + scene->h_start[si], scene->h_count[si], T, P, d_canvas); if (rc) return rc; }
  ClipEmitArgs A;
  A.gcanvas = d_canvas; A.cols = P.cols; A.xy = scene->d_xy + scene->h_start[si]; A.nrm = scene->d_nrm + scene->h_start[si];
  A.T = T; A.S = make_iso(sensor_in_robot);
  const bool s_ident = sensor_in_robot[0] == 0.0f && sensor_in_robot[1] == 0.0f && sensor_in_robot[2] == 0.0f;
  A.s_identity = s_ident || vox;                  // voxelisation happens in the sensor frame
  // synchronous form: source indices and the count go straight to the pinned staging buffer (no device-to-host copy)
  char* dvo = (char*) ctx->d_scratch;
  if (out_n) { rc = stage_device_view(ctx, &dvo); if (rc) return rc; *(int32_t*) ((char*) ctx->h_stage + o_cnt) = kStatusNotWritten; }
  A.out_xy = clipped->d_xy; A.out_nrm = clipped->d_nrm; A.out_src = (int32_t*) (dvo + o_src);
  A.out_count = vox ? (int32_t*) ((char*) ctx->d_scratch + o_cnt) : (int32_t*) (dvo + o_cnt);      // with voxelisation the final count is k_voxelize_clipped's
  A.out_count_dev = clipped->d_count; A.host_polls = out_n != nullptr && !vox;
  if (small) {
    ClipSmallArgs CS; CS.xy = A.xy; CS.nrm = A.nrm; CS.n = scene->h_count[si]; CS.n_dev = scene->count_pending ? scene->d_count : nullptr; CS.proj = P; CS.emit = A;
    hipLaunchKernelGGL(k_clip_small, dim3(1), dim3(kFindBlock), sizeof(u64) * (size_t) P.cols, ctx->stream, CS);
  } else {
    hipLaunchKernelGGL(k_clip_emit, dim3(1), dim3(kFindBlock), 0, ctx->stream, A);
  }
  HIPCHK(ctx, hipGetLastError());
  if (vox) {
    VoxArgs V;
    V.xy = clipped->d_xy; V.nrm = clipped->d_nrm; V.count_dev = clipped->d_count;
    V.inv_rx = 1.0f / vox_res; V.inv_rn = 1.0f / 0.1f;      // coefficients (res, res, 0.1, 0.1): scene_clipper_projective_2d.cpp:46
    V.S = make_iso(sensor_in_robot); V.s_identity = s_ident;
    V.out_count = out_n ? (int32_t*) (dvo + o_cnt) : nullptr; V.host_polls = out_n != nullptr;
    hipLaunchKernelGGL(k_voxelize_clipped, dim3(1), dim3(kVoxBlock), 0, ctx->stream, V);
    HIPCHK(ctx, hipGetLastError());
  }
  if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  note_timed(ctx, ctx->kernel_timing);
  if (!out_n) {                               // at most one point per column
    clipped->h_count[0] = P.cols; clipped->total = P.cols; clipped->count_pending = true;
    return LSM2D_SUCCESS;
  }
  HIPCHK(ctx, wait_for_statuses(ctx, (const int32_t*) ((char*) ctx->h_stage + o_cnt), 1));      // the kernel writes the count last
  const int32_t n = *(const int32_t*) ((char*) ctx->h_stage + o_cnt);
  clipped->h_count[0] = n; clipped->total = n; clipped->count_pending = false; *out_n = n;
  if (out_src) memcpy(out_src, (char*) ctx->h_stage + o_src, sizeof(int32_t) * (size_t) n);
  return LSM2D_SUCCESS;
}

extern "C" int lsm2d_clip_scene(lsm2d_context* ctx, const lsm2d_projector* pr, const lsm2d_cloudset* scene, int32_t si,
                                const float robot_in_local_map[3], const float sensor_in_robot[3], lsm2d_cloudset* clipped,
                                int32_t* out_n, int32_t* out_src) {
  return clip_scene_impl(ctx, pr, scene, si, robot_in_local_map, sensor_in_robot, 0.0f, clipped, out_n, out_src);
}
extern "C" int lsm2d_clip_scene_voxelized(lsm2d_context* ctx, const lsm2d_projector* pr, const lsm2d_cloudset* scene, int32_t si,
                                          const float robot_in_local_map[3], const float sensor_in_robot[3], float voxelize_resolution,
                                          lsm2d_cloudset* clipped, int32_t* out_n, int32_t* out_src) {
  return clip_scene_impl(ctx, pr, scene, si, robot_in_local_map, sensor_in_robot, voxelize_resolution, clipped, out_n, out_src);
}

// ---- MergerProjective2D ----------------------------------------------------------------------------------------
extern "C" int lsm2d_merge_scene(lsm2d_context* ctx, const lsm2d_projector* pr, lsm2d_cloudset* scene, const lsm2d_cloudset* meas,
                                 int32_t mi, const float measurement_in_scene[3], float merge_threshold, int32_t* out_size,
                                 int32_t* out_counts) {
  if (!ctx || !pr || !scene || !measurement_in_scene || !valid_cloud_index(meas, mi) || scene->n_clouds != 1 || scene == meas || (!out_size && out_counts))
    return fail(ctx, LSM2D_BAD_ARGUMENT, "merge_scene: bad argument");
  ProjK P;
  if (!make_projk(*pr, &P)) return fail(ctx, LSM2D_BAD_ARGUMENT, "merge_scene: bad projector");
  const int64_t cap = scene->capacity > 0 ? scene->capacity : scene->padded_total - 2;
  // out_size == NULL: asynchronous.  Sizes only the device knows are upper bounds here; when a bound no longer settles a
  // decision (room left, single-workgroup path) it is replaced by the real number (one synchronisation)
  if (scene->count_pending && (int64_t) scene->h_count[0] + P.cols > cap) { const int rc0 = resolve_count(scene); if (rc0) return rc0; }
  // The single-workgroup path reads device-side sizes; the multi-launch path takes BOTH sizes by value (k_transform_cloud,
  // project_split, k_merge_apply), so whenever the bounds do not settle for the small path -- either set above 32 768 points, or
  // canvases beyond LDS -- every pending size is replaced by the real number first: an upper bound there would run stale slots of the
  // reserved set through the merge.
  const bool small_by_bounds = scene->h_count[0] <= 32768 && meas->h_count[mi] <= 32768 && (int) (sizeof(u64) * 2 * (size_t) P.cols) <= ctx->max_dyn_lds;
  if (!small_by_bounds) {
    int rc0 = resolve_count(scene); if (rc0) return rc0;
    rc0 = resolve_count(meas); if (rc0) return rc0;
  }
  const int n_scene = scene->h_count[0], n_meas = meas->h_count[mi];
  if ((int64_t) n_scene + P.cols > cap) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "merge_scene: scene set has no room for canvas_cols more points");
  if ((int) (sizeof(u64) * (size_t) P.cols) > ctx->max_dyn_lds) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "merge_scene: canvas does not fit LDS");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  cloudset_drop_grids(scene);
  { int rc0 = flush_pending(scene); if (rc0) return rc0; rc0 = flush_pending(meas); if (rc0) return rc0; }
  const size_t cols = (size_t) P.cols, nm = (size_t) (n_meas > 0 ? n_meas : 1);
  const size_t o_mcan = cols * 8, o_out = o_mcan + cols * 8, o_txy = o_out + 64, o_tn = o_txy + ((nm * 8 + 15) & ~(size_t) 15) + 16, bytes = o_tn + nm * 8 + 16;
  int rc = ensure_scratch(ctx, bytes); if (rc) return rc;
  rc = ensure_stage(ctx, o_out + 64); if (rc) return rc;
  float cam_inv[3]; inverse_host(measurement_in_scene, cam_inv);
  const Iso Tinv = make_iso(cam_inv), M = make_iso(measurement_in_scene);
  char* ds = (char*) ctx->d_scratch;
  char* dvo = ds;                                  // synchronous form: the four counters go straight to the pinned staging buffer
  if (out_size) { rc = stage_device_view(ctx, &dvo); if (rc) return rc; *(int32_t*) ((char*) ctx->h_stage + o_out) = kStatusNotWritten; }
  u64* d_scan = (u64*) ds; u64* d_mcan = (u64*) (ds + o_mcan);
  float2* d_txy = (float2*) (ds + o_txy); float2* d_tn = (float2*) (ds + o_tn);
  if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  const bool small = n_scene <= 32768 && n_meas <= 32768 && (int) (sizeof(u64) * 2 * (size_t) P.cols) <= ctx->max_dyn_lds;
  if (small) {                                               // one workgroup does the transform, both z-buffers and the column walk
    MergeSmallArgs MS;
    MS.m.scanvas = nullptr; MS.m.mcanvas = nullptr; MS.m.cols = P.cols; MS.m.sxy = scene->d_xy; MS.m.snrm = scene->d_nrm; MS.m.n_scene = n_scene;
    MS.m.mxy = meas->d_xy + meas->h_start[mi]; MS.m.mnrm = meas->d_nrm + meas->h_start[mi];
    MS.m.far_limit = 0.9f * pr->range_max; MS.m.merge_threshold = merge_threshold;
    MS.m.out = (int32_t*) (dvo + o_out); MS.m.count_dev = scene->d_count; MS.m.host_polls = out_size != nullptr;
    MS.proj = P; MS.Tinv = Tinv; MS.M = M; MS.n_meas = n_meas;
    MS.n_scene_dev = scene->count_pending ? scene->d_count : nullptr; MS.n_meas_dev = meas->count_pending ? meas->d_count + mi : nullptr;
    hipLaunchKernelGGL(k_merge_small, dim3(1), dim3(kFindBlock), sizeof(u64) * 2 * (size_t) P.cols, ctx->stream, MS);
    HIPCHK(ctx, hipGetLastError());
  } else {
  if (n_meas > 0) {
    hipLaunchKernelGGL(k_transform_cloud, dim3((unsigned) ((n_meas + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float2*) (meas->d_xy + meas->h_start[mi]), (const float2*) (meas->d_nrm + meas->h_start[mi]), n_meas, M, d_txy, d_tn);
    HIPCHK(ctx, hipGetLastError());
  }
  rc = project_split(ctx, scene->d_xy, n_scene, Tinv, P, d_scan); if (rc) return rc;
  rc = project_split(ctx, d_txy, n_meas, Tinv, P, d_mcan); if (rc) return rc;
  MergeArgs A;
  A.scanvas = d_scan; A.mcanvas = d_mcan; A.cols = P.cols; A.sxy = scene->d_xy; A.snrm = scene->d_nrm; A.n_scene = n_scene;
  A.mxy = d_txy; A.mnrm = d_tn; A.far_limit = 0.9f * pr->range_max; A.merge_threshold = merge_threshold;
  A.out = (int32_t*) (dvo + o_out); A.count_dev = scene->d_count; A.host_polls = out_size != nullptr;
  hipLaunchKernelGGL(k_merge_apply, dim3(1), dim3(kFindBlock), 0, ctx->stream, A);
  HIPCHK(ctx, hipGetLastError());
  }
  if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  note_timed(ctx, ctx->kernel_timing);
  if (!out_size) {                            // a merge appends at most one point per column
    scene->h_count[0] = n_scene + P.cols; scene->total = scene->h_count[0]; scene->count_pending = true;
    return LSM2D_SUCCESS;
  }
  HIPCHK(ctx, wait_for_statuses(ctx, (const int32_t*) ((char*) ctx->h_stage + o_out), 1));      // the kernel writes the new size last
  const int32_t* h = (const int32_t*) ((char*) ctx->h_stage + o_out);
  scene->h_count[0] = h[0]; scene->total = h[0]; scene->count_pending = false; *out_size = h[0];
  if (out_counts) { out_counts[0] = h[1]; out_counts[1] = h[2]; out_counts[2] = h[3]; }
  return LSM2D_SUCCESS;
}


// several measurements into one scene, in order: ONE launch when everything is small (the live tracker's two scans), else the
// calls one by one -- the same result either way
extern "C" int lsm2d_merge_scenes(lsm2d_context* ctx, const lsm2d_projector* pr, lsm2d_cloudset* scene, int32_t n_measurements,
                                  const lsm2d_cloudset* const* meas, const int32_t* meas_index, const float* measurement_in_scene,
                                  float merge_threshold, int32_t* out_size, int32_t* out_counts) {
  if (!ctx || !pr || !scene || !meas || !measurement_in_scene || n_measurements < 1 || scene->n_clouds != 1 || (!out_size && out_counts))
    return fail(ctx, LSM2D_BAD_ARGUMENT, "merge_scenes: bad argument");
  for (int k = 0; k < n_measurements; ++k)
    if (!valid_cloud_index(meas[k], meas_index ? meas_index[k] : 0) || meas[k] == scene) return fail(ctx, LSM2D_BAD_ARGUMENT, "merge_scenes: bad measurement");
  ProjK P;
  if (!make_projk(*pr, &P)) return fail(ctx, LSM2D_BAD_ARGUMENT, "merge_scenes: bad projector");
  const int64_t cap = scene->capacity > 0 ? scene->capacity : scene->padded_total - 2;
  const int n = n_measurements;
  bool together = n >= 2 && n <= kMergeMulti && (int) (sizeof(u64) * 2 * (size_t) P.cols) <= ctx->max_dyn_lds && !ctx->kernel_timing;
  if (together && scene->count_pending && (int64_t) scene->h_count[0] + (int64_t) n * P.cols > cap) { const int rc0 = resolve_count(scene); if (rc0) return rc0; }
  together = together && (int64_t) scene->h_count[0] + (int64_t) n * P.cols <= cap && (int64_t) scene->h_count[0] + (int64_t) (n - 1) * P.cols <= 32768;
  for (int k = 0; k < n && together; ++k) together = meas[k]->h_count[meas_index ? meas_index[k] : 0] <= 32768;
  if (!together) {                            // one by one (large scenes, one measurement, timed launches)
    for (int k = 0; k < n; ++k) {
      int32_t size = 0;
      const int rc = lsm2d_merge_scene(ctx, pr, scene, meas[k], meas_index ? meas_index[k] : 0, measurement_in_scene + 3 * k, merge_threshold,
                                       out_size ? &size : nullptr, out_counts ? out_counts + 3 * k : nullptr);
      if (rc) return rc;
      if (out_size) *out_size = size;
    }
    return LSM2D_SUCCESS;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  cloudset_drop_grids(scene);
  { const int rc0 = flush_pending(scene); if (rc0) return rc0; }
  for (int k = 0; k < n; ++k) { const int rc0 = flush_pending(meas[k]); if (rc0) return rc0; }
  const size_t cols = (size_t) P.cols, o_out = 0, bytes = 16 * (size_t) n + 64;
  int rc = ensure_scratch(ctx, bytes); if (rc) return rc;
  rc = ensure_stage(ctx, bytes); if (rc) return rc;
  char* dvo = (char*) ctx->d_scratch;         // synchronous form: the counters go straight to the pinned staging buffer, the last size last
  if (out_size) { rc = stage_device_view(ctx, &dvo); if (rc) return rc; *(int32_t*) ((char*) ctx->h_stage + o_out + 16 * (size_t) (n - 1)) = kStatusNotWritten; }
  MergeMultiArgs MM; MM.n = n;
  for (int k = 0; k < n; ++k) {
    const lsm2d_cloudset* ms = meas[k]; const int mi = meas_index ? meas_index[k] : 0;
    const float* mis = measurement_in_scene + 3 * k;
    float cam_inv[3]; inverse_host(mis, cam_inv);
    MergeSmallArgs& MS = MM.a[k];
    MS.m.scanvas = nullptr; MS.m.mcanvas = nullptr; MS.m.cols = P.cols; MS.m.sxy = scene->d_xy; MS.m.snrm = scene->d_nrm; MS.m.n_scene = scene->h_count[0];
    MS.m.mxy = ms->d_xy + ms->h_start[mi]; MS.m.mnrm = ms->d_nrm + ms->h_start[mi];
    MS.m.far_limit = 0.9f * pr->range_max; MS.m.merge_threshold = merge_threshold;
    MS.m.out = (int32_t*) (dvo + o_out + 16 * (size_t) k); MS.m.count_dev = scene->d_count; MS.m.host_polls = out_size != nullptr && k == n - 1;
    MS.proj = P; MS.Tinv = make_iso(cam_inv); MS.M = make_iso(mis); MS.n_meas = ms->h_count[mi];
    MS.n_scene_dev = scene->count_pending ? scene->d_count : nullptr; MS.n_meas_dev = ms->count_pending ? ms->d_count + mi : nullptr;
  }
  for (int k = n; k < kMergeMulti; ++k) MM.a[k] = MM.a[0];
  hipLaunchKernelGGL(k_merge_multi, dim3(1), dim3(kFindBlock), sizeof(u64) * 2 * cols, ctx->stream, MM);
  HIPCHK(ctx, hipGetLastError());
  ctx->have_timing = false;
  if (!out_size) {                            // every merge appends at most one point per column
    scene->h_count[0] = scene->h_count[0] + n * P.cols; scene->total = scene->h_count[0]; scene->count_pending = true;
    return LSM2D_SUCCESS;
  }
  HIPCHK(ctx, wait_for_statuses(ctx, (const int32_t*) ((char*) ctx->h_stage + o_out + 16 * (size_t) (n - 1)), 1));
  const int32_t* h = (const int32_t*) ((char*) ctx->h_stage + o_out);
  scene->h_count[0] = h[4 * (n - 1)]; scene->total = scene->h_count[0]; scene->count_pending = false; *out_size = scene->h_count[0];
  if (out_counts) for (int k = 0; k < n; ++k) { out_counts[3 * k] = h[4 * k + 1]; out_counts[3 * k + 1] = h[4 * k + 2]; out_counts[3 * k + 2] = h[4 * k + 3]; }
  return LSM2D_SUCCESS;
}

// ---- a3 ------------------------------------------------------------------------------------------------
extern "C" int lsm2d_project(lsm2d_context* ctx, const lsm2d_projector* pr, const lsm2d_cloudset* cloud, int32_t ci,
                             const float pose[3], int32_t* out_src, float* out_depth, float* out_xynn) {
  if (!ctx || !pr || !pose || !valid_cloud_index(cloud, ci)) return fail(ctx, LSM2D_BAD_ARGUMENT, "project: bad argument");
  { int rc0 = resolve_count(cloud); if (rc0) return rc0; rc0 = flush_pending(cloud); if (rc0) return rc0; }
  ProjectArgs A;
  if (!make_projk(*pr, &A.proj)) return fail(ctx, LSM2D_BAD_ARGUMENT, "project: bad projector");
  const size_t lds = sizeof(u64) * (size_t) A.proj.cols;
  if ((int) lds > ctx->max_dyn_lds) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "project: canvas does not fit LDS");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t cols = (size_t) A.proj.cols, bytes = cols * (4 + 4 + 16);
  int rc = ensure_scratch(ctx, bytes); if (rc) return rc;
  rc = ensure_stage(ctx, bytes); if (rc) return rc;
  A.cloud = cloud_dev(cloud, nullptr); A.ci = ci; A.T = make_iso(pose);
  char* dv = nullptr; rc = stage_device_view(ctx, &dv); if (rc) return rc;       // the canvas rows go straight to pinned host memory
  A.out_xynn = (float4*) dv;
  A.out_src = (int32_t*) (dv + cols * 16);
  A.out_depth = (float*) (dv + cols * 20);
  hipLaunchKernelGGL(k_project_canvas, dim3(1), dim3(kFindBlock), lds, ctx->stream, A);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, stream_sync(ctx));
  if (out_xynn) memcpy(out_xynn, ctx->h_stage, cols * 16);
  if (out_src) memcpy(out_src, (char*) ctx->h_stage + cols * 16, cols * 4);
  if (out_depth) memcpy(out_depth, (char*) ctx->h_stage + cols * 20, cols * 4);
  return LSM2D_SUCCESS;
}

// ---- plugin interface #1 ---------------------------------------------------------------------------------
// inl_tau > 0: only the pairs whose factor is an inlier under a Cauchy robustifier of that threshold (FindArgs::inl_tau)
static int find_correspondences_impl(lsm2d_context* ctx, const lsm2d_slice_params* sp, const lsm2d_cloudset* fixed,
                                     int32_t fi, const lsm2d_cloudset* moving, int32_t mi, const float pose[3],
                                     lsm2d_correspondence* out_pairs, int32_t capacity, int32_t* out_n, float inl_tau) {
  if (!ctx || !sp || !pose || !out_n || !valid_cloud_index(fixed, fi) || !valid_cloud_index(moving, mi) || capacity < 0 ||
      (capacity > 0 && !out_pairs))
    return fail(ctx, LSM2D_BAD_ARGUMENT, "find_correspondences: bad argument");
  { int rc0 = resolve_count(fixed); if (rc0) return rc0; rc0 = resolve_count(moving); if (rc0) return rc0; }
  { int rc0 = flush_pending(fixed); if (rc0) return rc0; rc0 = flush_pending(moving); if (rc0) return rc0; }
  *out_n = 0;
  if (sp->finder == LSM2D_FINDER_NN || sp->finder == LSM2D_FINDER_DISTMAP || sp->finder == LSM2D_FINDER_KDTREE) {
    if (sp->finder != LSM2D_FINDER_DISTMAP && !(sp->max_distance > 0.0f)) return fail(ctx, LSM2D_BAD_ARGUMENT, "find_correspondences: max_distance must be > 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    FindNNArgs N;
    N.fixed = cloud_dev(fixed, nullptr); N.moving = cloud_dev(moving, nullptr); N.fc = fi; N.mc = mi;
    N.use_distmap = sp->finder == LSM2D_FINDER_DISTMAP; N.use_kd = sp->finder == LSM2D_FINDER_KDTREE;
    int rc = N.use_distmap ? ensure_distmap(ctx, fixed, sp->max_distance, sp->resolution, &N.fixed.dist)
           : N.use_kd      ? ensure_kdtree(ctx, fixed, sp->kd_max_leaf_range, sp->kd_min_leaf_points, &N.fixed.kd)
                           : ensure_grid(ctx, fixed, sp->max_distance, &N.fixed.grid);
    if (rc) return rc;
    const size_t nm = (size_t) moving->h_count[mi], bytes = nm * 8 + 16;
    N.max_distance = sp->max_distance; N.normal_cos = sp->normal_cos; N.T = make_iso(pose); N.inl_tau = inl_tau;
    N.nn_group = fixed->h_count[fi] >= 4 * (int64_t) moving->h_count[mi] ? kNNGroup : 1;     // dense fixed cloud: cooperative search
    // more queries than one workgroup takes in a trip: one workgroup per trip's worth, two launches (search, then ordered compaction)
    const int per_step = kFindBlock / ((N.use_distmap || N.use_kd) ? 1 : N.nn_group);
    const int n_blocks = (int) ((nm + (size_t) per_step - 1) / (size_t) per_step);
    const bool multi = n_blocks > 2 && ctx->find_path != 1;      // (two trips of one workgroup beat two launches: 23 vs 29 us for 1081 distance-map queries)
    const size_t o_match = (bytes + 255) & ~(size_t) 255, o_cnt = o_match + ((nm * 4 + 255) & ~(size_t) 255);
    rc = ensure_scratch(ctx, multi ? o_cnt + 4 * (size_t) n_blocks : bytes); if (rc) return rc;
    rc = ensure_stage(ctx, bytes); if (rc) return rc;
    const bool direct = bytes <= (1u << 16);             // up to 8k pairs: written straight to pinned host memory
    char* dv = (char*) ctx->d_scratch;
    if (direct) { rc = stage_device_view(ctx, &dv); if (rc) return rc; }
    N.out_count = (int32_t*) dv; N.out_pairs = (int32_t*) (dv + 16);
    N.match = (int32_t*) ((char*) ctx->d_scratch + o_match); N.block_count = (int32_t*) ((char*) ctx->d_scratch + o_cnt);
    if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (multi) {
      hipLaunchKernelGGL(k_find_nn_multi<0>, dim3((unsigned) n_blocks), dim3(kFindBlock), 0, ctx->stream, N);
      hipLaunchKernelGGL(k_find_nn_multi<1>, dim3((unsigned) n_blocks), dim3(kFindBlock), 0, ctx->stream, N);
    } else {
      hipLaunchKernelGGL(k_find_nn, dim3(1), dim3(kFindBlock), 0, ctx->stream, N);
    }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    note_timed(ctx, ctx->kernel_timing);
    if (!direct) HIPCHK(ctx, hipMemcpyAsync(ctx->h_stage, ctx->d_scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, stream_sync(ctx));
    const int32_t n = *(const int32_t*) ctx->h_stage;
    *out_n = n;
    if (n > capacity) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "find_correspondences: out_pairs too small");
    memcpy(out_pairs, (char*) ctx->h_stage + 16, sizeof(lsm2d_correspondence) * (size_t) n);
    return LSM2D_SUCCESS;
  }
  if (sp->finder != LSM2D_FINDER_PROJECTIVE) return fail(ctx, LSM2D_BAD_ARGUMENT, "find_correspondences: finder not supported yet");
  FindArgs A;
  if (!make_projk(sp->projector, &A.proj)) return fail(ctx, LSM2D_BAD_ARGUMENT, "find_correspondences: bad projector");
  const size_t cols = (size_t) A.proj.cols, lds = sizeof(u64) * 2 * cols;
  if ((int) lds > ctx->max_dyn_lds) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "find_correspondences: canvases do not fit LDS");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = cols * 8 + 16;
  // a map-sized cloud is z-buffered over many workgroups first (the clipper's large-scene kernel; u64 minima do not depend on the order)
  const bool big_f = fixed->h_count[fi] > 32768 && ctx->find_path != 1, big_m = moving->h_count[mi] > 32768 && ctx->find_path != 1;
  const size_t o_can = (bytes + 255) & ~(size_t) 255;
  int rc = ensure_scratch(ctx, o_can + 2 * cols * sizeof(u64)); if (rc) return rc;
  rc = ensure_stage(ctx, bytes); if (rc) return rc;
  A.fixed = cloud_dev(fixed, nullptr); A.moving = cloud_dev(moving, nullptr); A.fc = fi; A.mc = mi;
  A.point_distance = sp->point_distance; A.normal_cos = sp->normal_cos; A.T = make_iso(pose); A.inl_tau = inl_tau;
  char* dv = nullptr; rc = stage_device_view(ctx, &dv); if (rc) return rc;       // <= one pair per column: written straight to pinned host memory
  A.out_count = (int32_t*) dv; A.out_pairs = (int32_t*) (dv + 16);
  A.fcan_global = nullptr; A.mcan_global = nullptr;
  if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  if (big_f) {
    u64* g = (u64*) ((char*) ctx->d_scratch + o_can); const Iso ident = {1.0f, 0.0f, 0.0f, 0.0f};
    rc = project_split(ctx, fixed->d_xy + fixed->h_start[fi], fixed->h_count[fi], ident, A.proj, g); if (rc) return rc;
    A.fcan_global = g;
  }
  if (big_m) {
    u64* g = (u64*) ((char*) ctx->d_scratch + o_can) + cols;
    rc = project_split(ctx, moving->d_xy + moving->h_start[mi], moving->h_count[mi], A.T, A.proj, g); if (rc) return rc;
    A.mcan_global = g;
  }
  hipLaunchKernelGGL(k_find_projective, dim3(1), dim3(kFindBlock), lds, ctx->stream, A);
  HIPCHK(ctx, hipGetLastError());
  if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  note_timed(ctx, ctx->kernel_timing);
  HIPCHK(ctx, stream_sync(ctx));
  const int32_t n = *(const int32_t*) ctx->h_stage;
  *out_n = n;
  if (n > capacity) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "find_correspondences: out_pairs too small");
  memcpy(out_pairs, (char*) ctx->h_stage + 16, sizeof(lsm2d_correspondence) * (size_t) n);
  return LSM2D_SUCCESS;
}

extern "C" int lsm2d_find_correspondences(lsm2d_context* ctx, const lsm2d_slice_params* sp, const lsm2d_cloudset* fixed,
                                          int32_t fi, const lsm2d_cloudset* moving, int32_t mi, const float pose[3],
                                          lsm2d_correspondence* out_pairs, int32_t capacity, int32_t* out_n) {
  return find_correspondences_impl(ctx, sp, fixed, fi, moving, mi, pose, out_pairs, capacity, out_n, 0.0f);
}

// ---- factor ---------------------------------------------------------------------------------------------------
extern "C" int lsm2d_linearize(lsm2d_context* ctx, const lsm2d_slice_params* sp, const lsm2d_cloudset* fixed, int32_t fi,
                               const lsm2d_cloudset* moving, int32_t mi, const lsm2d_correspondence* pairs, int32_t n_pairs,
                               const float pose[3], float out_H[9], float out_b[3], lsm2d_iteration_stats* st) {
  if (!ctx || !sp || !pose || !out_H || !out_b || !valid_cloud_index(fixed, fi) || !valid_cloud_index(moving, mi) || n_pairs < 0 ||
      (n_pairs > 0 && !pairs))
    return fail(ctx, LSM2D_BAD_ARGUMENT, "linearize: bad argument");
  { int rc0 = resolve_count(fixed); if (rc0) return rc0; rc0 = resolve_count(moving); if (rc0) return rc0; }
  { int rc0 = flush_pending(fixed); if (rc0) return rc0; rc0 = flush_pending(moving); if (rc0) return rc0; }
  for (int32_t k = 0; k < n_pairs; ++k)
    if (pairs[k].fixed_idx < 0 || pairs[k].fixed_idx >= fixed->h_count[fi] || pairs[k].moving_idx < 0 || pairs[k].moving_idx >= moving->h_count[mi])
      return fail(ctx, LSM2D_BAD_ARGUMENT, "linearize: correspondence index out of range");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int blocks = (n_pairs + 255) / 256; if (blocks < 1) blocks = 1; if (blocks > 1024) blocks = 1024;
  const size_t pair_bytes = sizeof(lsm2d_correspondence) * (size_t) n_pairs;
  const size_t part_off = (pair_bytes + 255) & ~(size_t) 255, out_off = part_off + sizeof(float) * kAccumWords * (size_t) blocks;
  const size_t dig_off = out_off + sizeof(float) * kAccumWords;      // 8-byte aligned: out_off is a multiple of 256, kAccumWords is even
  static_assert(kAccumWords % 2 == 0, "the digest behind the sums must be 8-byte aligned");
  const size_t bytes = dig_off + sizeof(unsigned long long);
  int rc = ensure_scratch(ctx, bytes); if (rc) return rc;
  rc = ensure_stage(ctx, bytes); if (rc) return rc;
  if (n_pairs) memcpy(ctx->h_stage, pairs, pair_bytes);
  // up to 8k pairs (a canvas worth): the kernels read the pairs from, and write the sums to, the pinned staging buffer directly
  const bool direct = n_pairs <= 8192;
  char* dv = (char*) ctx->d_scratch;
  if (direct) { rc = stage_device_view(ctx, &dv); if (rc) return rc; }
  else if (n_pairs) HIPCHK(ctx, hipMemcpyAsync(ctx->d_scratch, ctx->h_stage, pair_bytes, hipMemcpyHostToDevice, ctx->stream));
  LinArgs A;
  A.fixed = cloud_dev(fixed, nullptr); A.moving = cloud_dev(moving, nullptr); A.fc = fi; A.mc = mi;
  A.pairs = (const int32_t*) dv; A.n_pairs = n_pairs; A.T = make_iso(pose);
  A.cauchy = sp->robustifier == LSM2D_ROBUST_CAUCHY; A.tau = sp->chi_threshold;
  A.partial = (float*) ((char*) ctx->d_scratch + part_off); A.out = (float*) (dv + out_off);
  A.dig = (unsigned long long*) (dv + dig_off);
  if (direct) *(unsigned long long*) ((char*) ctx->h_stage + dig_off) = 0ull;
  else HIPCHK(ctx, hipMemsetAsync(A.dig, 0, sizeof(unsigned long long), ctx->stream));
  if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  if (ctx->sum_order) hipLaunchKernelGGL(k_linearize_seq, dim3(1), dim3(kAlignBlock), 0, ctx->stream, A);      // pair after pair, the order of the vector
  else {
    hipLaunchKernelGGL(k_linearize_partial, dim3(blocks), dim3(256), 0, ctx->stream, A);
    hipLaunchKernelGGL(k_linearize_final, dim3(1), dim3(64), 0, ctx->stream, (const float*) A.partial, blocks, A.out);
  }
  HIPCHK(ctx, hipGetLastError());
  if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  note_timed(ctx, ctx->kernel_timing);
  float* h = (float*) ((char*) ctx->h_stage + out_off);
  if (!direct) HIPCHK(ctx, hipMemcpyAsync(h, A.out, sizeof(float) * kAccumWords + sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, stream_sync(ctx));
  out_H[0] = h[0]; out_H[1] = h[1]; out_H[2] = h[2]; out_H[3] = h[1]; out_H[4] = h[3]; out_H[5] = h[4]; out_H[6] = h[2]; out_H[7] = h[4]; out_H[8] = h[5];
  out_b[0] = h[6]; out_b[1] = h[7]; out_b[2] = h[8];
  if (st) {
    int32_t iv[3]; memcpy(iv, h + 11, sizeof iv);
    st->n_inliers = iv[0]; st->n_outliers = iv[1]; st->n_correspondences = iv[2]; st->chi_inliers = h[9]; st->chi_outliers = h[10];
    unsigned long long dg; memcpy(&dg, (char*) ctx->h_stage + dig_off, sizeof dg);
    st->pair_digest_lo = (uint32_t) dg; st->pair_digest_hi = (uint32_t) (dg >> 32);
  }
  return LSM2D_SUCCESS;
}

// ---- plugin interface #2 ----------------------------------------------------------------------------------------
extern "C" uint64_t lsm2d_pair_hash(uint32_t slice, uint32_t fixed_idx, uint32_t moving_idx) {      // the kernels' pair_hash_dev, on the host
  const uint32_t a = fixed_idx * 0x9E3779B1u, b = (moving_idx ^ (slice * 0x632BE5ABu)) * 0x85EBCA77u;
  uint32_t lo = a ^ ((b << 13) | (b >> 19)), hi = b ^ ((a << 19) | (a >> 13));
  lo += ((lo << 17) | (lo >> 15)) ^ b;
  hi += ((hi << 11) | (hi >> 21)) ^ a;
  return ((uint64_t) hi << 32) | (uint64_t) lo;
}

extern "C" int32_t lsm2d_stats_capacity(const lsm2d_aligner_params* ap) {
  if (!ap) return 1;
  const long long c = (long long) (ap->max_iterations > 0 ? ap->max_iterations : 0) * (ap->enable_inlier_only_runs ? 2 : 1);
  return (int32_t) (c < 1 ? 1 : (c > 0x7fffffff ? 0x7fffffff : c));
}

// What a batch that has been LAUNCHED keeps until its results are asked for (lsm2d_align_batch_begin / _wait; the synchronous calls go through the same two halves)
struct lsm2d_pending {
  lsm2d_context* ctx = nullptr;
  int lane_id = 0; hipEvent_t ev_done = nullptr, ev0 = nullptr, ev1 = nullptr;
  char* hs = nullptr;                       // the lane's pinned staging buffer: where the results are (or are copied to)
  size_t o_pose = 0, o_H = 0, o_status = 0, o_its = 0, o_stats = 0, o_last_pose = 0, o_clock = 0;
  int n = 0, stats_stride = 0, n_clock = 0, clock_stride = 0;
  bool zero_copy = false, want_stats = false, want_last_pose = false, stamps = false, timed = false, async = false;
  uint32_t* xcd_sync = nullptr; int xcd_stride = 0, xcd_window = 0, xcd_positions = 0;
};
static int align_batch_finish(lsm2d_pending& P, float* out_pose, float* out_H, int32_t* out_status, int32_t* out_its, lsm2d_iteration_stats* out_stats, float* out_last_pose);

// out_last_pose [n][3] (may be NULL): the pose the last iteration every alignment started began at (what lsm2d_align_batch_pairs re-derives
// that iteration's correspondences from)
// out_work [n] (may be NULL): ONLY the work estimate of lsm2d_estimate_work is produced -- no alignment runs, the other outputs are not touched
// pend != NULL: lsm2d_align_batch_begin -- the call returns once everything is queued; the output pointers only say WHICH outputs are wanted (non-null), nothing is
// written through them; lsm2d_align_batch_wait -> align_batch_finish hands the results over.
static int align_batch_impl(lsm2d_context* ctx, const lsm2d_aligner_params* ap, const lsm2d_batch* b, float* out_pose,
                            float* out_H, int32_t* out_status, int32_t* out_its, lsm2d_iteration_stats* out_stats, float* out_last_pose,
                            int32_t* out_work = nullptr, lsm2d_pending* pend = nullptr) {
  if (!ctx || !ap || !b || ((!out_pose || !out_status) && !out_work)) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: null argument");
  if (ctx->lane_busy) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: two batches are in flight on this context: wait for the older one first (lsm2d_align_batch_wait)");
  const bool async = pend != nullptr;
  // what goes AHEAD of k_align (start poses, the placement's estimate): on the second stream while another batch is in flight
  const hipStream_t pre = (async && !out_work) ? pre_stream(ctx) : ctx->stream;
  HIPCHK(ctx, join_refill_stream(ctx, pre));      // scans refilled while a batch was in flight: their preprocessing comes before anything this call queues
  const int n = b->n_alignments, ns = b->n_slices;
  if (n < 0 || ns < 1 || ns > kMaxSlices || ap->max_iterations < 0 || !b->slices || !b->fixed || !b->moving || (n > 0 && !b->init_pose))
    return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: bad batch descriptor");
  if (n == 0) return LSM2D_SUCCESS;
  static_assert(sizeof(StatsDev) == sizeof(lsm2d_iteration_stats), "stats layout");
  HIPCHK(ctx, hipSetDevice(ctx->device));

  AlignArgs A; memset(&A, 0, sizeof A);
  A.n_align = n; A.n_slices = ns; A.max_it = ap->max_iterations; A.min_inliers = ap->min_num_inliers; A.damping = ap->damping;
  if (!(ap->termination_chi_epsilon >= 0.0f)) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: termination_chi_epsilon must be >= 0");
  A.term_eps = ap->termination_chi_epsilon;
  if (ap->max_iterations > 0x3fffffff) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: max_iterations out of range");
  A.inlier_runs = ap->enable_inlier_only_runs != 0;
  const int stats_stride = lsm2d_stats_capacity(ap), it_cap = A.inlier_runs ? 2 * ap->max_iterations : ap->max_iterations;
  A.stats_stride = stats_stride;
  // ---- device scratch layout: [init_pose | prior | indices | out_pose | out_H | status | its | stats]
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t) 255; return o; };
  const size_t o_pose_in = take(sizeof(float) * 3 * (size_t) n);
  const size_t o_prior = b->prior ? take(sizeof(PriorDev) * (size_t) n) : 0;
  size_t o_fidx[kMaxSlices] = {0}, o_midx[kMaxSlices] = {0};
  for (int s = 0; s < ns; ++s) {
    if (b->fixed_index) o_fidx[s] = take(sizeof(int32_t) * (size_t) n);
    if (b->moving_index) o_midx[s] = take(sizeof(int32_t) * (size_t) n);
  }
  const size_t in_bytes = off;
  const size_t o_pose = take(sizeof(float) * 3 * (size_t) n), o_H = take(sizeof(float) * 9 * (size_t) n);
  const size_t o_status = take(sizeof(int32_t) * (size_t) n), o_its = take(sizeof(int32_t) * (size_t) n);
  const size_t o_stats = out_stats ? take(sizeof(StatsDev) * (size_t) n * (size_t) stats_stride) : 0;
  const size_t o_last_pose = out_last_pose ? take(sizeof(float) * 3 * (size_t) n) : 0;
  // in-kernel clock stamps of ~32 workgroups spread over the grid (timed k_align launches only)
  const int clock_stride = ctx->clock_stride > 0 ? ctx->clock_stride : (n / 32 > 1 ? n / 32 : 1), n_clock = (n + clock_stride - 1) / clock_stride;
  const size_t o_clock = ctx->kernel_timing ? take(sizeof(unsigned long long) * 4 * (size_t) n_clock) : 0;
  const size_t out_bytes = off - o_pose;      // what travels back to the host
  const size_t o_work = take(sizeof(int32_t) * (size_t) n);                                                     // balanced placement: the estimate's counts (device only)
#ifdef LSM2D_EXPERIMENTS      // the two-launch form of a batch ("two_stage"): its order and the state between its launches (device only)
  const size_t o_order = take(sizeof(int32_t) * (size_t) n), o_resume = take(sizeof(ResumeDev) * (size_t) n);
#endif
  const size_t total_bytes = off;
  const bool had_inputs = ctx->inputs_valid;      // (the lane's scratch still holds the previous batch's input block: ensure_scratch below clears the flag)
  int rc = ensure_scratch(ctx, total_bytes); if (rc) return rc;
  rc = ensure_stage(ctx, total_bytes); if (rc) return rc;
  char* hs = (char*) ctx->h_stage; char* ds = (char*) ctx->d_scratch;

  // ---- which path: few alignments against a big cloud are spread over many workgroups each (projective slices only)
  bool has_proj = false, has_nn = false, has_dist = false, has_kd = false;
  int max_moving = 0;
  for (int s = 0; s < ns; ++s) {
    const int fd = b->slices[s].finder;
    if (fd == LSM2D_FINDER_PROJECTIVE) has_proj = true; else if (fd == LSM2D_FINDER_NN) has_nn = true; else if (fd == LSM2D_FINDER_KDTREE) has_kd = true; else has_dist = true;
    const lsm2d_cloudset* m = b->moving[s];
    if (m) for (int c = 0; c < m->n_clouds; ++c) if (m->h_count[c] > max_moving) max_moving = m->h_count[c];
  }
  const bool split_ok = has_proj && !has_nn && !has_dist && !has_kd && ap->max_iterations > 0 && n <= 32768;
  // measured (tools/small_batch_bench.py, profiles/r01/small_batch*.jsonl): the split path costs two launches per iteration per
  // alignment call and wins whenever one workgroup per alignment would leave most of the chip idle for long enough
  const bool use_split = split_ok && (ctx->align_path == 2 ||
                                      (ctx->align_path == 0 && n <= 192 && (long long) max_moving * ap->max_iterations >= 400000));
  // A handful of alignments in one launch (the live tracker: one): the kernel reads its few hundred bytes of arguments from, and
  // writes its results to, the PINNED staging buffer directly -- no host-to-device copy, no memset, no device-to-host copy, i.e.
  // three small transfers and their launch latencies off the critical path of the call.
  // (Not for big batches: with 1000 alignments reading their start poses and writing their results over the host link the step takes 1.485 ms
  // against 1.467 with the three small transfers; tools/zero_copy_ab.py.)
  // ("zero_copy_max" bounds every batch, so the A/B knob works below 256 too; batches that carry index arrays stay on the transfers above 256)
  const bool zero_copy = !out_work && !use_split && n <= ctx->zero_copy_max && (n <= 256 || (!b->fixed_index && !b->moving_index));
  if (zero_copy) ds = (char*) ctx->h_stage_dev;

  // ---- slices
  for (int s = 0; s < ns; ++s)        // ownership first: nothing is launched for a set that is not this context's (or is orphaned)
    if (!b->fixed[s] || !b->moving[s] || b->fixed[s]->ctx != ctx || b->moving[s]->ctx != ctx)
      return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: cloud set missing or from another context");
  {   // scans whose preprocessing is still pending (lsm2d_preprocess_scan_into): one launch for all of them
    const lsm2d_cloudset* rd[2 * kMaxSlices]; int nr = 0;
    for (int s = 0; s < ns; ++s) { rd[nr++] = b->fixed[s]; rd[nr++] = b->moving[s]; }
    const int prc = flush_preprocessing_together(ctx, rd, nr); if (prc) return prc;
  }
  {   // KD-tree slices over scan-sized fixed sets whose trees are not there yet (the live tracker: a new scan per laser and step -- the reference's reset(),
      // correspondence_finder_kd_tree_2d.cpp:6-8,31-38): ALL of them in one launch, side by side, with one wait (kd_scan_launch)
    KdScanPrep preps[kMaxSlices]; int np_ = 0;
    for (int s = 0; s < ns; ++s) {
      const lsm2d_slice_params& sp = b->slices[s];
      if (sp.finder != LSM2D_FINDER_KDTREE) continue;
      const lsm2d_cloudset* f = b->fixed[s];
      float mlr = sp.kd_max_leaf_range; int mlp = sp.kd_min_leaf_points; kd_defaults(mlr, mlp);
      int rc0 = resolve_count(f); if (rc0) return rc0;
      if (kd_cached(f, mlr, mlp) || !kd_scan_eligible(ctx, f)) continue;
      bool dup = false;
      for (int i = 0; i < np_; ++i) dup = dup || preps[i].cs == f;
      if (dup) continue;      // (one set in two slices: built once here; with other parameters the second goes through ensure_kdtree)
      rc0 = flush_pending(f); if (rc0) return rc0;
      rc0 = kd_scan_prepare(ctx, f, mlr, mlp, preps[np_]); if (rc0) return rc0;
      ++np_;
    }
    if (np_ >= 2) { const int rc0 = kd_scan_launch(ctx, preps, np_); if (rc0) return rc0; }
    else if (np_ == 1) { const int rc0 = kd_scan_launch(ctx, preps, 1); if (rc0) return rc0; }
  }
  int cols_max = 0, fcan_total = 0;
  const KdCache* kd_cache0 = nullptr;      // the KD-tree set of the (last) KD-tree slice
  for (int s = 0; s < ns; ++s) {
    const lsm2d_slice_params& sp = b->slices[s];
    SliceDev& S = A.s[s];
    const lsm2d_cloudset* f = b->fixed[s]; const lsm2d_cloudset* m = b->moving[s];
    if (!f || !m || f->ctx != ctx || m->ctx != ctx) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: cloud set missing or from another context");
    // sizes that only the device knows yet (asynchronous clip / merge) are fine for the projective finder -- the kernels read the
    // device-side counts and the host needs upper bounds only; the search structures of the other finders need the numbers
    if (sp.finder != LSM2D_FINDER_PROJECTIVE) { int rc0 = resolve_count(f); if (rc0) return rc0; rc0 = resolve_count(m); if (rc0) return rc0; }
    if (sp.finder != LSM2D_FINDER_PROJECTIVE && sp.finder != LSM2D_FINDER_NN && sp.finder != LSM2D_FINDER_DISTMAP && sp.finder != LSM2D_FINDER_KDTREE)
      return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: unknown finder");
    if (sp.finder == LSM2D_FINDER_PROJECTIVE) {
      if (!make_projk(sp.projector, &S.proj)) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: bad projector");
    } else {
      if ((sp.finder == LSM2D_FINDER_NN || sp.finder == LSM2D_FINDER_KDTREE) && !(sp.max_distance > 0.0f)) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: max_distance must be > 0");
      memset(&S.proj, 0, sizeof S.proj);
    }
    if (!b->fixed_index && f->n_clouds != 1 && f->n_clouds != n) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: fixed set must hold 1 or n_alignments clouds");
    if (!b->moving_index && m->n_clouds != 1 && m->n_clouds != n) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: moving set must hold 1 or n_alignments clouds");
    const int32_t* d_fi = nullptr; const int32_t* d_mi = nullptr;
    if (b->fixed_index) {
      const int32_t* src = b->fixed_index + (size_t) s * n;
      for (int i = 0; i < n; ++i) if (src[i] < 0 || src[i] >= f->n_clouds) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: fixed_index out of range");
      memcpy(hs + o_fidx[s], src, sizeof(int32_t) * (size_t) n); d_fi = (const int32_t*) (ds + o_fidx[s]);
    }
    if (b->moving_index) {
      const int32_t* src = b->moving_index + (size_t) s * n;
      for (int i = 0; i < n; ++i) if (src[i] < 0 || src[i] >= m->n_clouds) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: moving_index out of range");
      memcpy(hs + o_midx[s], src, sizeof(int32_t) * (size_t) n); d_mi = (const int32_t*) (ds + o_midx[s]);
    }
    // a fixed set whose upload still sits in its pinned buffer: single-alignment projective calls unpack it in the kernel's prologue
    // (decided once the kernel is known, below); every other reader gets it unpacked by a launch of its own, here
    const bool defer_unpack = f->unpack_pending && n == 1 && !use_split && has_proj && !has_nn && !has_dist && !has_kd && f != m;
    if (!defer_unpack) { const int urc = flush_pending(f); if (urc) return urc; }
    { const int urc = flush_pending(m); if (urc) return urc; }
    if (sp.finder == LSM2D_FINDER_PROJECTIVE) { const int lrc = ensure_lane_layout(ctx, m); if (lrc) return lrc; }
    // k_align's bin walk gathers both z-buffer winners as 16-byte rows of the sets' AoS copies (not for the calls the latency kernel or the split
    // path will take: the live tracker's sets change every step)
    const bool pair_candidate = !ctx->sum_order && ctx->align_path != 1 && (ns == 1 || ns == 2) && has_proj && !has_nn && !has_dist && !has_kd && (n <= 256 || ctx->align_path == 3) && ap->max_iterations > 0;
    if (sp.finder == LSM2D_FINDER_PROJECTIVE && !use_split && !pair_candidate && !defer_unpack) {
      int arc = ensure_aos(ctx, f); if (arc) return arc;
      arc = ensure_aos(ctx, m); if (arc) return arc;
    }
    S.fixed = cloud_dev(f, d_fi); S.moving = cloud_dev(m, d_mi);
    S.unpack_src = defer_unpack ? (const float4*) f->h_upload_dev : nullptr; S.unpack_n = defer_unpack ? f->h_count[0] : 0;
    if (sp.finder == LSM2D_FINDER_NN) { const int grc = ensure_grid(ctx, f, sp.max_distance, &S.fixed.grid); if (grc) return grc; }
    if (sp.finder == LSM2D_FINDER_DISTMAP) { const int grc = ensure_distmap(ctx, f, sp.max_distance, sp.resolution, &S.fixed.dist); if (grc) return grc; }
    if (sp.finder == LSM2D_FINDER_KDTREE) { const int grc = ensure_kdtree(ctx, f, sp.kd_max_leaf_range, sp.kd_min_leaf_points, &S.fixed.kd, &kd_cache0); if (grc) return grc; }
    {   // cooperative NN search pays when the fixed cloud is much denser than the queries (map as fixed, scans as queries)
      int64_t mf = 0, mm = 1;
      for (int c = 0; c < f->n_clouds; ++c) if (f->h_count[c] > mf) mf = f->h_count[c];
      for (int c = 0; c < m->n_clouds; ++c) if (m->h_count[c] > mm) mm = m->h_count[c];
      S.nn_group = mf >= 4 * mm ? kNNGroup : 1;
    }
    S.finder = sp.finder; S.point_distance = sp.point_distance; S.normal_cos = sp.normal_cos; S.max_distance = sp.max_distance;
    S.cauchy = sp.robustifier == LSM2D_ROBUST_CAUCHY; S.tau = sp.chi_threshold; S.min_corr = sp.min_num_correspondences;
    if (S.cauchy && !(S.tau > 0.0f)) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch: chi_threshold must be > 0");
    S.has_sensor = !(sp.sensor_in_robot[0] == 0.0f && sp.sensor_in_robot[1] == 0.0f && sp.sensor_in_robot[2] == 0.0f);
    inverse_host(sp.sensor_in_robot, S.Sinv); sincos_fixed(S.Sinv[2], S.sSinv, S.cSinv);
    S.fcan_offset = fcan_total; fcan_total += S.proj.cols; if (S.proj.cols > cols_max) cols_max = S.proj.cols;
  }
  A.cols_max = cols_max; A.fcan_total = fcan_total;
  A.cull_block = ctx->cull_block;
  A.cull = ctx->cull;      // (the test's column loop wraps once: canvases below 64 columns are not worth it and would need a second wrap)
  for (int s = 0; s < ns; ++s) if (b->slices[s].finder == LSM2D_FINDER_PROJECTIVE && A.s[s].proj.cols < 64) A.cull = 0;
  size_t lds = sizeof(u64) * (size_t) (cols_max + fcan_total + ((cols_max + fcan_total) & 1)) +
               sizeof(float) * kAccumWords * (kAlignBlock / 64);      // moving canvas, fixed canvases (padded to 16 bytes), reduction
  // one NN slice whose fixed clouds are scan-sized (one lane per query): stage each cloud's search tables in LDS.  Budget 38 KB
  // per workgroup keeps four workgroups on a CU; bigger clouds / grids search in global memory as before.  (Measured on configs[1]
  // role A: 3 sqrt(n) cells per side in 38 KB 9.6 ms; 4 sqrt(n) in 50 KB -- three workgroups per CU -- 12.2; 2 sqrt(n) 12.3.)
  // one NN, KD-tree or distance-map slice in the tracker's wiring (scan-sized fixed clouds, a big moving cloud): exact culling of the queries (k_align,
  // "point-query finders"): an occupancy bitmap of the fixed cloud (2 KB) and one keep bit per tile of 64 moving points, out of the same budget
  A.pq_cull_off = 0; A.pq_keep_words = 0;
  size_t pq_bytes = 0;
  if (ns == 1 && ctx->cull && (b->slices[0].finder == LSM2D_FINDER_NN || b->slices[0].finder == LSM2D_FINDER_KDTREE || b->slices[0].finder == LSM2D_FINDER_DISTMAP) &&
      !b->moving[0]->count_pending) {
    const lsm2d_cloudset* f = b->fixed[0]; const lsm2d_cloudset* m = b->moving[0];
    int mf = 0, mm = 0;
    for (int c = 0; c < f->n_clouds; ++c) if (f->h_count[c] > mf) mf = f->h_count[c];
    for (int c = 0; c < m->n_clouds; ++c) if (m->h_count[c] > mm) mm = m->h_count[c];
    const int keep_words = (((mm + 63) / 64 + kAlignBlock - 1) / kAlignBlock) * (kAlignBlock / 64);
    if (mf <= 16384 && mm >= 4096 && mm > 2 * mf && keep_words <= 512) {
      const int trc = ensure_tile_bounds(ctx, m); if (trc) return trc;
      if (m->d_tile_bounds) { A.pq_keep_words = keep_words; pq_bytes = 128 * 5 * 4 + sizeof(u64) * (size_t) keep_words + 16; A.s[0].moving = cloud_dev_with_tiles(A.s[0].moving, m); }
    }
  }
  const size_t lds_budget = 38 * 1024 - pq_bytes;
  A.nn_lds_points = 0; A.nn_lds_cells = 0;
  if (ns == 1 && b->slices[0].finder == LSM2D_FINDER_NN && A.s[0].nn_group == 1) {
    const lsm2d_cloudset* f = b->fixed[0];
    int mf = 0; for (int c = 0; c < f->n_clouds; ++c) if (f->h_count[c] > mf) mf = f->h_count[c];
    int cap = (int) ceil((mf >= 16384 ? 6.0 : 3.0) * sqrt((double) (mf > 0 ? mf : 1))); cap = cap < 16 ? 16 : cap;      // ensure_grid's rule
    const size_t need = sizeof(float2) * (size_t) mf + sizeof(uint16_t) * ((size_t) cap * cap + 4) + sizeof(uint16_t) * ((size_t) mf + 2);
    if (mf > 0 && mf <= 65535 && lds + need <= lds_budget) { A.nn_lds_points = mf; A.nn_lds_cells = cap * cap + 1; lds += need + 16; }
  }
  // one NN slice whose tables stay in global memory (the map is the fixed cloud): every query's cell and the candidate ranges of its 3 x 3 block are
  // cached in LDS from one iteration to the next (32 bytes per query of the biggest moving cloud), same budget
  A.nn_qcache = 0;
  if (ns == 1 && b->slices[0].finder == LSM2D_FINDER_NN && A.nn_lds_points == 0 && ctx->nn_qcache) {
    const lsm2d_cloudset* m = b->moving[0];
    int mm = 0; for (int c = 0; c < m->n_clouds; ++c) if (m->h_count[c] > mm) mm = m->h_count[c];
    lds = (lds + 15) & ~(size_t) 15;
    if (mm > 0 && lds + 32 * (size_t) mm + 16 <= lds_budget) { A.nn_qcache = mm; lds += 32 * (size_t) mm + 16; }
  }
  // one KD-tree slice: the top of the fixed cloud's tree (up to "kd_lds_nodes" nodes, 24 bytes each) rides in LDS -- same 38 KB budget
  A.kd_lds_nodes = 0; A.kd_lds_points = 0;
  if (ns == 1 && b->slices[0].finder == LSM2D_FINDER_KDTREE && kd_cache0 && ctx->kd_lds_nodes > 0) {
    int k = kd_cache0->max_nodes_per_cloud < ctx->kd_lds_nodes ? kd_cache0->max_nodes_per_cloud : ctx->kd_lds_nodes;
    const size_t room = lds < lds_budget ? (lds_budget - lds) / (sizeof(float4) + sizeof(int2)) : 0;
    if ((size_t) k > room) k = (int) room;
    if (k > 0) { A.kd_lds_nodes = k; lds += (size_t) k * (sizeof(float4) + sizeof(int2)) + 16; }
    // scan-sized fixed clouds whose whole tree fits: the leaf arrays too (coordinates and normals: 16 bytes per point)
    const lsm2d_cloudset* f = b->fixed[0];
    int mf = 0; for (int c = 0; c < f->n_clouds; ++c) if (f->h_count[c] > mf) mf = f->h_count[c];
    const size_t need = (size_t) (mf + 2) * (sizeof(float2) + sizeof(float2)) + 32;
    if (k > 0 && k == kd_cache0->max_nodes_per_cloud && mf > 0 && mf <= 65535 && lds + need <= lds_budget) { A.kd_lds_points = mf; lds += need; }
  }
  if (pq_bytes) { lds = (lds + 15) & ~(size_t) 15; A.pq_cull_off = (int32_t) lds; lds += pq_bytes; }
  ctx->last_query_cull = A.pq_cull_off > 0;
  // the projective instantiation with the culled stream only: every slice's moving set has its lane-chunked copy and chunk circles, and culling is on
  // (its unit lists -- kCullBlocks x 512 16-bit entries per slice, kept across iterations -- sit behind everything else in dynamic LDS; "cull_block" is the
  // round-3 stream's tuning knob: a batch that sets it runs the shared instantiation)
  bool proj_culled_for_all = has_proj && !has_nn && !has_dist && !has_kd && A.cull == 1 && ctx->proj_modes && ctx->cull_block == 0;
  for (int s = 0; s < ns && proj_culled_for_all; ++s)
    proj_culled_for_all = A.s[s].moving.lane_xy != nullptr && A.s[s].moving.lane_bounds != nullptr && A.s[s].moving.block_bounds != nullptr;
  A.units_off = 0; A.cull_keep = ctx->cull_keep;
  A.cull_mt = ctx->cull_keep ? 1e-6f * (float) ctx->cull_margin_um : 0.0f; A.cull_mth = ctx->cull_keep ? 1e-6f * (float) ctx->cull_margin_urad : 0.0f; A.cull_mt2 = A.cull_mt * A.cull_mt;      // (lists rebuilt every iteration: no margins)
  if (proj_culled_for_all) {
    int nb_max = kCullBlocks;
    for (int s = 0; s < ns; ++s) if (b->moving[s]->block_stride > nb_max) nb_max = b->moving[s]->block_stride;
    A.units_stride = nb_max * kAlignBlock;
    const size_t at = (lds + 15) & ~(size_t) 15, need = sizeof(uint16_t) * (size_t) ns * (size_t) A.units_stride;
    if ((int) (at + need) + 2048 <= ctx->max_dyn_lds && at + need + 2048 <= 40 * 1024) { A.units_off = (int32_t) at; lds = at + need; }      // four workgroups per CU must still fit (160 KB)
    else proj_culled_for_all = false;
  }
  // "sum_order" 1: the trip's pair records (lsm2d_device.h), behind everything else -- three workgroups per CU instead of four
  A.seq_off = 0;
  if (ctx->sum_order) { lds = (lds + 15) & ~(size_t) 15; A.seq_off = (int32_t) lds; lds += kSeqLdsBytes; }
  // the XCD window (AlignArgs::xcd_sync): a big-map batch of ONE dispatch round -- every workgroup resident from the start (64 VGPRs, <= 40 KB of LDS: four per
  // CU) -- whose position space fits the counters
  A.xcd_sync = nullptr; A.xcd_window = 0; A.xcd_stride = 0; A.xcd_positions = 0;
  bool xcd_on = false;
  if (kExperiments && proj_culled_for_all && ctx->xcd_lockstep > 0 && n > 1 && n <= 4 * ctx->n_cu && !out_work) {
    bool big = false;
    for (int s = 0; s < ns; ++s) big = big || b->moving[s]->block_stride == kCullBlocksMax;
    const long long positions = (long long) it_cap * ns;      // one per (iteration, slice) pass
    if (big && positions > 0 && positions <= 65536) { xcd_on = true; A.xcd_positions = (int32_t) positions; A.xcd_stride = (int32_t) ((16 + positions + 63) & ~63ll); A.xcd_window = ctx->xcd_lockstep - 1; }
  }
  // the NN instantiation without the search in global memory: the staging holds every alignment's tables (sized for the largest fixed cloud above), and no
  // alignment takes the cooperative loop, which searches in global memory (the kernel's rule: fixed cloud >= 4 x moving cloud) -- whatever the pairing
  bool nn_lds_for_all = false;
  if (A.nn_lds_points > 0 && ctx->nn_lds_only && !b->moving[0]->count_pending && !b->fixed[0]->count_pending) {
    const lsm2d_cloudset* f = b->fixed[0]; const lsm2d_cloudset* m = b->moving[0];
    long long mf = 0, mn = 0x7fffffff;
    for (int c = 0; c < f->n_clouds; ++c) if (f->h_count[c] > mf) mf = f->h_count[c];
    for (int c = 0; c < m->n_clouds; ++c) if (m->h_count[c] < mn) mn = m->h_count[c];
    nn_lds_for_all = m->n_clouds > 0 && mf < 4 * mn;
  }
  if ((int) lds + 512 > ctx->max_dyn_lds) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "align_batch: canvases do not fit LDS");
  // one or two projective slices and too few alignments to fill the chip (the live tracker: one alignment per scan): the latency
  // kernel (k_align_pair; bit-identical sums) -- 512 threads per slice, two slices' passes side by side instead of one after the
  // other, registers to spare for the serial solve step.  Measured against k_align on single-slice calls (tools/latency_kernel_ab.py):
  // 1 scan vs 10k points 0.163 -> 0.146 ms, vs a 700-point clipped scene with prior 0.063 -> 0.045, 256 candidates 0.172 -> 0.154
  // its LDS: fixed winners and canvases as k_align, a moving canvas and a block of wave totals per slice, and -- room permitting --
  // the moving clouds themselves (kPairMovCap points of 16 bytes per slice)
  const size_t lds_pair0 = sizeof(float4) * (size_t) fcan_total + sizeof(u64) * (size_t) fcan_total +
                           (size_t) ns * (sizeof(u64) * (size_t) cols_max + sizeof(float) * kPairRedStride * (kAlignBlock / 64));
  const size_t lds_mov = (size_t) ns * kPairMovCap * sizeof(float4);
  A.pair_mov_cap = (int) (lds_pair0 + lds_mov) + 512 <= ctx->max_dyn_lds ? kPairMovCap : 0;
  // ... and the fixed clouds (sizes the host knows, or upper bounds of sizes only the device knows: the kernel compares the real ones)
  int max_fixed_rows = 0;
  for (int s = 0; s < ns; ++s) { const lsm2d_cloudset* f = b->fixed[s]; for (int c = 0; c < f->n_clouds; ++c) if (f->h_count[c] > max_fixed_rows) max_fixed_rows = f->h_count[c]; }
  max_fixed_rows = (max_fixed_rows + 63) & ~63;
  const size_t lds_fix = (size_t) ns * (size_t) max_fixed_rows * sizeof(float4);
  A.pair_fix_cap = max_fixed_rows <= 4096 && (int) (lds_pair0 + (A.pair_mov_cap ? lds_mov : 0) + lds_fix) + 512 <= ctx->max_dyn_lds ? max_fixed_rows : 0;
  const size_t lds_pair = lds_pair0 + (A.pair_mov_cap ? lds_mov : 0) + (A.pair_fix_cap ? lds_fix : 0);
  // ("sum_order" 1: the latency kernel keeps the tree order -- such calls take k_align_seq, one workgroup per alignment)
  const bool use_pair = !ctx->sum_order && !use_split && ctx->align_path != 1 && (ns == 1 || ns == 2) && has_proj && !has_nn && !has_dist && !has_kd &&
                        (n <= 256 || ctx->align_path == 3) && ap->max_iterations > 0 && (int) lds_pair + 512 <= ctx->max_dyn_lds;

  // ---- inputs
  memcpy(hs + o_pose_in, b->init_pose, sizeof(float) * 3 * (size_t) n);
  if (b->prior) {
    PriorDev* p = (PriorDev*) (hs + o_prior);
    for (int i = 0; i < n; ++i) {
      inverse_host(b->prior[i].z, p[i].z_inv); sincos_fixed(p[i].z_inv[2], p[i].sz, p[i].cz);
      memcpy(p[i].omega, b->prior[i].omega, sizeof(float) * 9);
    }
    A.prior = (const PriorDev*) (ds + o_prior);
  }
  // one alignment (the live tracker's call): start pose and prior ride in the kernel arguments, so the kernel's prologue does not
  // wait for a read of host memory
  A.inline_n1 = n == 1 && !use_split;
  if (A.inline_n1) { memcpy(A.pose1, b->init_pose, sizeof A.pose1); if (b->prior) memcpy(&A.prior1, hs + o_prior, sizeof A.prior1); }
  // the stream this batch's own operations go to: its lane's when it was begun asynchronously (lane_stream) -- not for the split path (one workspace per context)
  // nor the experiments' XCD counters.  Everything queued so far went to the context's stream (set preparation: pending preprocessing, trees, lane copies): an
  // event behind it orders the lane's stream and the pre-kernels' stream
  const hipStream_t ks = (async && !out_work && !use_split && !xcd_on) ? lane_stream(ctx) : ctx->stream;
  if (ks != ctx->stream && pre != ctx->stream) {
    HIPCHK(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(pre, ctx->ev_main, 0));
  }
  // Round 5: a batch that comes again with the SAME input block (start poses, index arrays; no priors) while nothing else has touched the lane's scratch needs no
  // upload: the blit and the wait behind it are 8 us of a 0.77 ms step (kernel trace: copy 2.4 us + 5.6 us until k_align starts).  Compared byte for byte against a
  // host-side shadow of what was uploaded last -- up to 64 KB; a sweep's index arrays beyond that are uploaded as before.
  const bool same_inputs = !zero_copy && !out_work && !b->prior && had_inputs && in_bytes <= (64u << 10) && ctx->inputs_shadow.size() == in_bytes &&
                           !memcmp(ctx->inputs_shadow.data(), hs, in_bytes);
  if (!zero_copy && !same_inputs) {
    HIPCHK(ctx, hipMemcpyAsync(ds, hs, in_bytes, hipMemcpyHostToDevice, pre));
    if (!out_work && !b->prior && in_bytes <= (64u << 10)) ctx->inputs_shadow.assign((const unsigned char*) hs, (const unsigned char*) hs + in_bytes); else ctx->inputs_shadow.clear();
  }
  ctx->inputs_valid = !zero_copy && !ctx->inputs_shadow.empty();
  A.init_pose = (const float*) (ds + o_pose_in);
  if (xcd_on && !use_split && !use_pair && !zero_copy) {
    const size_t xb = sizeof(uint32_t) * 16 * (size_t) A.xcd_stride;
    if (xb > ctx->d_xcd_bytes) {
      if (ctx->d_xcd) { HIPCHK(ctx, stream_sync(ctx)); HIPCHK(ctx, hipFree(ctx->d_xcd)); ctx->d_xcd = nullptr; ctx->d_xcd_bytes = 0; }
      HIPCHK(ctx, hipMalloc((void**) &ctx->d_xcd, xb)); ctx->d_xcd_bytes = xb;
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->d_xcd, 0, xb, ctx->stream));
    A.xcd_sync = ctx->d_xcd;
  }
  ctx->last_xcd_lockstep = A.xcd_sync ? A.xcd_window + 1 : 0;
  // the SMALL results of a batch that travels by copies (56 bytes per alignment + the clock stamps) are written by the kernels straight into the pinned
  // staging buffer: the device-to-host copy behind the launch -- a hand-over to the copy engine, 9 us of gap + 6 us of copy on the timeline of a
  // 1000-alignment step -- is gone, the stream wait ends with the kernel.  The statistics (28 bytes per iteration and alignment) stay on the device and are copied.
  const bool host_results = !zero_copy && !use_split && ctx->results_to_host && n > 0;
  char* ro = host_results ? (char*) ctx->h_stage_dev : ds;
  A.out_pose = (float*) (ro + o_pose); A.out_H = (float*) (ro + o_H); A.out_status = (int32_t*) (ro + o_status); A.out_its = (int32_t*) (ro + o_its);
  A.out_stats = out_stats ? (StatsDev*) (ds + o_stats) : nullptr;
  A.out_last_pose = out_last_pose ? (float*) (ro + o_last_pose) : nullptr;

  ctx->last_clock_khz = 0; ctx->last_wg_lifetime_ns = 0;
  const bool stamps = ctx->kernel_timing && !use_split && !use_pair;
  if (stamps) { A.clock_out = (unsigned long long*) ((host_results ? (char*) ctx->h_stage_dev : ds) + o_clock); A.clock_stride = clock_stride; }
  A.host_polls = zero_copy;
  if (zero_copy) {
    memset(hs + o_pose, 0, out_bytes);
    int32_t* st = (int32_t*) (hs + o_status);
    for (int i = 0; i < n; ++i) st[i] = kStatusNotWritten;          // the kernels write an alignment's status last (release, system scope)
  }
  else {
    // every alignment's pose, information matrix, status and iteration count are written by its workgroup whatever happens to it; what a kernel may leave
    // untouched are the statistics of iterations that never started: only those are cleared (the clock stamps of a timed launch are written by every
    // stamping workgroup -- each alignment runs exactly once, wherever the placement puts it)
    if (out_stats) HIPCHK(ctx, hipMemsetAsync(ds + o_stats, 0, sizeof(StatsDev) * (size_t) n * (size_t) stats_stride, ks));
    // ... and "exactly once" is checked, not assumed (round-4 advisor): the status words start as kStatusNotWritten -- in the pinned buffer the kernels write to, a
    // host memset of 4 n bytes; on the device for the paths that copy -- and one that is still unwritten after the wait turns the call into LSM2D_DEVICE_ERROR
    if (host_results) memset(hs + o_status, 0xFF, sizeof(int32_t) * (size_t) n);
    else HIPCHK(ctx, hipMemsetAsync(ds + o_status, 0xFF, sizeof(int32_t) * (size_t) n, ks));
  }
  ctx->last_align_path = use_split ? 2 : (use_pair ? 3 : 1);
  // culled batches that run in about one dispatch round: balanced placement (one small launch ahead of k_align; see k_cull_estimate)
  A.order = nullptr; A.wg_place = nullptr;
  A.cull_est_mt = 1e-6f * (float) ctx->cull_est_um; A.cull_est_mth = 1e-6f * (float) ctx->cull_est_urad;
  if (out_work) {      // lsm2d_estimate_work: the chunks of the moving cloud each alignment's FIRST iteration will stream (k_cull_estimate), or 1 everywhere
    int bs = -1;
    for (int s = 0; s < ns && bs < 0; ++s) if (A.s[s].finder == LSM2D_FINDER_PROJECTIVE && A.s[s].moving.lane_xy && A.s[s].moving.lane_bounds && A.cull) bs = s;
    if (bs < 0) { for (int i = 0; i < n; ++i) out_work[i] = 1; return LSM2D_SUCCESS; }
    int32_t* d_work = (int32_t*) ((char*) ctx->d_scratch + o_work);
    hipLaunchKernelGGL(k_cull_estimate, dim3((unsigned) n), dim3(kAlignBlock), sizeof(u64) * (size_t) A.s[bs].proj.cols, ctx->stream, A, bs, d_work, (int32_t*) nullptr, (const int32_t*) nullptr, ctx->n_cu, (unsigned int*) nullptr);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(hs + o_work, d_work, sizeof(int32_t) * (size_t) n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, stream_sync(ctx));
    memcpy(out_work, hs + o_work, sizeof(int32_t) * (size_t) n);
    return LSM2D_SUCCESS;
  }
  A.stage = 0; A.stage_split = 0; A.resume = nullptr; A.stage_work = nullptr;
  // a culled batch of about one dispatch round, two launches: iteration 0 anywhere (k_first_iteration), then the rest placed by what iteration 1's lists hold
#ifdef LSM2D_EXPERIMENTS
  const bool two_stage = !use_split && !use_pair && !zero_copy && A.cull && ctx->balance && ctx->two_stage && n > 256 && n <= 1024 && proj_culled_for_all &&
                         has_proj && !has_nn && !has_dist && !has_kd && ap->max_iterations >= 4;
  if (two_stage) {
    int32_t* d_work = (int32_t*) ((char*) ctx->d_scratch + o_work); int32_t* d_order = (int32_t*) ((char*) ctx->d_scratch + o_order);
    const unsigned long long shape = ((unsigned long long) (unsigned) n << 32) ^ ((unsigned long long) lds << 8) ^ 6ull;
    if (!ctx->d_wg_place) {
      HIPCHK(ctx, hipMalloc(&ctx->d_wg_place, sizeof(int32_t) * 1025)); ctx->wg_place_shape = 0;
      HIPCHK(ctx, hipMemsetAsync(ctx->d_wg_place, 0, sizeof(int32_t) * 1025, ctx->stream));
    }
    const bool notes = ctx->balance_notes && ctx->wg_place_shape == shape;
    A.stage = 1; A.stage_split = 1; A.resume = (ResumeDev*) ((char*) ctx->d_scratch + o_resume); A.stage_work = d_work;
    hipLaunchKernelGGL(k_first_iteration, dim3((unsigned) n), dim3(kAlignBlock), lds, ctx->stream, A);
    hipLaunchKernelGGL(k_balance_only, dim3(1), dim3(kAlignBlock), sizeof(BalanceLds), ctx->stream, (const int32_t*) d_work, n, ctx->n_cu, d_order,
                       notes ? (const int32_t*) ctx->d_wg_place : (const int32_t*) nullptr);
    HIPCHK(ctx, hipGetLastError());
    A.stage = 2; A.order = d_order;
    if (ctx->balance_notes) { A.wg_place = ctx->d_wg_place; ctx->wg_place_shape = shape; }
  }
  else
#endif
  // (a batch of many dispatch rounds balances itself, and the estimate of 65 536 alignments costs more than its heaviest-first order saves:
  // configs[3] 47.6 vs 47.2 ms per step -- the placement is for batches of up to four rounds)
  ctx->last_cull_estimate = 0;
  if (!use_split && !use_pair && !zero_copy && A.cull && ctx->balance && n > 256 && n <= 4096 && has_proj) {
    int bs = -1;
    for (int s = 0; s < ns && bs < 0; ++s) if (A.s[s].finder == LSM2D_FINDER_PROJECTIVE && A.s[s].moving.lane_xy && A.s[s].moving.lane_bounds) bs = s;
    if (bs >= 0) {
      int32_t* d_work = (int32_t*) ((char*) ctx->d_scratch + o_work);
      // the workgroups of the previous launch of the same shape noted the CU they ran on (AlignArgs::wg_place): the placement groups by those notes
      const unsigned long long shape = ((unsigned long long) (unsigned) n << 32) ^ ((unsigned long long) lds << 8) ^ (unsigned long long) (proj_culled_for_all ? 5 : 0);
      if (!ctx->d_wg_place) {      // 1024 notes + the estimate's ticket counter
        HIPCHK(ctx, hipMalloc(&ctx->d_wg_place, sizeof(int32_t) * 1025)); ctx->wg_place_shape = 0;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_wg_place, 0, sizeof(int32_t) * 1025, pre));      // (on the stream the estimate that reads the ticket is queued on)
      }
      if (!ctx->d_order) { HIPCHK(ctx, hipMalloc(&ctx->d_order, sizeof(int32_t) * 4096)); ctx->order_valid = false; }
      const bool notes = ctx->balance_notes && ctx->wg_place_shape == shape;
      // Round 5: the order lives in a buffer of its own and is KEPT.  A caller that runs the same batch again -- the same sets (uid and version), index arrays,
      // slice parameters, launch shape and START POSES: a candidate sweep re-scored, bench.py's resident step -- gets the placement made for it the first time
      // with notes, and no estimate launch (33 us of a 0.8 ms step).  Anything that differs makes it afresh; only where alignments run depends on it, never a result.
      unsigned long long key = 1469598103934665603ull;
      auto mix = [&](const void* p, size_t bytes) { const unsigned char* q = (const unsigned char*) p; for (size_t i = 0; i < bytes; ++i) { key ^= q[i]; key *= 1099511628211ull; } };
      mix(&shape, sizeof shape); mix(&bs, sizeof bs); mix(&A.cull_est_mt, 4); mix(&A.cull_est_mth, 4); mix(&ns, sizeof ns);
      for (int s = 0; s < ns; ++s) {
        const unsigned long long id[4] = {b->fixed[s]->uid, b->fixed[s]->version, b->moving[s]->uid, b->moving[s]->version};
        mix(id, sizeof id); mix(&b->slices[s], sizeof(lsm2d_slice_params));
        if (b->fixed_index) mix(b->fixed_index + (size_t) s * n, sizeof(int32_t) * (size_t) n);
        if (b->moving_index) mix(b->moving_index + (size_t) s * n, sizeof(int32_t) * (size_t) n);
      }
      const bool reuse = ctx->estimate_reuse && notes && ctx->order_valid && ctx->order_key == key && ctx->order_poses.size() == 3 * (size_t) n &&
                         !memcmp(ctx->order_poses.data(), b->init_pose, sizeof(float) * 3 * (size_t) n);
      // A batch begun while another one is in flight starts on the slots that one's tail leaves free -- wherever they are: the launch balances itself as a batch
      // of many dispatch rounds does, and what is left of the placement's gain (2 % with a kept order) is less than the estimate's own chip time when it has to be made
      // afresh (streamed pipeline 0.685 against 0.696 ms per step): no estimate then, workgroup b = alignment b.  A kept order is still used -- and a batch this
      // lane has seen before (same sets, versions, parameters: a caller that runs it again and again) gets its estimate once, to be kept from then on.
      const bool joins_a_batch_in_flight = ks != ctx->stream && ctx->inflight >= 1 && ctx->lane_streams;
      if (!reuse && joins_a_batch_in_flight && !(ctx->estimate_reuse && ctx->order_key == key)) { ctx->order_valid = false; ctx->order_key = key; }
      else {
      if (!reuse) {
        size_t est_lds = sizeof(u64) * (size_t) A.s[bs].proj.cols; if (est_lds < sizeof(BalanceLds)) est_lds = sizeof(BalanceLds);
        // (estimates share ONE ticket counter: one queued on the second stream waits for the latest one queued on the first)
        if (pre != ctx->stream && ctx->a_est_recorded) HIPCHK(ctx, hipStreamWaitEvent(pre, ctx->ev_a_est, 0));
        if (pre == ctx->stream && ctx->b_recorded) HIPCHK(ctx, hipStreamWaitEvent(pre, ctx->ev_b, 0));      // (... and the other way round: a synchronous call while a begun batch's estimate may still be running)
        hipLaunchKernelGGL(k_cull_estimate, dim3((unsigned) n), dim3(kAlignBlock), est_lds, pre, A, bs, d_work, ctx->d_order,
                           notes ? (const int32_t*) ctx->d_wg_place : (const int32_t*) nullptr, ctx->n_cu, (unsigned int*) (ctx->d_wg_place + 1024));
        const hipError_t le = hipGetLastError();
        if (le == hipSuccess && pre == ctx->stream) { HIPCHK(ctx, hipEventRecord(ctx->ev_a_est, ctx->stream)); ctx->a_est_recorded = true; }
        if (le != hipSuccess) {      // (round-4 advisor) a launch that failed may have left the ticket counter mid-count: the next call must not start mis-counted
          (void) hipMemsetAsync(ctx->d_wg_place + 1024, 0, sizeof(int32_t), ctx->stream); ctx->order_valid = false;
          HIPCHK(ctx, le);
        }
        ctx->last_cull_estimate = 1;
        // (kept only once it was made WITH notes: the first call of a shape orders by the round-3 assumption, the second by what the first really did)
        ctx->order_valid = notes; ctx->order_key = key;
        if (notes) ctx->order_poses.assign(b->init_pose, b->init_pose + 3 * (size_t) n);
      }
      A.order = ctx->d_order;
      if (ctx->balance_notes && n <= 1024) { A.wg_place = ctx->d_wg_place; ctx->wg_place_shape = shape; }
      }
    }
  }
  if (ks != ctx->stream) {      // ... and the lane's stream behind everything the context's own stream was given up to here (set preparation; with nothing in flight also this batch's start poses and estimate)
    HIPCHK(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ks, ctx->ev_main, 0));
  }
  HIPCHK(ctx, join_pre_stream(ctx, ks));      // whatever the second stream holds for this batch (its start poses, its estimate) comes first
  if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev0, ks));
  if (use_split) {
    // workspace: global canvases + running pose / flags, grown on demand and kept by the context
    const size_t can_bytes = sizeof(u64) * 2 * (size_t) fcan_total * (size_t) n;
    const size_t w_pose = (can_bytes + 255) & ~(size_t) 255, w_done = w_pose + (((sizeof(float) * 3 * (size_t) n) + 255) & ~(size_t) 255);
    const size_t w_H = w_done + (((sizeof(int32_t) * (size_t) n) + 255) & ~(size_t) 255), w_last = w_H + (((sizeof(float) * 9 * (size_t) n) + 255) & ~(size_t) 255);
    const size_t w_phase = w_last + ((sizeof(StatsDev) * (size_t) n + 255) & ~(size_t) 255);
    const size_t w_total = w_phase + sizeof(int32_t) * 3 * (size_t) n;
    if (w_total > ctx->d_split_bytes) {
      if (ctx->d_split) { HIPCHK(ctx, stream_sync(ctx)); HIPCHK(ctx, hipFree(ctx->d_split)); ctx->d_split = nullptr; ctx->d_split_bytes = 0; }
      HIPCHK(ctx, hipMalloc(&ctx->d_split, w_total + w_total / 2));
      ctx->d_split_bytes = w_total + w_total / 2;
    }
    char* w = (char*) ctx->d_split;
    SplitArgs SA; SA.A = A;
    SA.gcan = (u64*) w; SA.pose = (float*) (w + w_pose); SA.done = (int32_t*) (w + w_done); SA.H_last = (float*) (w + w_H); SA.last = (StatsDev*) (w + w_last);
    SA.phase = (int32_t*) (w + w_phase);
    HIPCHK(ctx, hipMemsetAsync(SA.gcan, 0xFF, can_bytes, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(SA.done, 0, sizeof(int32_t) * (size_t) n, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(SA.phase, 0, sizeof(int32_t) * 3 * (size_t) n, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(SA.pose, A.init_pose, sizeof(float) * 3 * (size_t) n, hipMemcpyDeviceToDevice, ctx->stream));
    int max_fixed = 0;
    for (int s = 0; s < ns; ++s) { const lsm2d_cloudset* f = b->fixed[s]; for (int c = 0; c < f->n_clouds; ++c) if (f->h_count[c] > max_fixed) max_fixed = f->h_count[c]; }
    auto chunks_for = [&](int max_points) {
      int c = (max_points / 2 + 2047) / 2048;                 // >= 4 pairs per thread and chunk
      const int budget = 2048 / (n * ns) > 1 ? 2048 / (n * ns) : 1;
      if (c > budget) c = budget;
      return c < 1 ? 1 : c;
    };
    const size_t clds = sizeof(u64) * (size_t) cols_max;
    SA.it = 0;
    hipLaunchKernelGGL((k_split_project<true>), dim3((unsigned) chunks_for(max_fixed), (unsigned) n, (unsigned) ns), dim3(512), clds, ctx->stream, SA);
    const int mchunks = chunks_for(max_moving);
    for (int it = 0; it < it_cap; ++it) {      // (alignments that are done leave their launches at once: S.done)
      SA.it = it;
      hipLaunchKernelGGL((k_split_project<false>), dim3((unsigned) mchunks, (unsigned) n, (unsigned) ns), dim3(512), clds, ctx->stream, SA);
      if (ctx->sum_order) hipLaunchKernelGGL(k_split_finish<true>, dim3((unsigned) n), dim3(kAlignBlock), 0, ctx->stream, SA);
      else hipLaunchKernelGGL(k_split_finish<false>, dim3((unsigned) n), dim3(kAlignBlock), 0, ctx->stream, SA);
    }
  } else {
    const dim3 grid((unsigned) n), block(kAlignBlock);
    if (use_pair) hipLaunchKernelGGL(k_align_pair, grid, dim3((unsigned) (kAlignBlock * ns)), lds_pair, ks, A);
    else {
      // which instantiation: the finders the batch's slices use, and -- for a batch of ONE finder kind -- the form of its inner loop the host could prove
      // serves every alignment (kNNMode of align_body).  One table (round 5; a 12-way ladder before); the mixed instantiations take whatever is left.
      const unsigned finders = (has_proj ? kFProj : 0u) | (has_nn ? kFNN : 0u) | (has_dist ? kFDist : 0u) | (has_kd ? kFKd : 0u);
      int mode = 0;
      if (finders == kFProj) mode = proj_culled_for_all ? (A.xcd_sync ? 6 : 5) : 0;                                                    // every slice: the culled stream over kept unit lists
      else if (finders == kFNN) mode = A.nn_lds_points == 0 ? 1 : (nn_lds_for_all ? 2 : 0);                        // tables in global memory / in LDS for every alignment
      else if (finders == kFKd && ns == 1 && ctx->kd_modes) mode = A.kd_lds_points > 0 ? 3 : 4;                    // whole trees in LDS / only their tops
      AlignKernel fn = nullptr;
      if (ctx->sum_order) {      // the reference's order of summation: k_align_seq, the culled projective stream or the finder kind's general form
        if (mode != 5) mode = 0;
        for (const AlignVariant& v : kAlignVariantsSeq) if (v.finders == finders && v.mode == mode) { fn = v.fn; break; }
        if (!fn) fn = has_kd ? (AlignKernel) k_align_seq<true, true, true, true> : (AlignKernel) k_align_seq<true, true, true>;
      }
      else
      for (const AlignVariant& v : kAlignVariants) if (v.finders == finders && v.mode == mode) { fn = v.fn; break; }
      if (!fn) fn = has_kd ? (AlignKernel) k_align<true, true, true, true> : (AlignKernel) k_align<true, true, true>;      // mixed finders
      hipLaunchKernelGGL(fn, grid, block, lds, ks, A);
    }
  }
  HIPCHK(ctx, hipGetLastError());
  for (int s = 0; s < ns; ++s)                  // sets the kernel's prologue unpacks (SliceDev::unpack_src)
    if (A.s[s].unpack_src) { b->fixed[s]->unpack_pending = false; b->fixed[s]->staged_epoch = ctx->sync_epoch; }
  if (ctx->kernel_timing) HIPCHK(ctx, hipEventRecord(ctx->ev1, ks));
  ctx->have_timing = ctx->kernel_timing;
  if (host_results) { if (out_stats) HIPCHK(ctx, hipMemcpyAsync(hs + o_stats, ds + o_stats, sizeof(StatsDev) * (size_t) n * (size_t) stats_stride, hipMemcpyDeviceToHost, ks)); }
  else if (!zero_copy) HIPCHK(ctx, hipMemcpyAsync(hs + o_pose, ds + o_pose, out_bytes, hipMemcpyDeviceToHost, ks));
  // ---- the batch is queued.  What its results need: kept in a lsm2d_pending (the caller's for an asynchronous begin, a local one otherwise)
  lsm2d_pending local; lsm2d_pending& P = pend ? *pend : local;
  P.ctx = ctx; P.lane_id = ctx->lane_id; P.ev_done = ctx->ev_done; P.ev0 = ctx->ev0; P.ev1 = ctx->ev1; P.hs = hs;
  P.o_pose = o_pose; P.o_H = o_H; P.o_status = o_status; P.o_its = o_its; P.o_stats = o_stats; P.o_last_pose = o_last_pose; P.o_clock = o_clock;
  P.n = n; P.stats_stride = stats_stride; P.n_clock = n_clock; P.clock_stride = clock_stride;
  P.zero_copy = zero_copy; P.want_stats = out_stats != nullptr; P.want_last_pose = out_last_pose != nullptr; P.stamps = stamps; P.timed = ctx->kernel_timing != 0; P.async = async;
  P.xcd_sync = A.xcd_sync; P.xcd_stride = A.xcd_stride; P.xcd_window = A.xcd_window; P.xcd_positions = A.xcd_positions;
  if (async) {
    HIPCHK(ctx, hipEventRecord(ctx->ev_done, ks));      // (zero-copy batches too: what their wait falls back to when the statuses do not arrive within the spin budget)
    ctx->lane_busy = true; ++ctx->inflight;
    swap_lanes(ctx);      // whatever is called next works on the other lane
    return LSM2D_SUCCESS;
  }
  return align_batch_finish(P, out_pose, out_H, out_status, out_its, out_stats, out_last_pose);
}

// the second half: wait for the batch, check that every alignment reported, hand the results over
static int align_batch_finish(lsm2d_pending& P, float* out_pose, float* out_H, int32_t* out_status, int32_t* out_its, lsm2d_iteration_stats* out_stats, float* out_last_pose) {
  lsm2d_context* ctx = P.ctx;
  char* hs = P.hs; const int n = P.n;
  const size_t o_pose = P.o_pose, o_H = P.o_H, o_status = P.o_status, o_its = P.o_its, o_stats = P.o_stats, o_last_pose = P.o_last_pose, o_clock = P.o_clock;
  const int stats_stride = P.stats_stride, n_clock = P.n_clock, clock_stride = P.clock_stride;
  const bool zero_copy = P.zero_copy, stamps = P.stamps;
  if (P.async) {
    // (an event of the batch's own, not a wait for the stream: the NEXT batch may be queued behind it already.  The stream's epoch does not move: work queued
    // after this batch has not necessarily run)
    hipError_t we = hipSuccess;
    if (zero_copy) { const unsigned long long epoch = ctx->sync_epoch; we = wait_for_statuses(ctx, (const int32_t*) (hs + o_status), n, P.ev_done); ctx->sync_epoch = epoch; }
    else we = hipEventSynchronize(P.ev_done);
    if (ctx->lane_id == P.lane_id) ctx->lane_busy = false; else if (ctx->parked.id == P.lane_id) ctx->parked.busy = false;
    if (ctx->inflight > 0) --ctx->inflight;
    HIPCHK(ctx, we);
  }
  else if (zero_copy) HIPCHK(ctx, wait_for_statuses(ctx, (const int32_t*) (hs + o_status), n));
  else HIPCHK(ctx, stream_sync(ctx));
  ctx->last_ev0 = P.ev0; ctx->last_ev1 = P.ev1; ctx->have_timing = P.timed;
  {
    static_assert(kStatusNotWritten == -1, "the memsets above write 0xFF bytes");
    const int32_t* st = (const int32_t*) (hs + o_status);
    for (int i = 0; i < n; ++i) if (st[i] == kStatusNotWritten) { ctx->order_valid = false; ctx->parked.order_valid = false; return fail(ctx, LSM2D_DEVICE_ERROR, "align_batch: an alignment's workgroup never reported (placement or launch fault)"); }
  }
  memcpy(out_pose, hs + o_pose, sizeof(float) * 3 * (size_t) n);
  if (out_H) memcpy(out_H, hs + o_H, sizeof(float) * 9 * (size_t) n);
  memcpy(out_status, hs + o_status, sizeof(int32_t) * (size_t) n);
  if (out_its) memcpy(out_its, hs + o_its, sizeof(int32_t) * (size_t) n);
  if (out_stats) memcpy(out_stats, hs + o_stats, sizeof(StatsDev) * (size_t) n * (size_t) stats_stride);
  if (out_last_pose) memcpy(out_last_pose, hs + o_last_pose, sizeof(float) * 3 * (size_t) n);
  if (P.xcd_sync) if (const char* dump = getenv("LSM2D_DUMP_XCD")) {      // diagnostics: the XCD window's counters after the launch, one line per XCC that took part
    std::vector<uint32_t> h((size_t) 16 * P.xcd_stride);
    if (hipMemcpy(h.data(), P.xcd_sync, sizeof(uint32_t) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) if (FILE* f = fopen(dump, "a")) {
      fprintf(f, "# launch n=%d lockstep=%d passes=%d\n", n, P.xcd_window + 1, P.xcd_positions);
      for (int x = 0; x < 16; ++x) {
        const uint32_t* c = h.data() + (size_t) x * P.xcd_stride;
        if (!c[0]) continue;
        fprintf(f, "xcc %2d registered %u gone %u watchdog %u (last: need %u saw %u reg %u gone %u at g %u) done:", x, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
        for (int q = 0; q < P.xcd_positions && q < 48; ++q) fprintf(f, " %u", c[16 + q]);
        fprintf(f, " ... %u\n", c[16 + P.xcd_positions - 1]);
      }
      fclose(f);
    }
  }
  if (stamps) {     // median over the stamped workgroups: shader cycles per 10 ns tick of the constant 100 MHz counter
    const unsigned long long* ck = (const unsigned long long*) (hs + o_clock);
    std::vector<double> khz; std::vector<unsigned long long> life;
    for (int i = 0; i < n_clock; ++i) if (ck[4 * i + 1] > 0) { khz.push_back((double) ck[4 * i] / (double) ck[4 * i + 1] * 1e5); life.push_back(ck[4 * i + 1] * 10ull); }
    if (const char* dump = getenv("LSM2D_DUMP_STAMPS")) {      // diagnostics: one line per stamped workgroup (tools/occupancy_probe.py reads them)
      if (FILE* f = fopen(dump, "a")) {
        unsigned long long t0 = ~0ull; for (int i = 0; i < n_clock; ++i) if (ck[4 * i + 1] > 0 && ck[4 * i + 2] < t0) t0 = ck[4 * i + 2];
        fprintf(f, "# launch n=%d stride=%d\n", n, clock_stride);
        for (int i = 0; i < n_clock; ++i)
          fprintf(f, "%d %llu %llu %llu 0x%llx\n", i * clock_stride, ck[4 * i], ck[4 * i + 1], ck[4 * i + 2] - t0, ck[4 * i + 3]);
        fclose(f);
      }
    }
    if (!khz.empty()) {
      std::nth_element(khz.begin(), khz.begin() + khz.size() / 2, khz.end()); ctx->last_clock_khz = (long long) khz[khz.size() / 2];
      std::nth_element(life.begin(), life.begin() + life.size() / 2, life.end()); ctx->last_wg_lifetime_ns = (long long) life[life.size() / 2];
    }
  }
  return LSM2D_SUCCESS;
}

extern "C" int lsm2d_align_batch(lsm2d_context* ctx, const lsm2d_aligner_params* ap, const lsm2d_batch* b, float* out_pose,
                                 float* out_H, int32_t* out_status, int32_t* out_its, lsm2d_iteration_stats* out_stats) {
  return align_batch_impl(ctx, ap, b, out_pose, out_H, out_status, out_its, out_stats, nullptr);
}

// ---- a batch in flight: MultiAligner2D::compute over a batch, split where the host would otherwise sleep.  begin() queues everything -- inputs, placement,
// kernels, the copies of the results -- and returns; wait() blocks until THAT batch's last operation has run (an event of its own: a younger batch may be
// queued behind it) and hands the results over.  While one batch is in flight, the next one's pre-kernels go to a second stream (pre_stream) and fill the
// slots its tail leaves free.  Two lanes of staging / scratch: at most two batches in flight, waited for in the order they were begun.
extern "C" int lsm2d_align_batch_begin(lsm2d_context* ctx, const lsm2d_aligner_params* ap, const lsm2d_batch* b, int32_t want_stats, lsm2d_pending** out_pending) {
  if (!ctx || !out_pending) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch_begin: null argument");
  *out_pending = nullptr;
  lsm2d_pending* P = new (std::nothrow) lsm2d_pending;
  if (!P) return LSM2D_OUT_OF_MEMORY;
  static float dummy_pose; static int32_t dummy_status; static lsm2d_iteration_stats dummy_stats;      // (only their being non-null is looked at)
  const int rc = align_batch_impl(ctx, ap, b, &dummy_pose, nullptr, &dummy_status, nullptr, want_stats ? &dummy_stats : nullptr, nullptr, nullptr, P);
  if (rc != LSM2D_SUCCESS || !P->ctx) { const bool empty = rc == LSM2D_SUCCESS; delete P; if (!empty) return rc; P = new (std::nothrow) lsm2d_pending; if (!P) return LSM2D_OUT_OF_MEMORY; }      // (n == 0: an empty batch, nothing in flight)
  *out_pending = P;
  return LSM2D_SUCCESS;
}
extern "C" int lsm2d_align_batch_wait(lsm2d_pending* pending, float* out_pose, float* out_H, int32_t* out_status, int32_t* out_its, lsm2d_iteration_stats* out_stats) {
  if (!pending) return LSM2D_BAD_ARGUMENT;
  if (!pending->ctx) { delete pending; return LSM2D_SUCCESS; }      // an empty batch
  lsm2d_context* ctx = pending->ctx;
  int rc = LSM2D_SUCCESS;
  if (!out_pose || !out_status || (out_stats && !pending->want_stats)) {
    // the batch must still be retired: its lane stays busy otherwise
    float* p = (float*) malloc(sizeof(float) * 3 * (size_t) (pending->n > 0 ? pending->n : 1)); int32_t* st = (int32_t*) malloc(sizeof(int32_t) * (size_t) (pending->n > 0 ? pending->n : 1));
    if (p && st) (void) align_batch_finish(*pending, p, nullptr, st, nullptr, nullptr, nullptr);
    free(p); free(st);
    rc = fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch_wait: out_pose / out_status missing, or statistics asked for that the batch was not begun with");
  }
  else rc = align_batch_finish(*pending, out_pose, out_H, out_status, out_its, out_stats, nullptr);
  delete pending;
  return rc;
}

extern "C" int lsm2d_estimate_work(lsm2d_context* ctx, const lsm2d_batch* b, int32_t* out_work) {
  if (!out_work) return fail(ctx, LSM2D_BAD_ARGUMENT, "estimate_work: null argument");
  lsm2d_aligner_params ap; memset(&ap, 0, sizeof ap); ap.max_iterations = 1;
  return align_batch_impl(ctx, &ap, b, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out_work);
}

// MultiAligner2D::compute + what it leaves in the slices' correspondence vectors (apps/visual_test_aligner_2d.cpp:129-143).  The aligner kernels keep
// no pair lists (their pairs live for one bin walk / one query); the vectors are re-derived from the pose the last started iteration began at by
// the finder-level kernels -- the same arithmetic, hence the same pairs (tests: their digest equals the in-kernel one) -- one finder pass per
// alignment and slice: an observability surface (40-90 us per pass), not a throughput path.
extern "C" int lsm2d_align_batch_pairs(lsm2d_context* ctx, const lsm2d_aligner_params* ap, const lsm2d_batch* b, float* out_pose, float* out_H,
                                       int32_t* out_status, int32_t* out_its, lsm2d_iteration_stats* out_stats,
                                       lsm2d_correspondence* out_pairs, int32_t pair_capacity, int32_t* out_n_pairs) {
  if (!out_pairs) return align_batch_impl(ctx, ap, b, out_pose, out_H, out_status, out_its, out_stats, nullptr);
  if (!ctx || !ap || !b || !out_n_pairs || pair_capacity < 0) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch_pairs: bad argument");
  const int n = b->n_alignments, ns = b->n_slices;
  if (n < 0 || ns < 1 || ns > kMaxSlices || !b->slices || !b->fixed || !b->moving) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch_pairs: bad batch descriptor");
  for (int s = 0; s < ns; ++s) {      // a slice's largest possible vector must fit
    const lsm2d_cloudset* m = b->moving[s];
    if (!m || !b->fixed[s]) return fail(ctx, LSM2D_BAD_ARGUMENT, "align_batch_pairs: cloud set missing");
    long long need = 0;
    if (b->slices[s].finder == LSM2D_FINDER_PROJECTIVE) need = b->slices[s].projector.canvas_cols;
    else { const int rc0 = resolve_count(m); if (rc0) return rc0; for (int c = 0; c < m->n_clouds; ++c) if (m->h_count[c] > need) need = m->h_count[c]; }
    if (need > pair_capacity) return fail(ctx, LSM2D_CAPACITY_EXCEEDED, "align_batch_pairs: pair_capacity below a slice's largest possible correspondence vector");
  }
  std::vector<float> last_pose((size_t) 3 * (size_t) (n > 0 ? n : 1));
  std::vector<int32_t> its_local;
  int32_t* its = out_its;
  if (!its) { its_local.assign((size_t) (n > 0 ? n : 1), 0); its = its_local.data(); }
  int rc = align_batch_impl(ctx, ap, b, out_pose, out_H, out_status, its, out_stats, last_pose.data());
  if (rc) return rc;
  for (int i = 0; i < n; ++i)
    for (int s = 0; s < ns; ++s) {
      const lsm2d_slice_params& sp = b->slices[s];
      const lsm2d_cloudset* f = b->fixed[s]; const lsm2d_cloudset* m = b->moving[s];
      int32_t* cnt = out_n_pairs + (size_t) i * ns + s;
      lsm2d_correspondence* dst = out_pairs + ((size_t) i * ns + s) * (size_t) pair_capacity;
      *cnt = 0;
      if (its[i] < 1) continue;      // no iteration started: the vectors stay empty
      const int fc = b->fixed_index ? b->fixed_index[(size_t) s * n + i] : (f->n_clouds == 1 ? 0 : i);
      const int mc = b->moving_index ? b->moving_index[(size_t) s * n + i] : (m->n_clouds == 1 ? 0 : i);
      float Xe[3] = {last_pose[3 * (size_t) i], last_pose[3 * (size_t) i + 1], last_pose[3 * (size_t) i + 2]};
      const bool has_sensor = !(sp.sensor_in_robot[0] == 0.0f && sp.sensor_in_robot[1] == 0.0f && sp.sensor_in_robot[2] == 0.0f);
      if (has_sensor) { float Sinv[3], X[3] = {Xe[0], Xe[1], Xe[2]}; inverse_host(sp.sensor_in_robot, Sinv); compose_host(Sinv, X, Xe); }      // X_eff = S^-1 X, the kernels' operations
      const float inl_tau = (ap->keep_only_inlier_correspondences && sp.robustifier == LSM2D_ROBUST_CAUCHY) ? sp.chi_threshold : 0.0f;
      rc = find_correspondences_impl(ctx, &sp, f, fc, m, mc, Xe, dst, pair_capacity, cnt, inl_tau);
      if (rc) return rc;
    }
  return LSM2D_SUCCESS;
}

// ---- loop-closure / relocalisation sweep over several devices in ONE process (no Python, no MPI) ------------------------------
// What MultiLoopDetectorBruteForce2D's candidate loop (MULTI.json:964-986) becomes on a node of MI355Xs: every device gets its own
// context, a replica of the submap (device-to-device copies from the first device) and of the distinct candidate scans; the
// candidates are block-sharded, one host thread drives each device, results land in candidate order.  No collective on the data
// path; several entries of device_ids may name the same device (rehearsal on a one-GPU box).
struct lsm2d_sweep {
  std::vector<lsm2d_context*> ctx;
  std::vector<lsm2d_cloudset*> map, scans;
  std::string last_error;
  int peer_copy = 0;                                                  // 0 automatic, 1 replicas always filled from the host buffer
  long long by_peer = 0, through_host = 0, same_device = 0;           // how the replicas of the last set_map / set_scans were filled
};
static int sweep_fail(lsm2d_sweep* sw, int code, const std::string& msg) { if (sw) sw->last_error = msg; g_last_error = msg; return code; }
static void sweep_drop(std::vector<lsm2d_cloudset*>& v) { for (auto* s : v) lsm2d_cloudset_destroy(s); v.clear(); }

extern "C" int lsm2d_sweep_create(const int32_t* device_ids, int32_t n_devices, lsm2d_sweep** out) {
  if (!device_ids || n_devices < 1 || n_devices > 64 || !out) return fail(nullptr, LSM2D_BAD_ARGUMENT, "sweep_create: bad argument");
  lsm2d_sweep* sw = new (std::nothrow) lsm2d_sweep;
  if (!sw) return fail(nullptr, LSM2D_OUT_OF_MEMORY, "sweep_create: out of memory");
  for (int r = 0; r < n_devices; ++r) {
    lsm2d_context* c = nullptr;
    const int rc = lsm2d_create(device_ids[r], nullptr, &c);
    if (rc) { for (auto* k : sw->ctx) lsm2d_destroy(k); delete sw; return rc; }
    sw->ctx.push_back(c);
  }
  *out = sw;
  return LSM2D_SUCCESS;
}
extern "C" void lsm2d_sweep_destroy(lsm2d_sweep* sw) {
  if (!sw) return;
  sweep_drop(sw->map); sweep_drop(sw->scans);
  for (auto* c : sw->ctx) lsm2d_destroy(c);
  delete sw;
}
extern "C" int32_t lsm2d_sweep_num_devices(const lsm2d_sweep* sw) { return sw ? (int32_t) sw->ctx.size() : 0; }
extern "C" int lsm2d_sweep_set_option(lsm2d_sweep* sw, const char* key, int64_t value) {
  if (!sw || !key) return fail(nullptr, LSM2D_BAD_ARGUMENT, "sweep_set_option: bad argument");
  if (!strcmp(key, "peer_copy")) { if (value < 0 || value > 1) return sweep_fail(sw, LSM2D_BAD_ARGUMENT, "peer_copy must be 0 or 1"); sw->peer_copy = (int) value; return LSM2D_SUCCESS; }
  return sweep_fail(sw, LSM2D_BAD_ARGUMENT, "sweep_set_option: unknown option");
}
extern "C" int lsm2d_sweep_get_option(const lsm2d_sweep* sw, const char* key, int64_t* out) {
  if (!sw || !key || !out) return fail(nullptr, LSM2D_BAD_ARGUMENT, "sweep_get_option: bad argument");
  if (!strcmp(key, "peer_copy")) { *out = sw->peer_copy; return LSM2D_SUCCESS; }
  if (!strcmp(key, "replicas_by_peer_copy")) { *out = sw->by_peer; return LSM2D_SUCCESS; }
  if (!strcmp(key, "replicas_through_host")) { *out = sw->through_host; return LSM2D_SUCCESS; }
  if (!strcmp(key, "replicas_same_device")) { *out = sw->same_device; return LSM2D_SUCCESS; }
  return fail(nullptr, LSM2D_BAD_ARGUMENT, "sweep_get_option: unknown option");
}
extern "C" const char* lsm2d_sweep_last_error(const lsm2d_sweep* sw) { return sw ? sw->last_error.c_str() : g_last_error.c_str(); }

// one host cloud set replicated on every device of the sweep: host -> first device once, then device -> device
static int sweep_replicate(lsm2d_sweep* sw, const float* pts, const int32_t* offsets, int32_t n_clouds, int64_t total, std::vector<lsm2d_cloudset*>* out) {
  sweep_drop(*out);
  if (!pts || total < 0 || n_clouds < 1) return sweep_fail(sw, LSM2D_BAD_ARGUMENT, "sweep: bad cloud");
  const size_t bytes = sizeof(float) * 4 * (size_t) (total > 0 ? total : 1);
  lsm2d_context* c0 = sw->ctx[0];
  void* d0 = nullptr;
  if (hipSetDevice(c0->device) != hipSuccess || hipMalloc(&d0, bytes) != hipSuccess) return sweep_fail(sw, LSM2D_OUT_OF_MEMORY, "sweep: staging allocation failed");
  if (total > 0 && hipMemcpy(d0, pts, sizeof(float) * 4 * (size_t) total, hipMemcpyHostToDevice) != hipSuccess) { (void) hipFree(d0); return sweep_fail(sw, LSM2D_DEVICE_ERROR, "sweep: upload failed"); }
  int rc = LSM2D_SUCCESS;
  sw->by_peer = sw->through_host = sw->same_device = 0;
  for (size_t r = 0; r < sw->ctx.size() && rc == LSM2D_SUCCESS; ++r) {
    lsm2d_context* c = sw->ctx[r];
    void* dr = d0;
    if (r > 0) {      // a replica of its own: over the fabric (xGMI between the GPUs of a node) where the two devices reach each other, else from the host buffer
      if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&dr, bytes) != hipSuccess) { rc = sweep_fail(sw, LSM2D_OUT_OF_MEMORY, "sweep: replica allocation failed"); break; }
      const size_t nbytes = sizeof(float) * 4 * (size_t) total;
      bool done = total == 0;
      if (!done && sw->peer_copy == 0) {
        if (c->device == c0->device) {                        // a rehearsal of several shards on one device: a plain device-to-device copy
          done = hipMemcpy(dr, d0, nbytes, hipMemcpyDeviceToDevice) == hipSuccess;
          if (done) ++sw->same_device;
        } else {
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, c->device, c0->device) == hipSuccess && can) {
            const hipError_t pe = hipDeviceEnablePeerAccess(c0->device, 0);      // current device: c->device; once per pair
            if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) done = hipMemcpyPeer(dr, c->device, d0, c0->device, nbytes) == hipSuccess;
            if (done) ++sw->by_peer;
          }
        }
        (void) hipGetLastError();                             // a refused peer path is not an error: the host path follows
      }
      if (!done) {                                            // no peer access (or switched off, or the peer copy failed): the caller's host buffer is the source
        if (hipSetDevice(c->device) != hipSuccess || hipMemcpy(dr, pts, nbytes, hipMemcpyHostToDevice) != hipSuccess) { (void) hipFree(dr); rc = sweep_fail(sw, LSM2D_DEVICE_ERROR, "sweep: replica upload failed"); break; }
        ++sw->through_host;
      }
    }
    lsm2d_cloudset* set = nullptr;
    rc = lsm2d_cloudset_create_from_device(c, dr, offsets, n_clouds, total, &set);
    if (rc == LSM2D_SUCCESS) { rc = lsm2d_synchronize(c); out->push_back(set); }      // the split of the AoS staging buffer has run before it is freed
    if (r > 0) { (void) hipSetDevice(c->device); (void) hipFree(dr); }
  }
  (void) hipSetDevice(c0->device); (void) hipFree(d0);
  if (rc != LSM2D_SUCCESS) { sweep_drop(*out); return rc; }
  return LSM2D_SUCCESS;
}
extern "C" int lsm2d_sweep_set_map(lsm2d_sweep* sw, const float* map_xynn, int64_t n_points) {
  if (!sw) return fail(nullptr, LSM2D_BAD_ARGUMENT, "sweep_set_map: null sweep");
  return sweep_replicate(sw, map_xynn, nullptr, 1, n_points, &sw->map);
}
extern "C" int lsm2d_sweep_set_scans(lsm2d_sweep* sw, const float* scans_xynn, const int32_t* offsets, int32_t n_scans) {
  if (!sw || !offsets || n_scans < 1) return fail(nullptr, LSM2D_BAD_ARGUMENT, "sweep_set_scans: bad argument");
  return sweep_replicate(sw, scans_xynn, offsets, n_scans, offsets[n_scans], &sw->scans);
}

extern "C" int lsm2d_sweep_align(lsm2d_sweep* sw, const lsm2d_aligner_params* ap, const lsm2d_slice_params* slice, int32_t n_candidates,
                                 const int32_t* scan_index, const float* init_pose, float* out_pose, float* out_H, int32_t* out_status,
                                 int32_t* out_iterations, lsm2d_iteration_stats* out_last_stats) {
  if (!sw || !ap || !slice || n_candidates < 0 || (n_candidates > 0 && (!init_pose || !out_pose || !out_status)))
    return fail(nullptr, LSM2D_BAD_ARGUMENT, "sweep_align: bad argument");
  if (sw->map.size() != sw->ctx.size() || sw->scans.size() != sw->ctx.size()) return sweep_fail(sw, LSM2D_BAD_ARGUMENT, "sweep_align: set_map / set_scans first");
  if (n_candidates == 0) return LSM2D_SUCCESS;      // an empty candidate list is a no-op, as an empty batch is for lsm2d_align_batch
  const int G = (int) sw->ctx.size();
  const int n_scans = lsm2d_cloudset_num_clouds(sw->scans[0]);
  if (!scan_index && n_scans != n_candidates && n_scans != 1) return sweep_fail(sw, LSM2D_BAD_ARGUMENT, "sweep_align: scan_index needed unless there is one scan per candidate");
  // no exception crosses the C ABI: allocation failures inside a worker become that device's status, a failure to start a thread joins
  // the ones already running and returns LSM2D_OUT_OF_MEMORY
  std::vector<int> rcs; std::vector<std::thread> workers;
  try { rcs.assign((size_t) G, LSM2D_SUCCESS); workers.reserve((size_t) G); } catch (...) { return sweep_fail(sw, LSM2D_OUT_OF_MEMORY, "sweep_align: out of memory"); }
  bool spawn_failed = false;
  for (int r = 0; r < G && !spawn_failed; ++r) {
    const long long lo = (long long) n_candidates * r / G, hi = (long long) n_candidates * (r + 1) / G;
    if (hi <= lo) continue;
    try {
    workers.emplace_back([=, &rcs]() {
      try {
      const int n = (int) (hi - lo);
      // without an index array candidate i uses scan i: the shard needs an explicit index then (its scans start at lo)
      std::vector<int32_t> own_index;
      const int32_t* idx = scan_index ? scan_index + lo : nullptr;
      if (!idx && n_scans != 1) { own_index.resize((size_t) n); for (int i = 0; i < n; ++i) own_index[(size_t) i] = (int32_t) (lo + i); idx = own_index.data(); }
      const lsm2d_cloudset* fx = sw->scans[(size_t) r]; const lsm2d_cloudset* mv = sw->map[(size_t) r];
      lsm2d_batch b; memset(&b, 0, sizeof b);
      b.n_alignments = n; b.n_slices = 1; b.slices = slice; b.fixed = &fx; b.moving = &mv; b.fixed_index = idx; b.init_pose = init_pose + 3 * lo;
      std::vector<lsm2d_iteration_stats> stats;
      if (out_last_stats) stats.resize((size_t) n * (size_t) lsm2d_stats_capacity(ap));
      std::vector<int32_t> its((size_t) n);
      const int rc = lsm2d_align_batch(sw->ctx[(size_t) r], ap, &b, out_pose + 3 * lo, out_H ? out_H + 9 * lo : nullptr, out_status + lo, its.data(),
                                       out_last_stats ? stats.data() : nullptr);
      rcs[(size_t) r] = rc;
      if (rc != LSM2D_SUCCESS) return;
      if (out_iterations) memcpy(out_iterations + lo, its.data(), sizeof(int32_t) * (size_t) n);
      if (out_last_stats)
        for (int i = 0; i < n; ++i) {
          lsm2d_iteration_stats z; memset(&z, 0, sizeof z);
          out_last_stats[lo + i] = its[(size_t) i] > 0 ? stats[(size_t) i * (size_t) lsm2d_stats_capacity(ap) + (size_t) (its[(size_t) i] - 1)] : z;
        }
      } catch (const std::bad_alloc&) { rcs[(size_t) r] = LSM2D_OUT_OF_MEMORY; } catch (...) { rcs[(size_t) r] = LSM2D_DEVICE_ERROR; }
    });
    } catch (...) { spawn_failed = true; }      // std::system_error from the thread constructor, bad_alloc from the vector
  }
  for (auto& w : workers) w.join();
  if (spawn_failed) return sweep_fail(sw, LSM2D_OUT_OF_MEMORY, "sweep_align: could not start a worker thread");
  for (int r = 0; r < G; ++r) if (rcs[(size_t) r] != LSM2D_SUCCESS) return sweep_fail(sw, rcs[(size_t) r], std::string("sweep_align: device ") + std::to_string(r) + ": " + lsm2d_last_error(sw->ctx[(size_t) r]));
  return LSM2D_SUCCESS;
}
