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
  int pack_extra_max = 32;     // packed batches above 2048 alignments: up to this many more than whole rounds hold (experiments build: tuning knob)
  int align_width = 0;         // threads per workgroup of a culled projective batch: 0 automatic (align_width_for), 512 / 256 forced; results never depend on it
  int last_align_width = 0;    // what the latest k_align launch used
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
  int order_cluster = 0;       // experiments build: > 0 = big-map batches are dealt to workgroup ids by the sensors' positions and headings (value / 10 = metres per radian of heading): see make_placement
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
  (void) hipFuncSetAttribute((const void*) k_align_narrow<256>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds);
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
  {"align_width",        &lsm2d_context::align_width,        0, 1024,       0},
  // ---- read-only
  {"last_align_path",    &lsm2d_context::last_align_path,    0, 0, kOptReadOnly},
  {"last_align_width",   &lsm2d_context::last_align_width,   0, 0, kOptReadOnly},
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
  {"order_cluster",      &lsm2d_context::order_cluster,      0, 10000,   kOptExperiment},
  {"pack_extra_max",     &lsm2d_context::pack_extra_max,     0, 1024,    kOptExperiment},
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

// ---- the rest of the host side, by subject (one translation unit: they share the context, the cloud sets and the helpers above) ----
#include "lsm2d_capi_cloudsets.inc"
#include "lsm2d_capi_structures.inc"
#include "lsm2d_capi_mapping.inc"
#include "lsm2d_capi_finder.inc"
#include "lsm2d_capi_aligner.inc"
#include "lsm2d_capi_sweep.inc"
