// lsm2d_k_align_args.h -- what an aligner launch is handed (SliceDev, AlignArgs) and the small pieces every aligner kernel shares: the bin gates, X_eff = S^-1 X, the odometry prior (MultiAligner2D, registration/aligner_slice_processor_laser_2d.h:7-42).
// Part of lsm2d_kernels.h (included there, inside namespace lsm2d, in this order); not a translation unit of its own.
struct SliceDev {
  CloudDev fixed, moving;
  int32_t finder;
  ProjK   proj;
  float   point_distance, normal_cos, max_distance;
  int32_t nn_group;          // host hint (largest fixed vs largest moving cloud of the sets): 1 = scan-sized fixed clouds, worth staging their tables in LDS
  int32_t cauchy;
  float   tau;
  int32_t min_corr;
  int32_t has_sensor;        // X_eff = S^-1 * X
  float   Sinv[3], cSinv, sSinv;
  int32_t fcan_offset;       // start of this slice's fixed canvas, in cells
  // single-alignment calls whose fixed set still sits in its pinned upload buffer (lsm2d_cloudset_upload defers the unpacking):
  // the kernel's prologue reads the host's AoS points over the bus itself, writes the set's arrays and count (later consumers
  // find them there) and goes on -- no separate k_upload_unpack launch in front of the alignment
  const float4* unpack_src;  // device view of the pinned AoS points, or nullptr
  int32_t unpack_n;
};

struct PriorDev { float z_inv[3], cz, sz, omega[9]; };   // Z^-1 and cos/sin of its angle, host-computed
struct StatsDev { int32_t n_corr, n_in, n_out; float chi_in, chi_out; uint32_t dig_lo, dig_hi; };

// lsm2d_pair_hash (include/lsm2d.h) on the device: the per-pair term of lsm2d_iteration_stats.pair_digest.  slice_salt = slice * 0x632BE5AB
// (wave-uniform).  Integer arithmetic only; tests hold the two definitions against each other through the oracle's digest.
LSM2D_DEV u64 pair_hash_dev(uint32_t slice_salt, uint32_t f, uint32_t m) {
  const uint32_t a = f * 0x9E3779B1u, b = (m ^ slice_salt) * 0x85EBCA77u;
  uint32_t lo = a ^ __builtin_rotateleft32(b, 13), hi = b ^ __builtin_rotateleft32(a, 19);
  lo += __builtin_rotateleft32(lo, 17) ^ b;
  hi += __builtin_rotateleft32(hi, 11) ^ a;
  return ((u64) hi << 32) | (u64) lo;
}
// one pair into the iteration's digest: a fire-and-forget 64-bit LDS add (order-independent: the sum wraps mod 2^64), no register held across the loops
LSM2D_DEV void digest_add(u64* s_dig, uint32_t slice_salt, int f, int m) { atomicAdd(reinterpret_cast<unsigned long long*>(s_dig), (unsigned long long) pair_hash_dev(slice_salt, (uint32_t) f, (uint32_t) m)); }

struct ResumeDev {      // what an alignment carries from one iteration to the next (thread 0's serial state in k_align)
  float pose[3]; float H[9]; float prev_chi;
  int32_t phase, phase_start, phase_end, last_n_in, status, done, it;
};
struct AlignArgs {
  int32_t n_align, n_slices, max_it, min_inliers;
  float   damping;
  float   term_eps;                         // lsm2d_aligner_params.termination_chi_epsilon (0 = run all iterations)
  int32_t inlier_runs;                      // lsm2d_aligner_params.enable_inlier_only_runs: a second loop of up to max_it iterations over inliers only (lsm2d.h)
  int32_t stats_stride;                     // iterations an alignment may run = row length of out_stats: max_it * (1 + inlier_runs), at least 1
  float*  out_last_pose;                    // [n][3] or nullptr: the pose the LAST started iteration began at (lsm2d_align_batch_pairs re-derives that iteration's pairs from it)
  int32_t cols_max, fcan_total;
  int32_t nn_lds_points, nn_lds_cells;      // > 0: single NN slice over scan-sized fixed clouds -- their search tables are staged in LDS (room for this many)
  int32_t nn_qcache;                        // > 0: single NN slice with its tables in global memory (kNNGlobal): room in LDS for this many queries' cached cell ranges (32 bytes each)
  int32_t kd_lds_nodes;                     // > 0: single KD-tree slice -- room in LDS for this many nodes of the fixed cloud's tree (its top levels)
  int32_t kd_lds_points;                    // > 0: ... and, for scan-sized fixed clouds, for this many leaf points (whole trees on chip)
  const int32_t* order;                     // alignment handled by workgroup b (nullptr: b itself) -- the balanced placement of k_balance_order
  float cull_est_mt, cull_est_mth;          // margins of k_cull_estimate's chunk test (metres, radians)
  // two launches for one batch (k_first_iteration, then k_align: see k_first_iteration): stage 0 the whole alignment in this launch; 1 the iterations before
  // stage_split, then the next iteration's unit lists for their LENGTH only, the state to `resume`, the length to `stage_work`; 2 the rest, from `resume`
  int32_t stage, stage_split;
  struct ResumeDev* resume;                 // [n]
  int32_t* stage_work;                      // [n] 0 .. 512: the units the alignment will stream per iteration (0: it finished in the first stage)
  float cull_mt2;                           // cull_mt squared (host)
  int32_t* wg_place;                        // [grid] or nullptr: every workgroup notes the CU it ran on (place_key) for the next call's placement
  int32_t cull_block;                       // steps per unit of the culled stream (0: automatic; tuning knob)
  int32_t cull;                             // 1: projective slices drop the chunks of the moving cloud that cannot yield a pair (chunk_may_matter), results unchanged
  // kProjCulled (round 4): every slice keeps a LIST of the (block, chunk) units that survive the test at block level, in dynamic LDS at units_off
  // (kCullBlocks * kAlignBlock 16-bit entries per slice), built with margins (cull_mt metres, cull_mth radians) and kept while the slice's transform
  // stays within them of the one it was built at (cull_keep 0: rebuilt every iteration, zero margins -- A/B knob)
  int32_t units_off, cull_keep;
  int32_t units_stride;                     // entries per slice of the unit lists: the largest block_stride of the batch's moving sets x kAlignBlock
  float   cull_mt, cull_mth;
  // Round 5, big maps (k_align<1,0,0,0,6>): the workgroups of one XCD walk the map IN STEP, pass by pass.  A 1M-point map's lane copy (8 MB) does not fit an
  // XCD's 4 MiB L2, and 125 workgroups streaming different parts of it at the same time missed on 44 % of their requests (41 GB of fabric reads per
  // 1000-alignment launch for 50 MB of data, the chip at 1.83 GHz under that load: profiles/r05/size_sweep_r05a.txt).  All workgroups of a one-round launch start
  // together and walk their unit lists in the same block-major order -- what pulls them apart is only that their lists differ in length, a quarter of a pass per
  // iteration.  So every workgroup counts itself into done[g] when ITS pass g = (iteration, slice) is over, and its thread 0 -- at the end of the serial solve,
  // while the other threads stand at the iteration's closing barrier anyway -- waits until every workgroup registered on ITS XCD has finished the pass that
  // lies xcd_window passes back (0: the one just finished) or has gone.  One atomic add and a handful of scalar looks per workgroup and iteration; the counters
  // of an XCD are touched by that XCD only (plain L2 atomics, no fabric traffic, no fence).  Nothing but the ORDER IN TIME of the z-buffer updates changes:
  // every result keeps its bits.  xcd_sync == nullptr: free-running (the host offers the lockstep only to launches of one dispatch round, whose workgroups
  // are all resident from the start).  What was tried before this form -- per wave and per BLOCK of the map -- and what it cost: DESIGN App. A.
  uint32_t* xcd_sync;                       // [16 XCC ids][xcd_stride]: word 0 workgroups registered, word 1 workgroups gone, words 2..6 the watchdog's notes, word 16 + g: workgroups that have finished pass g
  int32_t xcd_stride, xcd_window, xcd_positions;
  int32_t pq_cull_off;                      // > 0: byte offset in dynamic LDS of the point-query finders' culling state (occupancy bitmap of the fixed cloud, then
  int32_t pq_keep_words;                    //   pq_keep_words 64-bit words of per-tile keep bits); single-slice NN / KD-tree batches with scan-sized fixed clouds
  int32_t pair_mov_cap;                     // latency kernel: moving points per slice it may keep in LDS (kPairMovCap, or 0: no room)
  int32_t pair_fix_cap;                     // latency kernel: fixed points per slice it may keep in LDS (0: no room)
  const float* init_pose;
  const PriorDev* prior;
  int32_t  host_polls;                      // results go to pinned host memory and the host polls the status words: release them to the system
  int32_t  inline_n1;                       // 1: a single alignment whose start pose / prior travel in the kernel arguments (pose1, prior1)
  float    pose1[3];
  PriorDev prior1;
  float* out_pose; float* out_H; int32_t* out_status; int32_t* out_its; StatsDev* out_stats;
  // kernel timing on (lsm2d_set_option "kernel_timing"): thread 0 of every clock_stride-th workgroup stamps s_memtime (shader
  // cycles) and s_memrealtime (100 MHz) once at its start and once at its end -- [a / clock_stride][4] = {cycles, 10 ns ticks, start tick, hardware id} of
  // the workgroup's lifetime, from which the host reads the clock the chip held under THIS load (MI355X_MICROARCH.md, DVFS note 6)
  unsigned long long* clock_out; int32_t clock_stride;
  // "sum_order" 1 (the k_align_seq / k_split_finish<true> instantiations): byte offset in dynamic LDS of the trip's pair records, kAlignBlock x kSeqFields floats
  // (lsm2d_device.h, "sum_order"); sits in what was padding, so the other fields keep their offsets
  int32_t seq_off;
  const int32_t* order2;                    // (round 6, k_align_two) a SECOND alignment for workgroup b to run after its first (-1: none), or nullptr: a batch of a few alignments more than
                                            // the chip holds packs its lightest ones two to a workgroup instead of starting another dispatch round (balance_order, "packed")
  SliceDev s[kMaxSlices];
};

LSM2D_DEV int pick_cloud(const CloudDev& c, int a) {
  return c.index ? c.index[a] : (c.n_clouds == 1 ? 0 : a);
}

// bin walk of one column: gates of correspondence_finder_projective_2d.cpp:61-69
LSM2D_DEV bool match_bin(u64 fk, u64 mk, const SliceDev& S, const Iso& T, const float2* fn, const float2* mn,
                         int& fi, int& mi, float2& nf, float2& nm) {
  if (mk == kEmptyCell || fk == kEmptyCell) return false;
  const float fd = __uint_as_float((uint32_t) (fk >> 32)), md = __uint_as_float((uint32_t) (mk >> 32));
  if (__builtin_fabsf(fd - md) > S.point_distance) return false;
  fi = (int) (uint32_t) fk; mi = (int) (uint32_t) mk;
  nf = fn[fi]; nm = mn[mi];
  float nqx, nqy;
  xf_normal(T, nm.x, nm.y, nqx, nqy);
  const float dot = __builtin_fmaf(nqx, nf.x, nqy * nf.y);
  return !(dot < S.normal_cos);
}

LSM2D_DEV Iso slice_iso_of(int has_sensor, float cSinv, float sSinv, const float Sinv[3], const float pose[3]) {
  float Xe[3] = {pose[0], pose[1], pose[2]};
  if (has_sensor) compose(cSinv, sSinv, Sinv, pose, Xe);
  Iso T; sincos_fixed(Xe[2], T.s, T.c); T.tx = Xe[0]; T.ty = Xe[1];
  return T;
}
LSM2D_DEV Iso slice_iso(const SliceDev& S, const float pose[3]) {      // X_eff = S^-1 X (AlignerSliceProcessorLaser2DWithSensor) as rotation + translation
  return slice_iso_of(S.has_sensor, S.cSinv, S.sSinv, S.Sinv, pose);
}
// prologue of the single-alignment kernels: unpack the slices' freshly uploaded fixed sets (see SliceDev::unpack_src)
LSM2D_DEV void unpack_fixed_set(const SliceDev& S, int tid, int nthreads) {
  float2* xy = const_cast<float2*>(S.fixed.xy); float2* nrm = const_cast<float2*>(S.fixed.nrm);
  for (int i = tid; i < S.unpack_n; i += nthreads) {
    const float4 v = S.unpack_src[i];
    xy[i] = make_float2(v.x, v.y); nrm[i] = make_float2(v.z, v.w);
  }
  if (tid == 0) *const_cast<int32_t*>(S.fixed.count) = S.unpack_n;
}
#ifndef LSM2D_ALIGN_MIN_WAVES
#define LSM2D_ALIGN_MIN_WAVES 8      // waves per SIMD the register allocator must leave room for
#endif
#ifndef LSM2D_QUERY_MIN_WAVES
#define LSM2D_QUERY_MIN_WAVES 8      // the same for the instantiations without a projective slice (point-query finders); 6 and 4 measured 15-50 % slower
#endif
// SE2 odometry prior (AlignerSliceOdom2DPrior, MULTI.json:402-422): e = t2v(Z^-1 X), J = blkdiag(R_e, 1) for the right perturbation;
// adds J^T Omega J to H and J^T Omega e to b.  One definition for k_align and the split path: the same operation order in both.
// prior_terms: the nine and three values that go into H and b -- they do not depend on H or b, so whoever has the pose can have them ready
// (the latency kernel's thread 0 computes them while it would otherwise wait at the barrier for the slowest wave)
// (kAdd: the terms are added to H and b as they come -- the form k_align's out-of-line call takes: twelve registers fewer)
template <bool kAdd>
LSM2D_DEV void prior_apply(const PriorDev& Pz, const float pose[3], float Hp[9], float bp[3]) {
  float E[3]; compose(Pz.cz, Pz.sz, Pz.z_inv, pose, E);
  float c, s_; sincos_fixed(E[2], s_, c);
  const float Jp[9] = {c, -s_, 0.0f, s_, c, 0.0f, 0.0f, 0.0f, 1.0f};
  float OJ[9], Oe[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    Oe[r] = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) Oe[r] += Pz.omega[3 * r + k] * E[k];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      OJ[3 * r + cc] = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) OJ[3 * r + cc] += Pz.omega[3 * r + k] * Jp[3 * k + cc];
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      float v = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) v += Jp[3 * k + r] * OJ[3 * k + cc];
      if (kAdd) Hp[3 * r + cc] += v; else Hp[3 * r + cc] = v;
    }
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) v += Jp[3 * k + r] * Oe[k];
    if (kAdd) bp[r] += v; else bp[r] = v;
  }
}
LSM2D_DEV void prior_terms(const PriorDev& Pz, const float pose[3], float Hp[9], float bp[3]) { prior_apply<false>(Pz, pose, Hp, bp); }
LSM2D_DEV void add_prior_inline(const PriorDev& Pz, const float pose[3], float H[9], float b[3]) { prior_apply<true>(Pz, pose, H, b); }
// k_align (64 VGPRs) and the split path call it: rarely taken, and out of the register allocation of their loops
__device__ __noinline__ void add_prior(const PriorDev& Pz, const float pose[3], float H[9], float b[3]) { add_prior_inline(Pz, pose, H, b); }

LSM2D_DEV int block_compact_pos(bool flag, int* s_tot, int parity, int& base, int tid, int nwaves);      // (defined with the mapping kernels below)

// kHasProj / kHasNN: which finders the batch's slices use -- the unused one is compiled out so the
// projective hot loop does not carry the NN path's register pressure (and vice versa).
// kHasDist: a slice uses the distance-map finder (compiled out otherwise so the NN search keeps its registers).
// kHasKd: a slice uses the KD-tree finder (LSM2D_FINDER_KDTREE); compiled out otherwise.
// kNNGlobal: a pure grid-NN batch whose search tables stay in global memory (the map is the fixed cloud: BASELINE's wording with the exact search) --
// an instantiation of its own, so that its position-keeping search (nn_query_pos) does not share 64 registers with the LDS-table path of the
// tracker's wiring (both forms in one kernel: scratch 16 -> 80 bytes, role A 7.2 -> 10.6 ms)
// kNNMode 2: the counterpart -- a pure grid-NN batch whose tables the host has PROVED to fit the LDS staging for every alignment (nn_lds_points is the
// largest fixed cloud, nn_lds_cells the grid ensure_grid() gives that size): the search in global memory and the cooperative loop are compiled out
// where a workgroup runs: the key the balanced placement groups workgroup ids by (see k_balance_order)
static constexpr int kPlaceKeys = 4096;      // XCC (4 bits) | SE (3) | SH (1) | CU (4)
LSM2D_DEV int place_key() {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);      // HW_REG_HW_ID, HW_REG_XCC_ID
  return (int) (((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u));
}
