// lsm2d_k_split_finder.h -- the split aligner path (k_split_project / k_split_finish), the finder-level kernels (CorrespondenceFinder_::compute), the projector and the factor over a correspondence vector.
// Part of lsm2d_kernels.h (included there, inside namespace lsm2d, in this order); not a translation unit of its own.
// ---- split path: the same alignment spread over many workgroups -------------------------------------------------
// For a handful of alignments against a big cloud one workgroup per alignment leaves the chip empty, so each iteration
// becomes two launches: k_split_project z-buffers slices of the cloud in LDS and folds them into a global canvas
// (atomicMin u64 is order independent), k_split_finish does the bin walk, the reduction (same thread <-> column mapping,
// same order as k_align, hence bit-identical sums), the 3x3 solve and the pose update.  Projective slices only.
struct SplitArgs {
  AlignArgs A;
  u64* gcan;             // [n_align][2 * fcan_total]: fixed canvases then moving canvases, pre-filled with kEmptyCell
  float* pose;           // [n_align][3] current estimate
  int32_t* done;         // [n_align] 0 = running
  float* H_last;         // [n_align][9]
  StatsDev* last;        // [n_align]
  int32_t* phase;        // [n_align][3]: phase (0 regular, 1 inlier-only runs), its first iteration, its end -- zero-filled means (0, 0, max_it)
  int32_t it;            // iteration this launch belongs to
};


template <bool kFixed>
__global__ __launch_bounds__(512) void k_split_project(const SplitArgs S) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* can = reinterpret_cast<u64*>(smem);
  __shared__ Iso s_T;
  const int a = blockIdx.y, sl = blockIdx.z, tid = threadIdx.x;
  if (S.done[a]) return;
  const SliceDev& SL = S.A.s[sl];
  const CloudDev& C = kFixed ? SL.fixed : SL.moving;
  const int ci = pick_cloud(C, a), n = C.count[ci];
  const int npairs = (n + 1) >> 1;
  const int per = (npairs + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = lo + per < npairs ? lo + per : npairs;
  if (lo >= hi) return;
  if (tid == 0) {
    if (kFixed) { s_T.c = 1.0f; s_T.s = 0.0f; s_T.tx = 0.0f; s_T.ty = 0.0f; }
    else { const float p[3] = {S.pose[3 * a], S.pose[3 * a + 1], S.pose[3 * a + 2]}; s_T = slice_iso(SL, p); }
  }
  const ProjK P = SL.proj;
  for (int i = tid; i < P.cols; i += 512) can[i] = kEmptyCell;
  __syncthreads();
  const Iso T = s_T;
  const float4* xy4 = reinterpret_cast<const float4*>(C.xy + C.start[ci]);
  for (int j = lo + tid; j < hi; j += 512) {
    const float4 v = xy4[j];
    project_point(T, P, v.x, v.y, 2 * j, can);
    if (2 * j + 1 < n) project_point(T, P, v.z, v.w, 2 * j + 1, can);
  }
  __syncthreads();
  u64* g = S.gcan + (size_t) a * 2 * S.A.fcan_total + (kFixed ? 0 : S.A.fcan_total) + SL.fcan_offset;
  for (int i = tid; i < P.cols; i += 512) { const u64 k = can[i]; if (k != kEmptyCell) atomicMin(&g[i], k); }
}

// kSeq: "sum_order" 1 -- the sums pair after pair in ascending column (lsm2d_device.h: pair_terms / seq_walk), as k_align_seq forms them
template <bool kSeq>
__global__ __launch_bounds__(kAlignBlock) void k_split_finish(const SplitArgs S) {
  const AlignArgs& A = S.A;
  __shared__ float red[(kAlignBlock / 64) * kAccumWords];
  __shared__ __attribute__((aligned(16))) float s_rec[kSeq ? kSeqHalf * kSeqFields : 4];
  __shared__ Iso s_iso[kMaxSlices];
  // as in k_align: the iteration's sums are added in LDS by the lanes that gathered them, the matrix is assembled, given its prior and
  // solved where it lies (no private arrays, no scratch on the serial stretch)
  __shared__ float s_H[9], s_rhs[3], s_sum[kAccumWords + 2], s_pose[3];
  __shared__ int s_n_corr, s_active;
  __shared__ u64 s_dig;
  const int a = blockIdx.x, tid = threadIdx.x;
  constexpr int nwaves = kAlignBlock / 64;
  if (S.done[a]) return;
  const bool want_dig = A.out_stats != nullptr;
  const bool inl_only = A.inlier_runs && S.phase[3 * a] != 0;
  if (tid == 0) {
    s_dig = 0ull;
    s_pose[0] = S.pose[3 * a]; s_pose[1] = S.pose[3 * a + 1]; s_pose[2] = S.pose[3 * a + 2];
    if (A.out_last_pose) { A.out_last_pose[3 * a] = s_pose[0]; A.out_last_pose[3 * a + 1] = s_pose[1]; A.out_last_pose[3 * a + 2] = s_pose[2]; }
    for (int s = 0; s < A.n_slices; ++s) s_iso[s] = slice_iso(A.s[s], s_pose);
    for (int k = 0; k < 11; ++k) s_sum[k] = 0.0f;
    s_sum[11] = s_sum[12] = __int_as_float(0);
    s_n_corr = s_active = 0;
  }
  __syncthreads();
  u64* gF = S.gcan + (size_t) a * 2 * A.fcan_total; u64* gM = gF + A.fcan_total;
  for (int s = 0; s < A.n_slices; ++s) {
    const SliceDev& SL = A.s[s];
    const Iso T = s_iso[s];
    const int fc = pick_cloud(SL.fixed, a), mc = pick_cloud(SL.moving, a);
    const int mbase = SL.moving.start[mc], fbase = SL.fixed.start[fc];
    const float2* fn = SL.fixed.nrm + fbase; const float2* mn = SL.moving.nrm + mbase;
    const float2* fp = SL.fixed.xy + fbase;  const float2* mp = SL.moving.xy + mbase;
    Accum acc; accum_zero(acc);
    float seq_acc = 0.0f;
    if constexpr (kSeq) {
      for (int col0 = 0; col0 < SL.proj.cols; col0 += kAlignBlock) {      // trips of kAlignBlock consecutive columns, every thread in every trip (barriers)
        const int col = col0 + tid;
        float t[kSeqFields]; seq_zero(t);
        if (col < SL.proj.cols) {
          const u64 mk = gM[SL.fcan_offset + col];
          gM[SL.fcan_offset + col] = kEmptyCell;
          int fi, mi; float2 nf, nm;
          if (match_bin(gF[SL.fcan_offset + col], mk, SL, T, fn, mn, fi, mi, nf, nm)) {
            if (want_dig) digest_add(&s_dig, (uint32_t) s * 0x632BE5ABu, fi, mi);
            bool inl; pair_terms(T, fp[fi], nf, mp[mi], nm, SL.cauchy != 0, SL.tau, inl_only, t, inl);
            ++acc.n_corr; acc.n_in += inl ? 1 : 0; acc.n_out += inl ? 0 : 1;
          }
        }
        const int n_rec = SL.proj.cols - col0 < kAlignBlock ? SL.proj.cols - col0 : kAlignBlock;
        for (int h0 = 0; h0 < n_rec; h0 += kSeqHalf) {      // the trip's records in two halves
          if (tid >= h0 && tid < h0 + kSeqHalf) seq_store(s_rec, tid - h0, t);
          __syncthreads();
          const int left = n_rec - h0;
          if (tid < 64) seq_acc = seq_walk(s_rec, left < kSeqHalf ? left : kSeqHalf, tid, seq_acc);
          __syncthreads();
        }
      }
    } else
    for (int col = tid; col < SL.proj.cols; col += kAlignBlock) {
      const u64 mk = gM[SL.fcan_offset + col];
      gM[SL.fcan_offset + col] = kEmptyCell;                  // ready for the next iteration's projection
      int fi, mi; float2 nf, nm;
      if (match_bin(gF[SL.fcan_offset + col], mk, SL, T, fn, mn, fi, mi, nf, nm)) {
        if (want_dig) digest_add(&s_dig, (uint32_t) s * 0x632BE5ABu, fi, mi);
        accumulate_pair(T, fp[fi], nf, mp[mi], nm, SL.cauchy != 0, SL.tau, acc, inl_only);
      }
    }
    block_reduce_store(acc, red, tid);
    __syncthreads();
    if (tid < 64) {
      float v; int vi; block_reduce_gather_lane(red, nwaves, tid, v, vi);
      if constexpr (kSeq) v = seq_total(seq_acc, tid);
      const int n_corr = __builtin_amdgcn_readlane(vi, 13);
      if (tid == 0) s_n_corr += n_corr;
      if (n_corr > SL.min_corr) {
        if (tid < 11) s_sum[tid] += v;
        else if (tid < 13) s_sum[tid] = __int_as_float(__float_as_int(s_sum[tid]) + vi);
        if (tid == 0) ++s_active;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    StatsDev last; last.n_corr = s_n_corr; last.n_in = __float_as_int(s_sum[11]); last.n_out = __float_as_int(s_sum[12]); last.chi_in = s_sum[9]; last.chi_out = s_sum[10];
    if (A.out_stats) { const u64 dg = s_dig; last.dig_lo = (uint32_t) dg; last.dig_hi = (uint32_t) (dg >> 32); A.out_stats[(size_t) a * A.stats_stride + S.it] = last; }
    int status = LSM2D_RUNNING;
    bool stop_now = false;
    const int ph = A.inlier_runs ? S.phase[3 * a] : 0, ph_start = ph ? S.phase[3 * a + 1] : 0, ph_end = ph ? S.phase[3 * a + 2] : A.max_it;
    if (!s_active) {
      status = LSM2D_NOT_ENOUGH_CORRESPONDENCES;
      for (int k = 0; k < 9; ++k) s_H[k] = S.it == 0 ? 0.0f : S.H_last[9 * a + k];      // the information matrix stays the last solved iteration's
    } else {
      s_H[0] = s_sum[0]; s_H[1] = s_sum[1]; s_H[2] = s_sum[2]; s_H[3] = s_sum[1]; s_H[4] = s_sum[3]; s_H[5] = s_sum[4];
      s_H[6] = s_sum[2]; s_H[7] = s_sum[4]; s_H[8] = s_sum[5];
      s_rhs[0] = s_sum[6]; s_rhs[1] = s_sum[7]; s_rhs[2] = s_sum[8];
      if (A.prior) add_prior(A.prior[a], s_pose, s_H, s_rhs);
      for (int k = 0; k < 9; ++k) S.H_last[9 * a + k] = s_H[k];
      if (!solve_update(s_H, s_rhs, A.damping, s_pose)) status = LSM2D_SINGULAR_H;
      else {
        S.pose[3 * a] = s_pose[0]; S.pose[3 * a + 1] = s_pose[1]; S.pose[3 * a + 2] = s_pose[2];
        if (A.term_eps > 0.0f) {       // as in k_align; the previous iteration's statistics wait in S.last
          const float chi_now = last.chi_in + last.chi_out;
          if (S.it > ph_start) { const StatsDev pv = S.last[a]; stop_now = __builtin_fabsf((pv.chi_in + pv.chi_out) - chi_now) < A.term_eps * chi_now; }
          S.last[a] = last;
        }
      }
    }
    bool last_it = S.it + 1 >= ph_end || stop_now;
    if (status == LSM2D_RUNNING && last_it && A.inlier_runs && ph == 0 && last.n_in >= A.min_inliers) {      // as in k_align: on to the inlier-only runs
      S.phase[3 * a] = 1; S.phase[3 * a + 1] = S.it + 1; S.phase[3 * a + 2] = S.it + 1 + A.max_it; last_it = false;
    }
    if (status == LSM2D_RUNNING && last_it) status = last.n_in < A.min_inliers ? LSM2D_NOT_ENOUGH_INLIERS : LSM2D_SUCCESS;
    if (status != LSM2D_RUNNING) {
      S.done[a] = 1;
      A.out_status[a] = status;
      A.out_pose[3 * a] = s_pose[0]; A.out_pose[3 * a + 1] = s_pose[1]; A.out_pose[3 * a + 2] = s_pose[2];
      if (A.out_H) for (int k = 0; k < 9; ++k) A.out_H[9 * a + k] = S.it == 0 && !s_active ? 0.0f : S.H_last[9 * a + k];
      if (A.out_its) A.out_its[a] = S.it + 1;
    }
  }
}

// ---- finder-level: one (fixed, moving, pose) -> pairs in ascending column ------------------------
struct FindArgs {
  CloudDev fixed, moving; int32_t fc, mc;
  ProjK proj; float point_distance, normal_cos;
  Iso T;
  int32_t* out_pairs;  // [cols][2]
  int32_t* out_count;
  const u64* fcan_global; const u64* mcan_global;      // a map-sized cloud's canvas, projected over many workgroups beforehand (k_project_split), or nullptr
  float inl_tau;         // > 0: only pairs whose factor is an inlier under a Cauchy robustifier of this threshold (chi^2 < tau) are emitted -- the aligner's
                         // keep_only_inlier_correspondences (lsm2d_align_batch_pairs); 0: every pair
};

__global__ __launch_bounds__(kFindBlock) void k_find_projective(const FindArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* mcan = reinterpret_cast<u64*>(smem);
  u64* fcan = mcan + A.proj.cols;
  __shared__ int s_wave_tot[kFindBlock / 64];
  __shared__ int s_base;
  const int tid = threadIdx.x;
  for (int i = tid; i < A.proj.cols; i += kFindBlock) { mcan[i] = kEmptyCell; fcan[i] = kEmptyCell; }
  if (tid == 0) s_base = 0;
  __syncthreads();
  const Iso ident = {1.0f, 0.0f, 0.0f, 0.0f};
  const int fbase = A.fixed.start[A.fc], mbase = A.moving.start[A.mc];
  if (A.fcan_global) { for (int i = tid; i < A.proj.cols; i += kFindBlock) fcan[i] = A.fcan_global[i]; }
  else project_cloud(A.fixed.xy + fbase, A.fixed.count[A.fc], ident, A.proj, fcan, tid, kFindBlock);
  if (A.mcan_global) { for (int i = tid; i < A.proj.cols; i += kFindBlock) mcan[i] = A.mcan_global[i]; }
  else project_cloud(A.moving.xy + mbase, A.moving.count[A.mc], A.T, A.proj, mcan, tid, kFindBlock);
  __syncthreads();
  SliceDev S; S.point_distance = A.point_distance; S.normal_cos = A.normal_cos;
  const int lane = tid & 63, wave = tid >> 6;
  for (int c0 = 0; c0 < A.proj.cols; c0 += kFindBlock) {
    const int col = c0 + tid;
    int fi = -1, mi = -1; float2 nf, nm; bool ok = false;
    if (col < A.proj.cols) ok = match_bin(fcan[col], mcan[col], S, A.T, A.fixed.nrm + fbase, A.moving.nrm + mbase, fi, mi, nf, nm);
    if (ok && A.inl_tau > 0.0f) ok = pair_chi(A.T, A.fixed.xy[fbase + fi], nf, A.moving.xy[mbase + mi], nm) < A.inl_tau;
    // order-preserving compaction: ballot prefix inside the wave, wave totals through LDS
    const u64 bal = __ballot(ok);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int before = s_base, total = 0;
    for (int w = 0; w < kFindBlock / 64; ++w) { const int t = s_wave_tot[w]; if (w < wave) before += t; total += t; }
    if (ok) { A.out_pairs[2 * (before + prefix)] = fi; A.out_pairs[2 * (before + prefix) + 1] = mi; }
    __syncthreads();
    if (tid == 0) s_base += total;
    __syncthreads();
  }
  if (tid == 0) *A.out_count = s_base;
}

// ---- finder-level NN: pairs in ascending moving index (correspondence_finder_kd_tree_2d.cpp:12-27) ------
struct FindNNArgs {
  CloudDev fixed, moving; int32_t fc, mc; int32_t use_distmap; int32_t use_kd;      // at most one of the two set; neither: the exact grid search
  float max_distance, normal_cos; Iso T; int32_t nn_group;
  int32_t* out_pairs; int32_t* out_count;
  int32_t* match; int32_t* block_count;      // k_find_nn_multi: per query the matched fixed index or -1; pairs per workgroup
  float inl_tau;                              // as FindArgs::inl_tau
};

__global__ __launch_bounds__(kFindBlock) void k_find_nn(const FindNNArgs A) {
  __shared__ int s_wave_tot[kFindBlock / 64];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  const int fbase = A.fixed.start[A.fc], mbase = A.moving.start[A.mc], n = A.moving.count[A.mc];
  GridMeta g; DistMeta dm;
  const int32_t* cst = nullptr; const int32_t* sidx = nullptr; const float2* sxy = nullptr;
  const KdNode* knd = nullptr;
  if (A.use_distmap) dm = A.fixed.dist.meta[A.fc];
  else if (A.use_kd) { knd = A.fixed.kd.nodes + A.fixed.kd.meta[A.fc].node_base; sxy = A.fixed.kd.leaf_xy + fbase; sidx = A.fixed.kd.leaf_idx + fbase; }
  else {
    g = A.fixed.grid.meta[A.fc]; cst = A.fixed.grid.cell_start + g.cell_base;
    sidx = A.fixed.grid.sorted_idx + fbase; sxy = A.fixed.grid.sorted_xy + fbase;
  }
  const float md2 = A.max_distance * A.max_distance;
  const int group = (A.use_distmap || A.use_kd) ? 1 : A.nn_group, sub = tid & (group - 1);
  const int per_step = kFindBlock / group;
  auto query = [&](float qx, float qy) {
    if (A.use_kd) return kd_query(knd, sxy, sidx, qx, qy, md2);
    return group == kNNGroup ? nn_query<kNNGroup>(g, cst, sidx, sxy, qx, qy, A.max_distance, md2, sub)
                             : nn_query<1>(g, cst, sidx, sxy, qx, qy, A.max_distance, md2, sub);
  };
  for (int j0 = 0; j0 < n; j0 += per_step) {
    const int j = j0 + tid / group;
    int best = -1; bool ok = false;
    if (j < n) {
      const float2 pm = A.moving.xy[mbase + j];
      float qx, qy; xf_point(A.T, pm.x, pm.y, qx, qy);
      best = A.use_distmap ? distmap_lookup(dm, A.fixed.dist.parent, qx, qy) : query(qx, qy);
      if (best >= 0 && sub == 0) {
        const float2 nm = A.moving.nrm[mbase + j], nf = A.fixed.nrm[fbase + best];
        float nqx, nqy; xf_normal(A.T, nm.x, nm.y, nqx, nqy);
        ok = !(__builtin_fmaf(nqx, nf.x, nqy * nf.y) < A.normal_cos);
        if (ok && A.inl_tau > 0.0f) ok = pair_chi(A.T, A.fixed.xy[fbase + best], nf, pm, nm) < A.inl_tau;
      }
    }
    // lanes are in ascending query order (tid / group), so the ballot compaction keeps ascending moving index
    const u64 bal = __ballot(ok);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int before = s_base, total = 0;
    for (int w = 0; w < kFindBlock / 64; ++w) { const int t = s_wave_tot[w]; if (w < wave) before += t; total += t; }
    if (ok) { A.out_pairs[2 * (before + prefix)] = best; A.out_pairs[2 * (before + prefix) + 1] = j; }
    __syncthreads();
    if (tid == 0) s_base += total;
    __syncthreads();
  }
  if (tid == 0) *A.out_count = s_base;
}

// The same finder over many workgroups (more queries than one workgroup takes in one trip: a map-sized moving cloud against a scan's
// structure is 98 trips of one workgroup otherwise).  Workgroup b owns the queries [b * per_step, (b + 1) * per_step), ascending.
// Phase 0: search, normal gate, match[j] = fixed index or -1, pairs per workgroup.  Phase 1 (a second launch of the same shape): every
// workgroup adds up the counts in front of it, ranks its own pairs by ballot and writes them -- ascending moving index, as the
// reference emits them (correspondence_finder_kd_tree_2d.cpp:12-27, correspondence_finder_nn_2d.cpp:63-80).
template <int kPhase>
__global__ __launch_bounds__(kFindBlock) void k_find_nn_multi(const FindNNArgs A) {
  __shared__ int s_wave_tot[kFindBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = A.moving.count[A.mc];
  const int group = (A.use_distmap || A.use_kd) ? 1 : A.nn_group, sub = tid & (group - 1);
  const int per_step = kFindBlock / group;
  const int j = blockIdx.x * per_step + tid / group;
  if (kPhase == 0) {
    const int fbase = A.fixed.start[A.fc], mbase = A.moving.start[A.mc];
    int best = -1; bool ok = false;
    if (j < n) {
      const float2 pm = A.moving.xy[mbase + j];
      float qx, qy; xf_point(A.T, pm.x, pm.y, qx, qy);
      if (A.use_distmap) best = distmap_lookup(A.fixed.dist.meta[A.fc], A.fixed.dist.parent, qx, qy);
      else if (A.use_kd) {
        best = kd_query(A.fixed.kd.nodes + A.fixed.kd.meta[A.fc].node_base, A.fixed.kd.leaf_xy + fbase, A.fixed.kd.leaf_idx + fbase, qx, qy, A.max_distance * A.max_distance);
      } else {
        const GridMeta g = A.fixed.grid.meta[A.fc];
        const int32_t* cst = A.fixed.grid.cell_start + g.cell_base; const int32_t* sidx = A.fixed.grid.sorted_idx + fbase; const float2* sxy = A.fixed.grid.sorted_xy + fbase;
        const float md2 = A.max_distance * A.max_distance;
        best = group == kNNGroup ? nn_query<kNNGroup>(g, cst, sidx, sxy, qx, qy, A.max_distance, md2, sub) : nn_query<1>(g, cst, sidx, sxy, qx, qy, A.max_distance, md2, sub);
      }
      if (best >= 0 && sub == 0) {
        const float2 nm = A.moving.nrm[mbase + j], nf = A.fixed.nrm[fbase + best];
        float nqx, nqy; xf_normal(A.T, nm.x, nm.y, nqx, nqy);
        ok = !(__builtin_fmaf(nqx, nf.x, nqy * nf.y) < A.normal_cos);
        if (ok && A.inl_tau > 0.0f) ok = pair_chi(A.T, A.fixed.xy[fbase + best], nf, pm, nm) < A.inl_tau;
      }
      if (sub == 0) A.match[j] = ok ? best : -1;
    }
    const u64 bal = __ballot(ok);
    if (lane == 0) s_wave_tot[wave] = __popcll(bal);
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < kFindBlock / 64; ++w) t += s_wave_tot[w]; A.block_count[blockIdx.x] = t; }
  } else {
    __shared__ int s_before;
    if (tid == 0) s_before = 0;
    __syncthreads();
    int mine = 0;
    for (int b = tid; b < (int) blockIdx.x; b += kFindBlock) mine += A.block_count[b];
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0 && mine) atomicAdd(&s_before, mine);
    const int best = (j < n && sub == 0) ? A.match[j] : -1;
    const bool ok = best >= 0;
    const u64 bal = __ballot(ok);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int before = s_before, total = 0;
    for (int w = 0; w < kFindBlock / 64; ++w) { const int t = s_wave_tot[w]; if (w < wave) before += t; total += t; }
    if (ok) { A.out_pairs[2 * (before + prefix)] = best; A.out_pairs[2 * (before + prefix) + 1] = j; }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *A.out_count = s_before + total;
  }
}

// ---- projector-level: canvas of one cloud --------------------------------------------------------
struct ProjectArgs {
  CloudDev cloud; int32_t ci; ProjK proj; Iso T;
  int32_t* out_src; float* out_depth; float4* out_xynn;
};

__global__ __launch_bounds__(kFindBlock) void k_project_canvas(const ProjectArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* can = reinterpret_cast<u64*>(smem);
  const int tid = threadIdx.x;
  for (int i = tid; i < A.proj.cols; i += kFindBlock) can[i] = kEmptyCell;
  __syncthreads();
  const int base = A.cloud.start[A.ci];
  project_cloud(A.cloud.xy + base, A.cloud.count[A.ci], A.T, A.proj, can, tid, kFindBlock);
  __syncthreads();
  for (int col = tid; col < A.proj.cols; col += kFindBlock) {
    const u64 k = can[col];
    int src = -1; float depth = 3.402823466e+38f; float4 t = {0.0f, 0.0f, 0.0f, 0.0f};
    if (k != kEmptyCell) {
      src = (int) (uint32_t) k; depth = __uint_as_float((uint32_t) (k >> 32));
      const float2 p = A.cloud.xy[base + src], n = A.cloud.nrm[base + src];
      xf_point(A.T, p.x, p.y, t.x, t.y);
      xf_normal(A.T, n.x, n.y, t.z, t.w);
    }
    if (A.out_src) A.out_src[col] = src;
    if (A.out_depth) A.out_depth[col] = depth;
    if (A.out_xynn) A.out_xynn[col] = t;
  }
}

// ---- factor-level: H, b, stats for a given correspondence vector ----------------------------------
struct LinArgs {
  CloudDev fixed, moving; int32_t fc, mc;
  const int32_t* pairs; int32_t n_pairs;
  Iso T; int32_t cauchy; float tau;
  float* partial;     // [n_blocks][kAccumWords]
  float* out;         // [kAccumWords]
  unsigned long long* dig;      // the pairs' digest (lsm2d_iteration_stats.pair_digest, slice 0), zeroed by the host: every workgroup adds its share
};

__global__ __launch_bounds__(256) void k_linearize_partial(const LinArgs A) {
  __shared__ float red[4 * kAccumWords];
  __shared__ u64 s_dig;
  const int tid = threadIdx.x;
  const int fbase = A.fixed.start[A.fc], mbase = A.moving.start[A.mc];
  Accum acc; accum_zero(acc);
  if (tid == 0) s_dig = 0ull;
  __syncthreads();
  u64 dg = 0ull;
  for (int k = blockIdx.x * 256 + tid; k < A.n_pairs; k += gridDim.x * 256) {
    const int fi = A.pairs[2 * k], mi = A.pairs[2 * k + 1];
    dg += pair_hash_dev(0u, (uint32_t) fi, (uint32_t) mi);
    accumulate_pair(A.T, A.fixed.xy[fbase + fi], A.fixed.nrm[fbase + fi], A.moving.xy[mbase + mi], A.moving.nrm[mbase + mi],
                    A.cauchy != 0, A.tau, acc);
  }
  if (dg) atomicAdd(reinterpret_cast<unsigned long long*>(&s_dig), (unsigned long long) dg);
  block_reduce_store(acc, red, tid);
  __syncthreads();
  if (tid == 0) {
    if (A.dig && s_dig) atomicAdd(A.dig, (unsigned long long) s_dig);
    Accum t; block_reduce_gather(red, 4, t);
    float* p = A.partial + (size_t) blockIdx.x * kAccumWords;
    p[0] = t.h00; p[1] = t.h01; p[2] = t.h02; p[3] = t.h11; p[4] = t.h12; p[5] = t.h22; p[6] = t.b0; p[7] = t.b1; p[8] = t.b2;
    p[9] = t.chi_in; p[10] = t.chi_out; p[11] = __int_as_float(t.n_in); p[12] = __int_as_float(t.n_out); p[13] = __int_as_float(t.n_corr);
  }
}

// "sum_order" 1: the same factor with the sums formed pair after pair in the order of the correspondence vector (the reference's loop): ONE workgroup,
// trips of kAlignBlock consecutive pairs, their terms as records in LDS, eleven lanes of wave 0 adding them in ascending position (lsm2d_device.h)
__global__ __launch_bounds__(kAlignBlock) void k_linearize_seq(const LinArgs A) {
  __shared__ __attribute__((aligned(16))) float s_rec[kSeqHalf * kSeqFields];
  __shared__ float red[(kAlignBlock / 64) * kAccumWords];
  __shared__ u64 s_dig;
  const int tid = threadIdx.x;
  const int fbase = A.fixed.start[A.fc], mbase = A.moving.start[A.mc];
  Accum acc; accum_zero(acc);
  float seq_acc = 0.0f;
  if (tid == 0) s_dig = 0ull;
  __syncthreads();
  u64 dg = 0ull;
  for (int k0 = 0; k0 < A.n_pairs; k0 += kAlignBlock) {
    const int k = k0 + tid;
    float t[kSeqFields]; seq_zero(t);
    if (k < A.n_pairs) {
      const int fi = A.pairs[2 * k], mi = A.pairs[2 * k + 1];
      dg += pair_hash_dev(0u, (uint32_t) fi, (uint32_t) mi);
      bool inl; pair_terms(A.T, A.fixed.xy[fbase + fi], A.fixed.nrm[fbase + fi], A.moving.xy[mbase + mi], A.moving.nrm[mbase + mi], A.cauchy != 0, A.tau, false, t, inl);
      ++acc.n_corr; acc.n_in += inl ? 1 : 0; acc.n_out += inl ? 0 : 1;
    }
    const int n_rec = A.n_pairs - k0 < kAlignBlock ? A.n_pairs - k0 : kAlignBlock;
    for (int h0 = 0; h0 < n_rec; h0 += kSeqHalf) {
      if (tid >= h0 && tid < h0 + kSeqHalf) seq_store(s_rec, tid - h0, t);
      __syncthreads();
      const int left = n_rec - h0;
      if (tid < 64) seq_acc = seq_walk(s_rec, left < kSeqHalf ? left : kSeqHalf, tid, seq_acc);
      __syncthreads();
    }
  }
  if (dg) atomicAdd(reinterpret_cast<unsigned long long*>(&s_dig), (unsigned long long) dg);
  block_reduce_store(acc, red, tid);
  __syncthreads();
  if (tid < 64) {
    float v; int vi; block_reduce_gather_lane(red, kAlignBlock / 64, tid, v, vi);
    const float tot = seq_total(seq_acc, tid);
    if (tid < 11) A.out[tid] = tot;
    else if (tid < kAccumWords) A.out[tid] = __int_as_float(vi);
    if (tid == 0 && A.dig) *A.dig = (unsigned long long) s_dig;
  }
}

__global__ void k_linearize_final(const float* partial, int n_blocks, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Accum t; block_reduce_gather(partial, n_blocks, t);     // fixed block order => deterministic
  out[0] = t.h00; out[1] = t.h01; out[2] = t.h02; out[3] = t.h11; out[4] = t.h12; out[5] = t.h22; out[6] = t.b0; out[7] = t.b1; out[8] = t.b2;
  out[9] = t.chi_in; out[10] = t.chi_out; out[11] = __int_as_float(t.n_in); out[12] = __int_as_float(t.n_out); out[13] = __int_as_float(t.n_corr);
}
