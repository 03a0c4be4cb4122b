// lsm2d_k_align_pair.h -- k_align_pair: the latency kernel (one or two projective slices side by side; the live tracker's call).
// Part of lsm2d_kernels.h (included there, inside namespace lsm2d, in this order); not a translation unit of its own.
// ---- the latency kernel: one alignment per workgroup, tuned for calls that cannot fill the chip ------------------
// (one or two projective slices; with two, side by side)
// The live tracker's aligner has two laser slices (front and rear scanner, MULTI.json:396-401) and runs one alignment at a
// time: with one workgroup per alignment the chip is empty and the call is a chain of latencies on ONE compute unit, where sixteen
// waves share four SIMDs -- so the kernel is built around (i) the number of instructions all waves issue per iteration and (ii) the
// length of the stretch only one wave can run (sums -> 3x3 solve -> next transforms).  1024 threads: waves 0-7 own slice 0, waves
// 8-15 slice 1; every thread keeps the COLUMNS it has in k_align (thread = tid mod 512 of its slice: col, col + 512, ...), the
// per-wave sums are the same 64-leaf trees, gathered in the same wave order, the slice totals added in slice order, the prior's
// terms and the solve are the same IEEE operations -- the sums, hence the poses, have k_align's bits (tests: fused == latency kernel).
//   * a moving cloud of <= 1024 points (the tracker's clipped scene: one point per column) lives in LDS, coordinates and normal
//     in one 16-byte row: the bin walk's gather of the moving winner is one LDS read instead of two dependent global loads; its
//     coordinates also sit in registers, ONE point per thread (the few beyond 512 go to the highest lanes), so an iteration's
//     projection is one point's chain per thread with no load in front;
//   * wave totals: the eleven sums go through the DPP tree level by level (independent instructions back to back: no wait
//     states between a VALU write and the DPP read of it), the three counts through ballots and scalar popcounts; counts travel
//     as exact floats so that the gather is one add per word;
//   * the serial stretch runs on wave 0 as a VECTOR: lane q owns quantity q (6 of H, 3 of b, 2 chi, 3 counts; lanes 32-40 the
//     nine entries of the information matrix handed back), gathers it over waves and slices, adds ITS term of the prior -- which
//     every lane computed for itself before the barrier, while the other waves were still summing -- and only the nine inputs of
//     the 3x3 solve are broadcast (v_readlane).  The solve has no early exits (a failed pivot is a flag), the pose's sine and
//     cosine are ready before the barrier, and lanes 0 / 1 turn the new pose into the slices' transforms side by side.
static constexpr int kPairBlock = 2 * kAlignBlock;
static constexpr int kPairMovCap = 2 * kAlignBlock;     // moving points per slice kept on chip
static constexpr int kPairRedStride = 16;               // words per (slice, wave) record: 11 sums, n_in, n_out, n_corr (exact floats), 2 spare = one 64-byte row

// all threads of a slice call; afterwards red[wave][0..13] holds the wave's totals.  count_bits: bits a thread's counts can occupy
// (a thread accumulates at most ceil(cols / 512) pairs).  A slice without robustifier has chi_out == +0 and n_in == n_corr in
// every lane: nothing to add up.
LSM2D_DEV void pair_wave_sums(const Accum& A, float* red, int tid, bool cauchy, int count_bits) {
  const int lane = tid & 63, wave = tid >> 6;
  float f[11] = {A.h00, A.h01, A.h02, A.h11, A.h12, A.h22, A.b0, A.b1, A.b2, A.chi_in, A.chi_out};
  int nc = 0, ni = 0;
  if (cauchy) {
    wave_tree63<11>(f);
    for (int b = 0; b < count_bits; ++b) {
      nc += __builtin_popcountll(__ballot((A.n_corr >> b) & 1)) << b;
      ni += __builtin_popcountll(__ballot((A.n_in >> b) & 1)) << b;
    }
  } else {
    wave_tree63<10>(f);
    for (int b = 0; b < count_bits; ++b) nc += __builtin_popcountll(__ballot((A.n_corr >> b) & 1)) << b;
    ni = nc; f[10] = 0.0f;
  }
  if (lane == 63) {
    float4* r = reinterpret_cast<float4*>(red + wave * kPairRedStride);
    r[0] = make_float4(f[0], f[1], f[2], f[3]); r[1] = make_float4(f[4], f[5], f[6], f[7]);
    r[2] = make_float4(f[8], f[9], f[10], (float) ni);
    *reinterpret_cast<float2*>(r + 3) = make_float2((float) (nc - ni), (float) nc);
  }
}

// ONE entry of the odometry prior's J^T Omega [J | e] (prior_apply's operations for that entry, in its order): row r in 0..2, column c in
// 0..2 of the H term, c == 3 the b term.  J = [[cs, -sn, 0], [sn, cs, 0], [0, 0, 1]].
LSM2D_DEV float prior_term_lane(const PriorDev& Pz, const float pose[3], int r, int c) {
  float E[3]; compose(Pz.cz, Pz.sz, Pz.z_inv, pose, E);
  float cs, sn; sincos_fixed(E[2], sn, cs);
  const float msn = -sn;
  const float x0 = c == 0 ? cs : (c == 1 ? msn : (c == 2 ? 0.0f : E[0]));
  const float x1 = c == 0 ? sn : (c == 1 ? cs : (c == 2 ? 0.0f : E[1]));
  const float x2 = c == 2 ? 1.0f : (c == 3 ? E[2] : 0.0f);
  const float j0 = r == 0 ? cs : (r == 1 ? msn : 0.0f);
  const float j1 = r == 0 ? sn : (r == 1 ? cs : 0.0f);
  const float j2 = r == 2 ? 1.0f : 0.0f;
  float oj[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { float v = 0.0f; v += Pz.omega[3 * k + 0] * x0; v += Pz.omega[3 * k + 1] * x1; v += Pz.omega[3 * k + 2] * x2; oj[k] = v; }
  float v = 0.0f; v += j0 * oj[0]; v += j1 * oj[1]; v += j2 * oj[2];
  return v;
}

// solve_update's LDL^T without early exits: the same operations on the same values whenever it succeeds; a failed pivot or a
// non-finite step is reported at the end (what was computed behind it is discarded by the caller, as solve_update's return does)
LSM2D_DEV bool solve_flat(float h00, float h01, float h02, float h11, float h12, float h22, float b0, float b1, float b2, float damping,
                          float& dx, float& dy, float& dth) {
  const double a00 = (double) h00 + (double) damping, a01 = h01, a02 = h02;
  const double a11 = (double) h11 + (double) damping, a12 = h12, a22 = (double) h22 + (double) damping;
  const double r0 = -(double) b0, r1 = -(double) b1, r2 = -(double) b2;
  const double d0 = a00;
  const double l10 = a01 / d0, l20 = a02 / d0;
  const double d1 = a11 - l10 * a01;
  const double l21 = (a12 - l20 * a01) / d1;
  const double d2 = a22 - l20 * a02 - l21 * l21 * d1;
  const double y0 = r0, y1 = r1 - l10 * y0, y2 = r2 - l20 * y0 - l21 * y1;
  const double z2 = y2 / d2;
  const double z1 = y1 / d1 - l21 * z2;
  const double z0 = y0 / d0 - l10 * z1 - l20 * z2;
  dx = (float) z0; dy = (float) z1; dth = (float) z2;
  return (d0 > 0) & (d1 > 0) & (d2 > 0) & (bool) __builtin_isfinite(d0) & (bool) __builtin_isfinite(d1) & (bool) __builtin_isfinite(d2) &
         (bool) __builtin_isfinite(z0) & (bool) __builtin_isfinite(z1) & (bool) __builtin_isfinite(z2);
}

#ifndef LSM2D_PAIR_READ_FIRST
#define LSM2D_PAIR_READ_FIRST false      // z-buffer updates of the on-chip moving cloud: fire-and-forget (neighbouring lanes hold neighbouring columns' points)
#endif

__global__ __launch_bounds__(kPairBlock) void k_align_pair(const AlignArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int nwaves = kAlignBlock / 64;
  float4* fwin = reinterpret_cast<float4*>(smem);                                 // 16-byte rows first (alignment)
  float4* mwin2 = fwin + A.fcan_total;                                             // [n_slices][pair_mov_cap]: the moving clouds, (x, y, nx, ny)
  float4* fall2 = mwin2 + A.n_slices * A.pair_mov_cap;                             // [n_slices][pair_fix_cap]: the fixed clouds
  float* red2 = reinterpret_cast<float*>(fall2 + A.n_slices * A.pair_fix_cap);     // [n_slices][nwaves][kPairRedStride]
  u64* mcan2 = reinterpret_cast<u64*>(red2 + A.n_slices * nwaves * kPairRedStride);      // [n_slices][cols_max]: one moving canvas per slice
  u64* fcan = mcan2 + A.n_slices * A.cols_max;
  __shared__ Iso   s_iso[2];
  __shared__ int   s_done, s_inl;      // s_inl: the iteration about to run belongs to the inlier-only runs (enable_inlier_only_runs)
  __shared__ u64   s_dig;              // the iteration's pair digest, as in k_align
  __shared__ PriorDev s_prior;

  const int a = blockIdx.x, gtid = threadIdx.x, nthr = kAlignBlock * A.n_slices;      // launched with 512 threads per slice (one or two slices)
#ifdef LSM2D_PHASE_CLOCKS      // debug build: where one alignment's time goes (10 ns ticks), printed by thread 0
  unsigned long long pc_t = __builtin_amdgcn_s_memrealtime(), pc_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define LSM2D_PC(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); pc_acc[k] += n_ - pc_t; pc_t = n_; } while (0)
#else
#define LSM2D_PC(k) do { } while (0)
#endif
  const int half = __builtin_amdgcn_readfirstlane(gtid >> 9);      // wave-uniform: the slice this wave works for
  const bool w0 = __builtin_amdgcn_readfirstlane(gtid >> 6) == 0;  // wave 0: the serial stretch
  const int tid = gtid & (kAlignBlock - 1), lane = gtid & 63;
  u64* mcan = mcan2 + half * A.cols_max;
  float* red = red2 + half * nwaves * kPairRedStride;
  // ---- everything that comes from memory is asked for first: the clouds' places and sizes, then this thread's rows of both clouds (the
  //      fixed one possibly still in the host's pinned upload buffer: SliceDev::unpack_src) -- in flight while the canvases are cleared
  const SliceDev& S = A.s[half];
  const int mc = pick_cloud(S.moving, a), fc = pick_cloud(S.fixed, a);
  const bool unpack = A.inline_n1 && S.unpack_src;
  const int mbase = S.moving.start[mc], fbase = S.fixed.start[fc];
  const int m_count = S.moving.count[mc], f_count = unpack ? S.unpack_n : S.fixed.count[fc];
  const float2* mn = S.moving.nrm + mbase; const float2* mp = S.moving.xy + mbase;
  // clouds on chip (see the head comment): the moving one at most two points per thread (coordinates stay in registers), the fixed one as many rows as LDS has
  const bool m_on_chip = !S.moving.lane_xy && m_count <= A.pair_mov_cap;      // pair_mov_cap: kPairMovCap, or 0 when LDS has no room
  // (workgroup-uniform: the branch below holds a barrier.  pair_fix_cap > 0 means the host sized the rows for the LARGEST fixed cloud of every slice, so
  // both halves take the same side; with pair_fix_cap == 0 an empty fixed cloud must not count as "on chip" while the other slice's is not)
  const bool f_on_chip = A.pair_fix_cap > 0 && f_count <= A.pair_fix_cap;
  float4* mwin = mwin2 + half * A.pair_mov_cap;
  float4* fall = fall2 + half * A.pair_fix_cap;
  const int j1 = kPairMovCap - 1 - tid;                   // this thread's second moving point, if the cloud has more than 512
  float2 p0 = make_float2(0.0f, 0.0f), p1 = p0, n0 = p0, n1 = p0;
  if (m_on_chip) {
    if (tid < m_count) { p0 = mp[tid]; n0 = mn[tid]; }
    if (j1 < m_count) { p1 = mp[j1]; n1 = mn[j1]; }
  }
  auto fixed_row = [&](int i) {
    if (unpack) return S.unpack_src[i];
    const float2 p = S.fixed.xy[fbase + i], n = S.fixed.nrm[fbase + i];
    return make_float4(p.x, p.y, n.x, n.y);
  };
  float4 frow = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (f_on_chip && tid < f_count) frow = fixed_row(tid);
  constexpr int kPriorWords = (int) (sizeof(PriorDev) / sizeof(float));
  if (A.prior && gtid >= 64 && gtid < 64 + kPriorWords)
    ((float*) &s_prior)[gtid - 64] = A.inline_n1 ? ((const float*) &A.prior1)[gtid - 64] : ((const float*) (A.prior + a))[gtid - 64];
  if (unpack && !f_on_chip) unpack_fixed_set(S, tid, kAlignBlock);      // visible after the barrier below
  for (int i = gtid; i < A.fcan_total; i += nthr) fcan[i] = kEmptyCell;
  for (int i = gtid; i < A.n_slices * A.cols_max; i += nthr) mcan2[i] = kEmptyCell;

  // wave 0's state: the estimate (the same value in every lane), each lane's quantity and its entry of the prior, lanes 0 / 1 their slice's sensor offset
  float pose[3] = {0.0f, 0.0f, 0.0f}, hl = 0.0f;
  int status = LSM2D_RUNNING, last_n_in = 0;
  float prev_chi = 0.0f;            // total chi^2 of the previous iteration (termination_chi_epsilon)
  int phase = 0, phase_start = 0, phase_end = A.max_it;      // as in k_align
  int q = 15, pr = 0, pc = 0; bool has_pterm = false;
  float kS[3] = {0.0f, 0.0f, 0.0f}, kc = 1.0f, ks = 0.0f; int khs = 0;
  if (w0) {
    if (A.inline_n1) { pose[0] = A.pose1[0]; pose[1] = A.pose1[1]; pose[2] = A.pose1[2]; }
    else { pose[0] = A.init_pose[3 * a + 0]; pose[1] = A.init_pose[3 * a + 1]; pose[2] = A.init_pose[3 * a + 2]; }
    if (lane < 16) {
      q = lane;
      if (lane < 9) { has_pterm = true; pr = (int) ((0x210211000ull >> (4 * lane)) & 15); pc = (int) ((0x333221210ull >> (4 * lane)) & 15); }
    } else if (lane >= 32 && lane < 41) {
      const int k = lane - 32;
      q = (int) ((0x542431210ull >> (4 * k)) & 15); has_pterm = true; pr = k / 3; pc = k - 3 * pr;
    }
    const SliceDev& Sa = A.s[0]; const SliceDev& Sb = A.s[A.n_slices - 1];
    const bool second = lane == 1;
    kS[0] = second ? Sb.Sinv[0] : Sa.Sinv[0]; kS[1] = second ? Sb.Sinv[1] : Sa.Sinv[1]; kS[2] = second ? Sb.Sinv[2] : Sa.Sinv[2];
    kc = second ? Sb.cSinv : Sa.cSinv; ks = second ? Sb.sSinv : Sa.sSinv; khs = second ? Sb.has_sensor : Sa.has_sensor;
    if (lane < A.n_slices) s_iso[lane] = slice_iso_of(khs, kc, ks, kS, pose);
    if (lane == 0) { s_done = 0; s_inl = 0; s_dig = 0ull; }
    if (A.out_last_pose && lane < 3) A.out_last_pose[3 * a + lane] = lane == 0 ? pose[0] : (lane == 1 ? pose[1] : pose[2]);
  }
  __syncthreads();                  // canvases cleared, prior and first transforms in LDS
  PriorDev pz;                      // wave 0's copy of the prior, in registers
  if (w0 && A.prior) pz = s_prior;
  if (m_on_chip) {
    if (tid < m_count) mwin[tid] = make_float4(p0.x, p0.y, n0.x, n0.y);
    if (j1 < m_count) mwin[j1] = make_float4(p1.x, p1.y, n1.x, n1.y);
  }
  const Iso ident = {1.0f, 0.0f, 0.0f, 0.0f};
  if (f_on_chip) {
    // the fixed cloud: every row into LDS (the bin walk reads the winner's row there: no table of winners, no pass to fill it), a set that
    // was still in the upload buffer also into its arrays (later consumers find them there), and into the z-buffer -- project_cloud's
    // operations per point (project_point with the identity)
    float2* oxy = const_cast<float2*>(S.fixed.xy) + fbase; float2* onr = const_cast<float2*>(S.fixed.nrm) + fbase;
    for (int i = tid; i < f_count; i += kAlignBlock) {
      if (i != tid) frow = fixed_row(i);
      fall[i] = frow;
      if (unpack) { oxy[i] = make_float2(frow.x, frow.y); onr[i] = make_float2(frow.z, frow.w); }
      project_point(ident, S.proj, frow.x, frow.y, i, fcan + S.fcan_offset);
    }
    if (unpack && tid == 0) *const_cast<int32_t*>(S.fixed.count) = S.unpack_n;
  } else {
    project_cloud(S.fixed.xy + fbase, f_count, ident, S.proj, fcan + S.fcan_offset, tid, kAlignBlock);
    __syncthreads();
    for (int col = tid; col < S.proj.cols; col += kAlignBlock) {
      const u64 k = fcan[S.fcan_offset + col];
      if (k != kEmptyCell) {
        const int fi = (int) (uint32_t) k;
        const float2 p = S.fixed.xy[fbase + fi], n = S.fixed.nrm[fbase + fi];
        fwin[S.fcan_offset + col] = make_float4(p.x, p.y, n.x, n.y);
      }
    }
  }
  // what the serial stretch reads from the kernel arguments, fetched once (an s_load and its wait per use otherwise)
  int min_corr0 = A.s[0].min_corr, min_corr1 = A.s[1].min_corr, n_slices = A.n_slices;
  unsigned long long prior_ptr = reinterpret_cast<unsigned long long>(A.prior);
  int term_eps_b = __float_as_int(A.term_eps), damping_b = __float_as_int(A.damping);
  StatsDev* out_stats = A.out_stats ? A.out_stats + (size_t) a * A.stats_stride : nullptr;
  asm volatile("" : "+s"(min_corr0), "+s"(min_corr1), "+s"(n_slices), "+s"(prior_ptr), "+s"(term_eps_b), "+s"(damping_b));
  asm volatile("" : "+v"(out_stats));
  const bool two_slices = n_slices == 2, has_prior = prior_ptr != 0;
  const float term_eps = __int_as_float(term_eps_b), damping = __int_as_float(damping_b);
  // ... and what every wave's projection and walk read, per slice
  const bool on_chip = m_on_chip && f_on_chip;
  ProjK Pk = S.proj;
  int k00_b = __float_as_int(Pk.K00), k01_b = __float_as_int(Pk.K01), r2lo_b = __float_as_int(Pk.r2lo), r2hi_b = __float_as_int(Pk.r2hi), kcols = Pk.cols;
  int pd_b = __float_as_int(S.point_distance), ncos_b = __float_as_int(S.normal_cos), tau_b = __float_as_int(S.tau);
  asm volatile("" : "+s"(k00_b), "+s"(k01_b), "+s"(r2lo_b), "+s"(r2hi_b), "+s"(kcols), "+s"(pd_b), "+s"(ncos_b), "+s"(tau_b));
  Pk.K00 = __int_as_float(k00_b); Pk.K01 = __int_as_float(k01_b); Pk.r2lo = __int_as_float(r2lo_b); Pk.r2hi = __int_as_float(r2hi_b); Pk.cols = kcols;
  const float k_pd = __int_as_float(pd_b), k_ncos = __int_as_float(ncos_b), k_tau = __int_as_float(tau_b);
  const int per_thread = (S.proj.cols + kAlignBlock - 1) / kAlignBlock;      // pairs a thread can accumulate
  const int count_bits = 32 - __builtin_clz(per_thread | 1);
  const bool cauchy = S.cauchy != 0;
  const bool want_dig = A.out_stats != nullptr;
  const uint32_t salt = (uint32_t) half * 0x632BE5ABu;
  const int it_cap = A.inlier_runs ? 2 * A.max_it : A.max_it;
  __syncthreads();
  LSM2D_PC(0);

  const u64* fcs = fcan + S.fcan_offset; const float4* fws = fwin + S.fcan_offset;
  int it = 0;
  for (; it < it_cap; ++it) {
    const Iso T = s_iso[half];
    const bool inl_only = A.inlier_runs && __builtin_amdgcn_readfirstlane(s_inl) != 0;
    Accum acc; accum_zero(acc);
    if (on_chip) {
      // both clouds in LDS (the tracker's case): one point's z-buffer update per thread, then the walk one column at a time -- every gather is
      // an LDS row, so there is no latency worth a second column in flight, and a wave without a second column does not walk through one
      if (tid < m_count) project_point<LSM2D_PAIR_READ_FIRST>(T, Pk, p0.x, p0.y, tid, mcan);
      if (j1 < m_count) project_point<LSM2D_PAIR_READ_FIRST>(T, Pk, p1.x, p1.y, j1, mcan);
      __syncthreads();
      LSM2D_PC(1);
      for (int col = tid; col < Pk.cols; col += kAlignBlock) {
        const u64 fk = fcs[col], mk = mcan[col];
        mcan[col] = kEmptyCell;
        const uint32_t fdb = (uint32_t) (fk >> 32), mdb = (uint32_t) (mk >> 32);       // an empty cell's depth bits are all ones, no depth's are
        if (fdb != 0xFFFFFFFFu && mdb != 0xFFFFFFFFu && !(__builtin_fabsf(__uint_as_float(fdb) - __uint_as_float(mdb)) > k_pd)) {
          const float4 m = mwin[(uint32_t) mk], f = fall[(uint32_t) fk];
          float nqx, nqy; xf_normal(T, m.z, m.w, nqx, nqy);
          if (!(__builtin_fmaf(nqx, f.z, nqy * f.w) < k_ncos)) {
            if (want_dig) digest_add(&s_dig, salt, (int) (uint32_t) fk, (int) (uint32_t) mk);
            accumulate_pair<true>(T, make_float2(f.x, f.y), make_float2(f.z, f.w), make_float2(m.x, m.y), make_float2(m.z, m.w), cauchy, k_tau, acc, inl_only);
          }
        }
      }
    } else {
    if (S.moving.lane_xy) project_cloud_lanes(S.moving.lane_xy + S.moving.lane_start[mc], S.moving.lane_T[mc], T, S.proj, mcan, tid, kAlignBlock);
    else if (m_on_chip) {         // the same points every iteration: no load, no wait
      if (tid < m_count) project_point<LSM2D_PAIR_READ_FIRST>(T, S.proj, p0.x, p0.y, tid, mcan);
      if (j1 < m_count) project_point<LSM2D_PAIR_READ_FIRST>(T, S.proj, p1.x, p1.y, j1, mcan);
    }
    else project_cloud(mp, m_count, T, S.proj, mcan, tid, kAlignBlock);
    __syncthreads();
    LSM2D_PC(1);
    // k_align's bin walk, same thread <-> column mapping and order (col, then col + 512, ...), two columns per trip: both
    // columns' gathers of the moving winner are in flight together
    for (int col = tid; col < S.proj.cols; col += 2 * kAlignBlock) {
      const int col1 = col + kAlignBlock;
      const bool in1 = col1 < S.proj.cols;
      const u64 fk0 = fcs[col], mk0 = mcan[col];
      const u64 fk1 = in1 ? fcs[col1] : kEmptyCell, mk1 = in1 ? mcan[col1] : kEmptyCell;
      mcan[col] = kEmptyCell;
      if (in1) mcan[col1] = kEmptyCell;
      auto depth_gate = [&](u64 fk, u64 mk) {
        if (mk == kEmptyCell || fk == kEmptyCell) return false;
        const float fd = __uint_as_float((uint32_t) (fk >> 32)), md = __uint_as_float((uint32_t) (mk >> 32));
        return !(__builtin_fabsf(fd - md) > S.point_distance);
      };
      const bool g0 = depth_gate(fk0, mk0), g1 = depth_gate(fk1, mk1);
      const int mi0 = g0 ? (int) (uint32_t) mk0 : 0, mi1 = g1 ? (int) (uint32_t) mk1 : 0;
      float2 nm0, pm0, nm1, pm1;
      if (m_on_chip) {
        if (g0) { const float4 m = mwin[mi0]; pm0 = make_float2(m.x, m.y); nm0 = make_float2(m.z, m.w); }
        if (g1) { const float4 m = mwin[mi1]; pm1 = make_float2(m.x, m.y); nm1 = make_float2(m.z, m.w); }
      } else {
        if (g0) { nm0 = mn[mi0]; pm0 = mp[mi0]; }
        if (g1) { nm1 = mn[mi1]; pm1 = mp[mi1]; }
      }
      if (g0) {
        const float4 f = f_on_chip ? fall[(uint32_t) fk0] : fws[col];
        float nqx, nqy; xf_normal(T, nm0.x, nm0.y, nqx, nqy);
        if (!(__builtin_fmaf(nqx, f.z, nqy * f.w) < S.normal_cos)) {
          if (want_dig) digest_add(&s_dig, salt, (int) (uint32_t) fk0, mi0);
          accumulate_pair<true>(T, make_float2(f.x, f.y), make_float2(f.z, f.w), pm0, nm0, cauchy, S.tau, acc, inl_only);
        }
      }
      if (g1) {
        const float4 f = f_on_chip ? fall[(uint32_t) fk1] : fws[col1];
        float nqx, nqy; xf_normal(T, nm1.x, nm1.y, nqx, nqy);
        if (!(__builtin_fmaf(nqx, f.z, nqy * f.w) < S.normal_cos)) {
          if (want_dig) digest_add(&s_dig, salt, (int) (uint32_t) fk1, mi1);
          accumulate_pair<true>(T, make_float2(f.x, f.y), make_float2(f.z, f.w), pm1, nm1, cauchy, S.tau, acc, inl_only);
        }
      }
    }
    }
    LSM2D_PC(6);                 // thread 0's wave: bin walk
    pair_wave_sums(acc, red, tid, cauchy, count_bits);
    LSM2D_PC(7);                 // its wave sums
    // what depends on the pose alone, computed by wave 0 HERE, where it would otherwise wait for the slowest of the sixteen: each
    // lane's entry of the prior's terms, and the rotation of the update X <- X * v2t(dx)
    float P = 0.0f, sp = 0.0f, cp = 1.0f;
    if (w0) {
      sincos_fixed(pose[2], sp, cp);
      if (has_prior) P = prior_term_lane(pz, pose, pr, pc);
    }
    LSM2D_PC(9);                 // prior
    __syncthreads();
    LSM2D_PC(2);                 // waiting for the other waves
    if (w0) {
      // gather: this lane's quantity over the waves (wave order, from +0 -- block_reduce_gather's sums), both slices
      float v0 = 0.0f, v1 = 0.0f;
#pragma unroll
      for (int w = 0; w < nwaves; ++w) v0 += red2[w * kPairRedStride + q];
      const bool two = two_slices;
      if (two) {
#pragma unroll
        for (int w = 0; w < nwaves; ++w) v1 += red2[(nwaves + w) * kPairRedStride + q];
      }
      LSM2D_PC(3);
      // k_align's per-slice accumulation (zeroed sums, then slice 0, then slice 1; the pair count of every slice, the rest of active ones)
      const int nc0 = (int) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0), 13));
      const int nc1 = (int) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v1), 13));
      const bool act0 = nc0 > min_corr0, act1 = two && nc1 > min_corr1;
      const bool always = lane == 13;
      float tot = 0.0f;
      tot += (act0 || always) ? v0 : 0.0f;        // (adding +0 to a sum that started from +0 changes nothing)
      tot += (act1 || always) ? v1 : 0.0f;
      last_n_in = (int) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 11));
      if (out_stats && lane >= 9 && lane < 14) {        // StatsDev {n_corr, n_in, n_out, chi_in, chi_out} <- lanes 13, 11, 12, 9, 10
        const int word = lane == 13 ? 0 : (lane == 11 ? 1 : (lane == 12 ? 2 : lane - 6));
        reinterpret_cast<int32_t*>(out_stats + it)[word] = lane < 11 ? __float_as_int(tot) : (int) tot;
      }
      if (out_stats && lane == 0) {      // the iteration's pair digest (every pair's add landed before the barrier above); zeroed for the next iteration
        const u64 dg = s_dig; s_dig = 0ull;
        reinterpret_cast<uint32_t*>(out_stats + it)[5] = (uint32_t) dg; reinterpret_cast<uint32_t*>(out_stats + it)[6] = (uint32_t) (dg >> 32);
      }
      LSM2D_PC(8);               // sums of the slices, statistics
      bool done_now = false;
      if (!(act0 || act1)) { status = LSM2D_NOT_ENOUGH_CORRESPONDENCES; done_now = true; }
      else {
        const float Hq = (has_prior && has_pterm) ? tot + P : tot;
        hl = Hq;
#define LSM2D_RL_F(k) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Hq), k))
        float dx, dy, dth;
        const bool ok = solve_flat(LSM2D_RL_F(0), LSM2D_RL_F(1), LSM2D_RL_F(2), LSM2D_RL_F(3), LSM2D_RL_F(4), LSM2D_RL_F(5),
                                   LSM2D_RL_F(6), LSM2D_RL_F(7), LSM2D_RL_F(8), damping, dx, dy, dth);
#undef LSM2D_RL_F
        if (!ok) { status = LSM2D_SINGULAR_H; done_now = true; }
        else {
          const float nx = __builtin_fmaf(cp, dx, __builtin_fmaf(-sp, dy, pose[0]));
          const float ny = __builtin_fmaf(sp, dx, __builtin_fmaf(cp, dy, pose[1]));
          pose[0] = nx; pose[1] = ny; pose[2] = wrap_angle(pose[2] + dth);
          bool phase_over = it + 1 >= phase_end;
          if (term_eps > 0.0f) {         // as in k_align
            const float chi_now = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 9)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 10));
            if (it > phase_start && __builtin_fabsf(prev_chi - chi_now) < term_eps * chi_now) phase_over = true;
            prev_chi = chi_now;
          }
          if (phase_over) {              // as in k_align: the inlier-only runs follow a regular loop that ended well
            if (A.inlier_runs && phase == 0 && last_n_in >= A.min_inliers) { phase = 1; phase_start = it + 1; phase_end = it + 1 + A.max_it; if (lane == 0) s_inl = 1; }
            else done_now = true;
          }
        }
      }
      LSM2D_PC(10);              // 3x3 solve and pose update
      if (done_now) { if (lane == 0) s_done = 1; }
      else {
        if (lane < A.n_slices) s_iso[lane] = slice_iso_of(khs, kc, ks, kS, pose);      // the next iteration's transforms: one slice per lane
        if (A.out_last_pose && lane < 3) A.out_last_pose[3 * a + lane] = lane == 0 ? pose[0] : (lane == 1 ? pose[1] : pose[2]);
      }
      LSM2D_PC(4);
    }
    __syncthreads();
    LSM2D_PC(5);
    if (s_done) { ++it; break; }
  }
#ifdef LSM2D_PHASE_CLOCKS
  if (gtid == 0 && a == 0) printf("k_align_pair ticks(10ns): prologue %llu project %llu walk %llu wave-sums %llu wait %llu gather %llu solve %llu barrier %llu its %d\n",
                                  pc_acc[0], pc_acc[1], pc_acc[6], pc_acc[7], pc_acc[2], pc_acc[3], pc_acc[4] + pc_acc[8] + pc_acc[9] + pc_acc[10], pc_acc[5], it);
  if (gtid == 0 && a == 0) printf("  solve = sums %llu + prior %llu + ldlt/update %llu + next transforms %llu\n", pc_acc[8], pc_acc[9], pc_acc[10], pc_acc[4]);
#endif
#undef LSM2D_PC
  if (w0) {
    int st = status;
    if (st == LSM2D_RUNNING) st = (A.max_it > 0 && last_n_in < A.min_inliers) ? LSM2D_NOT_ENOUGH_INLIERS : LSM2D_SUCCESS;
    if (lane < 3) A.out_pose[3 * a + lane] = lane == 0 ? pose[0] : (lane == 1 ? pose[1] : pose[2]);
    if (A.out_H && lane >= 32 && lane < 41) A.out_H[9 * a + lane - 32] = hl;
    if (A.out_its && lane == 0) A.out_its[a] = it;
    if (A.host_polls) {      // status last (see k_align); the fence covers every lane's stores above
      __threadfence_system();
      if (lane == 0) __hip_atomic_store(&A.out_status[a], st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    else if (lane == 0) A.out_status[a] = st;
  }
}
