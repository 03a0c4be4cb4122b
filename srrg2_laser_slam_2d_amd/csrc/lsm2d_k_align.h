// lsm2d_k_align.h -- k_align: one workgroup owns one alignment for all its iterations -- the moving cloud streamed through the LDS z-buffer, the bin walk, the point-query passes, the reduction, the 3x3 solve (MultiAligner2D::compute; registration/correspondence_finder_projective_2d.cpp:35-74; octave/solver/nicp_post.m:69-97); k_align_seq: the same with the reference's order of summation.
// Part of lsm2d_kernels.h (included there, inside namespace lsm2d, in this order); not a translation unit of its own.
// kSeq: "sum_order" 1 -- H, b and the chi^2 statistics are added pair after pair in the reference's order (lsm2d_device.h: pair_terms / seq_walk) instead of
// in trees; instantiations of their own (k_align_seq), so the default kernels do not carry the records' code
// kW: threads of the workgroup (round 6).  512 = kAlignBlock everywhere but in the NARROW instantiation of the culled projective stream (k_align_narrow<256>):
// four waves per workgroup, one per SIMD, so six workgroups fit a CU (the canvases' LDS bounds it: 1536 resident alignments instead of 1024) -- for batches just
// above a multiple of 1024, whose last alignments would otherwise run a second, nearly empty dispatch round (tools/batch_size_sweep.py: 1100 .. 1536 alignments
// lost 17-25 % per alignment).  The z-buffer does not care who streams which point; the bin walk and the sums keep the 512 VIRTUAL threads of the wide kernel --
// thread t plays virtual threads t, t + kW, ... one after the other, each with its own partial sums, its waves' trees written to the same eight rows of `red` --
// so every result keeps its bits (tests: 256 against 512).
template <bool kHasProj, bool kHasNN, bool kHasDist, bool kHasKd = false, int kNNMode = 0, bool kFirstStage = false, bool kSeq = false, int kW = kAlignBlock>
LSM2D_DEV void align_body(const AlignArgs& A, int a_given = -1) {
  static_assert(kW == kAlignBlock || (kNNMode == 5 && !kFirstStage && !kSeq && kW % 64 == 0 && kW >= 256 && kW < kAlignBlock), "narrow workgroups: the culled projective stream only");
  __builtin_assume(A.n_slices >= 1 && A.n_slices <= kMaxSlices);      // (the host refuses anything else: the slice loops need no guard -- which, as a flag, was kept in a vector register and spilled)
  constexpr bool kNNGlobal = kNNMode == 1, kNNLds = kNNMode == 2;
  // the same for a pure KD-tree batch: 3 = every alignment's whole tree, leaf arrays included, is in LDS (the tracker's wiring: a tree per scan); 4 = only the
  // top of the tree is (the map is the fixed cloud): the other form of the descent and of the leaf scan is compiled out
  constexpr bool kKdAllLds = kNNMode == 3, kKdTop = kNNMode == 4;
  // ... and for a pure projective batch: 5 = every slice streams a lane-chunked moving cloud through the exact culling in units (what a batch against a map
  // does): the plain lane stream, the row-major variant and the per-pair stream of small clouds are compiled out
  // 6 = 5 with the XCD window (AlignArgs::xcd_sync: big maps, one dispatch round): an instantiation of its own, so that the headline's loop does not carry the
  // window's state (in one body: 16 bytes of scratch in the kernel that had none)
  constexpr bool kProjCulled = kNNMode == 5 || kNNMode == 6, kXcdWindow = kNNMode == 6;
  static_assert(kNNMode == 0 || ((kNNMode <= 2) && kHasNN && !kHasProj && !kHasDist && !kHasKd) || ((kNNMode == 3 || kNNMode == 4) && kHasKd && !kHasProj && !kHasDist && !kHasNN) ||
                ((kNNMode == 5 || kNNMode == 6) && kHasProj && !kHasNN && !kHasDist && !kHasKd), "kNNMode: one finder only");
  extern __shared__ __align__(16) unsigned char smem[];
  // (round 4: the fixed winners' payload no longer sits in LDS -- 16 bytes per column, 17 KB at 1081 -- the bin walk gathers it like the moving winner's,
  // one 16-byte row of the cloud's AoS copy each, both in flight together; the room holds the culled stream's unit lists)
  u64* mcan = reinterpret_cast<u64*>(smem);
  u64* fcan = mcan + A.cols_max;
  float* red = reinterpret_cast<float*>(fcan + A.fcan_total + ((A.cols_max + A.fcan_total) & 1));     // [nwaves][kAccumWords], 16-byte aligned
  // NN finder over a scan-sized fixed cloud (the tracker wiring: tree over the scan, every map point a query): the cloud's search
  // tables live in LDS for the whole alignment -- 20 iterations x N_m queries then touch global memory only for the query stream
  float2* l_sxy = reinterpret_cast<float2*>(red + (kAlignBlock / 64) * kAccumWords);
  int* l_qc = reinterpret_cast<int*>(red + (kAlignBlock / 64) * kAccumWords);      // kNNGlobal: [nn_qcache][8] cached cell ranges per query (16-byte aligned: the host pads)
  uint16_t* l_cst = reinterpret_cast<uint16_t*>(l_sxy + A.nn_lds_points);
  uint16_t* l_sidx = l_cst + ((A.nn_lds_cells + 2) & ~1);
  // KD-tree finder: the top levels of the fixed cloud's tree (its first kd_lds_nodes nodes) live in LDS for the whole alignment -- every
  // descent starts there (the host offers this only to pure KD-tree batches, where the region behind `red` is 16-byte aligned and free)
  float4* l_kpl = reinterpret_cast<float4*>(red + (kAlignBlock / 64) * kAccumWords);
  int2* l_klk = reinterpret_cast<int2*>(l_kpl + A.kd_lds_nodes);
  float2* l_kxy = reinterpret_cast<float2*>(l_klk + A.kd_lds_nodes + (A.kd_lds_nodes & 1));      // 16-byte aligned: pairs of points are read as one
  float2* l_knr = l_kxy + A.kd_lds_points + (A.kd_lds_points & 1);
  __shared__ float s_pose[3];
  __shared__ Iso   s_iso[kMaxSlices];
  // s_H: information matrix (H of the last solved iteration, built and solved IN LDS: thread 0's serial code has 64 registers like
  // everybody else, and what it kept in private arrays went to scratch -- eleven dependent round trips to memory per iteration);
  // s_sum: this iteration's sums in the order of Accum (h00 h01 h02 h11 h12 h22 b0 b1 b2 chi_in chi_out | n_in n_out as integers),
  // each added by the lane of wave 0 that gathered it
  __shared__ float s_H[9], s_rhs[3], s_sum[kAccumWords + 2];
  __shared__ int   s_n_corr, s_active, s_done, s_status, s_last_n_in;
  __shared__ float s_prev_chi;      // total chi^2 of the previous iteration (termination_chi_epsilon)
  __shared__ u64 s_dig;             // this iteration's pair digest (lsm2d_iteration_stats.pair_digest): every matched pair adds its hash; only when statistics go out
  __shared__ int s_it0;            // the iteration this launch starts at (0, or where the first of two launches stopped)
  __shared__ int s_phase, s_phase_start, s_phase_end;      // 0: the regular loop, 1: the inlier-only runs (enable_inlier_only_runs); iterations [start, end) belong to the phase
  __shared__ uint16_t s_surv[kAlignBlock];      // culling: the chunks of the moving cloud that survived this iteration's test, compacted in thread order
  __shared__ int s_wcnt[2 * (kAlignBlock / 64)];
  __shared__ int s_nunits[kMaxSlices], s_rebuild[kMaxSlices];      // kProjCulled: entries in a slice's unit list; the list must be rebuilt before it is streamed again
  __shared__ Iso s_list_iso[kMaxSlices];                             // ... and the transform it was built at
  uint16_t* l_units = reinterpret_cast<uint16_t*>(smem + A.units_off);      // [n_slices][kCullBlocks * kAlignBlock]
  float* l_rec = reinterpret_cast<float*>(smem + (kSeq ? A.seq_off : 0));   // kSeq: [kSeqHalf][kSeqFields]: half a trip's pair records
  __shared__ PriorDev s_prior;      // read once: with zero-copy arguments A.prior is host memory, a PCIe round trip per access

  // (the alignment's index is wave-uniform: said so, or everything indexed by it would live in vector registers)
  const int a = a_given >= 0 ? a_given : (A.order ? __builtin_amdgcn_readfirstlane(A.order[blockIdx.x]) : (int) blockIdx.x), tid = threadIdx.x;
  constexpr int nwaves = kW / 64;      // physical waves (the rows of `red` stay kAlignBlock / 64: one per VIRTUAL wave)
  constexpr int kPriorWords = (int) (sizeof(PriorDev) / sizeof(float));
  __shared__ unsigned long long s_clk[2];      // start stamps wait in LDS: no register is held across the kernel for them
#ifdef LSM2D_PHASE_PROBE      // diagnostics build: thread 0 sums the cycles it spends in the query / projection phase, at the barrier + reduction, and in the solve
  __shared__ unsigned long long s_ph[4];
  if (tid == 0) { s_ph[0] = s_ph[1] = s_ph[2] = 0; s_ph[3] = __builtin_amdgcn_s_memtime(); }
#define LSM2D_PH_STAMP(k) do { if (tid == 0) { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); s_ph[k] += now__ - s_ph[3]; s_ph[3] = now__; } } while (0)
  // -DLSM2D_PHASE_PROBE=1 (or empty): the three buckets above.  =2: unit lists + stream | bin walk + reduction | solve + the rest.  =3: unit lists | stream | everything else.
#if LSM2D_PHASE_PROBE + 0 == 2
#define LSM2D_PH(k) LSM2D_PH_STAMP((k) == 0 ? 1 : (k))
#define LSM2D_PH_LISTS() do { } while (0)
#define LSM2D_PH_STREAM() LSM2D_PH_STAMP(0)
#elif LSM2D_PHASE_PROBE + 0 == 3
#define LSM2D_PH(k) LSM2D_PH_STAMP(2)
#define LSM2D_PH_LISTS() LSM2D_PH_STAMP(0)
#define LSM2D_PH_STREAM() LSM2D_PH_STAMP(1)
#else
#define LSM2D_PH(k) LSM2D_PH_STAMP(k)
#define LSM2D_PH_LISTS() do { } while (0)
#define LSM2D_PH_STREAM() do { } while (0)
#endif
#else
#define LSM2D_PH(k) do { } while (0)
#define LSM2D_PH_LISTS() do { } while (0)
#define LSM2D_PH_STREAM() do { } while (0)
#endif
  if (A.wg_place && tid == 0) A.wg_place[blockIdx.x] = place_key();
  // the XCD lockstep (AlignArgs::xcd_sync): this workgroup's counters are its XCD's; it counts itself in before anything else
  uint32_t* xsync = nullptr;      // (wave-uniform: SGPRs)
  if (kXcdWindow && !kFirstStage && A.xcd_sync) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) & 15u;      // HW_REG_XCC_ID
    xsync = A.xcd_sync + (size_t) xcc * A.xcd_stride;
    if (tid == 0) __hip_atomic_fetch_add(&xsync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  const bool stamp = A.clock_out && tid == 0 && a % A.clock_stride == 0;
  if (stamp) { s_clk[0] = __builtin_amdgcn_s_memtime(); s_clk[1] = __builtin_amdgcn_s_memrealtime(); }
  if (A.prior && tid >= 64 && tid < 64 + kPriorWords)
    ((float*) &s_prior)[tid - 64] = A.inline_n1 ? ((const float*) &A.prior1)[tid - 64] : ((const float*) (A.prior + a))[tid - 64];

  // ---- point-query finders, tracker wiring (a scan-sized fixed cloud, every point of a big moving cloud a query): EXACT culling of the queries.
  // A pair needs a fixed point within max_distance of the transformed moving point (the grid search's gate d2 <= md2, the tree's d2 < md2, the
  // distance map's parent pixel within max_distance of the query's pixel), so a TILE of 64 consecutive moving points -- what one wave handles in
  // one trip of the query loop -- can be skipped when no fixed point lies within rho + max_distance of its bounding circle's centre
  // (k_tile_bounds: centre, rho).  The fixed cloud is rasterised ONCE per alignment into a 128 x 128 occupancy bitmap (cell side g: its extent
  // / 125, at least a sixth of the reach); the test looks at the (2k + 1)^2 cells around the centre's, k = floor(reach / g) + 1: if they are
  // clear, every fixed point is more than `reach` away.  The queries keep their threads and a skipped query could not have paired: every sum
  // keeps its bits.  (Measured on configs[1], role A: 49 % of the tiles survive where an exact distance test would keep 32 %.)
  constexpr int kPqRowWords = 5, kPqOccWords = 128 * kPqRowWords;      // rows of 128 bits and one word that stays zero: a row's window is read as two words
  __shared__ unsigned s_pqbb[4];      // the fixed cloud's bounding box as order-preserving integers: min x, min y, max x, max y
  __shared__ float s_pq[4];           // bitmap origin x, y, 1 / g, the reach beyond a tile's own radius (< 0: culling off for this alignment)
  const bool pq_on = (kHasNN || kHasKd || kHasDist) && A.pq_cull_off > 0;
  uint32_t* l_occ = reinterpret_cast<uint32_t*>(smem + (pq_on ? A.pq_cull_off : 0));
  u64* l_keep = reinterpret_cast<u64*>(l_occ + kPqOccWords);
  auto ordered = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
  auto unordered = [](unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); };
  if (pq_on) {
    for (int i = tid; i < kPqOccWords; i += kAlignBlock) l_occ[i] = 0u;
    if (tid < 4) s_pqbb[tid] = tid < 2 ? 0xFFFFFFFFu : 0u;
  }

  if (kHasProj && !kHasNN && !kHasDist && !kHasKd && A.inline_n1)
    for (int s = 0; s < A.n_slices; ++s) if (A.s[s].unpack_src) unpack_fixed_set(A.s[s], tid, kW);      // visible after the barrier below
  // ---- prologue: fixed canvases, camera at identity (correspondence_finder_projective_2d.cpp:37-44)
  for (int i = tid; i < A.fcan_total; i += kW) fcan[i] = kEmptyCell;
  for (int i = tid; i < A.cols_max; i += kW) mcan[i] = kEmptyCell;      // afterwards the bin walk resets what it reads
  // per-iteration set-up by lane 0: X_eff = S^-1 X per slice (AlignerSliceProcessorLaser2DWithSensor), cos/sin once per
  // slice, zeroed sums.  Done here for iteration 0 and at the end of every solve for the next one (no extra barrier).
  auto begin_iteration = [&]() {
    for (int s = 0; s < A.n_slices; ++s) {
      float Xe[3] = {s_pose[0], s_pose[1], s_pose[2]};
      if (A.s[s].has_sensor) compose(A.s[s].cSinv, A.s[s].sSinv, A.s[s].Sinv, s_pose, Xe);
      sincos_fixed(Xe[2], s_iso[s].s, s_iso[s].c); s_iso[s].tx = Xe[0]; s_iso[s].ty = Xe[1];
    }
    // (the zero is made HERE, every time: a constant the compiler may hoist becomes a zero quad that every thread keeps -- and spills --
    // across the whole kernel for thread 0's sake)
    float zf = 0.0f; int zi = 0;
    asm volatile("" : "+v"(zf), "+v"(zi));
    for (int k = 0; k < 11; ++k) s_sum[k] = zf;
    s_sum[11] = s_sum[12] = __int_as_float(zi);
    s_n_corr = s_active = zi;
    s_dig = (u64) (unsigned) zi;
    if (kProjCulled) for (int s = 0; s < A.n_slices; ++s) {
      // has the slice's transform left the neighbourhood its unit list serves?  Seen from the sensor the change is a rotation by dth about the origin
      // and a translation d = t - R(dth) t0 (chunk_may_matter): the list holds while |d| <= cull_mt and |dth| <= cull_mth
      const Iso N = s_iso[s], L = s_list_iso[s];
      const float cd = N.c * L.c + N.s * L.s, sd = N.s * L.c - N.c * L.s;
      const float dx = N.tx - (cd * L.tx - sd * L.ty), dy = N.ty - (sd * L.tx + cd * L.ty);
      s_rebuild[s] = (A.cull_keep && cd > 0.5f && __builtin_fabsf(sd) <= A.cull_mth && dx * dx + dy * dy <= A.cull_mt2) ? zi : 1;      // (the square comes with the arguments: formed here it was a
      // loop invariant in a vector register, kept -- and spilled -- across the whole iteration for thread 0's sake)
    }
    if (A.out_last_pose) { A.out_last_pose[3 * a + 0] = s_pose[0]; A.out_last_pose[3 * a + 1] = s_pose[1]; A.out_last_pose[3 * a + 2] = s_pose[2]; }
  };
  const bool resumed = kExperiments && kProjCulled && !kFirstStage && A.stage == 2;      // the second of two launches: the alignment goes on where k_first_iteration left it
  if (tid == 0) {
    if (resumed) {
      const ResumeDev R = A.resume[a];
      s_pose[0] = R.pose[0]; s_pose[1] = R.pose[1]; s_pose[2] = R.pose[2];
      s_done = R.done; s_status = R.status; s_last_n_in = R.last_n_in; s_prev_chi = R.prev_chi;
      s_phase = R.phase; s_phase_start = R.phase_start; s_phase_end = R.phase_end; s_it0 = R.it;
      for (int k = 0; k < 9; ++k) s_H[k] = R.H[k];
    } else {
      if (A.inline_n1) { s_pose[0] = A.pose1[0]; s_pose[1] = A.pose1[1]; s_pose[2] = A.pose1[2]; }
      else { s_pose[0] = A.init_pose[3 * a + 0]; s_pose[1] = A.init_pose[3 * a + 1]; s_pose[2] = A.init_pose[3 * a + 2]; }
      s_done = 0; s_status = LSM2D_RUNNING; s_last_n_in = 0;
      s_phase = 0; s_phase_start = 0; s_phase_end = A.max_it; s_it0 = 0;
      for (int k = 0; k < 9; ++k) s_H[k] = 0.0f;
    }
    for (int s = 0; s < kMaxSlices; ++s) { s_list_iso[s].c = 1.0f; s_list_iso[s].s = 0.0f; s_list_iso[s].tx = 0.0f; s_list_iso[s].ty = 0.0f; }
    if (!s_done) begin_iteration();      // (resumed: the transforms and the zeroed sums the first launch's last begin_iteration() made, made again from the same pose)
    for (int s = 0; s < kMaxSlices; ++s) s_rebuild[s] = 1;      // no list yet
  }
  __syncthreads();
  if (resumed && s_done) return;         // it finished in the first launch: its results are out
  if (kNNGlobal) for (int i = tid; i < A.nn_qcache; i += kAlignBlock) l_qc[8 * i] = 0x7fffffff;      // no cell cached yet (visible after the barriers below)
  bool nn_lds = false;
  if (kHasNN && !kNNGlobal && A.nn_lds_points > 0) {
    const SliceDev& S = A.s[0];
    const int fc = pick_cloud(S.fixed, a), nf = S.fixed.count[fc];
    const GridMeta g0 = S.fixed.grid.meta[fc];
    const int ncell = g0.gw * g0.gh;
    nn_lds = nf <= A.nn_lds_points && ncell + 1 <= A.nn_lds_cells;     // workgroup-uniform; else this alignment searches in global memory
    if (nn_lds) {
      const int32_t* cst = S.fixed.grid.cell_start + g0.cell_base;
      const int fbase = S.fixed.start[fc];
      for (int i = tid; i <= ncell; i += kAlignBlock) l_cst[i] = (uint16_t) cst[i];
      for (int i = tid; i < nf; i += kAlignBlock) { l_sxy[i] = S.fixed.grid.sorted_xy[fbase + i]; l_sidx[i] = (uint16_t) S.fixed.grid.sorted_idx[fbase + i]; }
    }
  }
  if (kNNLds && !nn_lds) {      // cannot happen (the host sized the staging for the set's largest cloud): refuse loudly rather than search tables that are not there
    if (tid == 0) {
      A.out_pose[3 * a + 0] = s_pose[0]; A.out_pose[3 * a + 1] = s_pose[1]; A.out_pose[3 * a + 2] = s_pose[2];
      if (A.out_H) for (int k = 0; k < 9; ++k) A.out_H[9 * a + k] = 0.0f;      // (no iteration ran: the information matrix is the zero the regular path would hand back)
      if (A.out_its) A.out_its[a] = 0;
      if (A.host_polls) { __threadfence_system(); __hip_atomic_store(&A.out_status[a], (int) LSM2D_CAPACITY_EXCEEDED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
      else A.out_status[a] = LSM2D_CAPACITY_EXCEEDED;
    }
    return;
  }
  int kd_lds = 0;                            // nodes of this alignment's tree that were staged (workgroup-uniform)
  bool kd_leaves_lds = false;                // ... and its leaf arrays
  if (kHasKd && A.kd_lds_nodes > 0) {
    const SliceDev& S = A.s[0];
    const KdMeta km = S.fixed.kd.meta[pick_cloud(S.fixed, a)];
    kd_lds = km.n_nodes < A.kd_lds_nodes ? km.n_nodes : A.kd_lds_nodes;
    const KdNode* nd = S.fixed.kd.nodes + km.node_base;
    for (int i = tid; i < kd_lds; i += kAlignBlock) {
      l_kpl[i] = reinterpret_cast<const float4*>(nd)[2 * i];
      const int4 w = reinterpret_cast<const int4*>(nd)[2 * i + 1]; l_klk[i] = make_int2(w.x, w.y);
    }
    // a scan-sized fixed cloud (the tracker wiring: a tree per scan, every map point a query): its leaf arrays ride in LDS too -- the whole
    // tree is on chip and 20 iterations x N_m queries touch global memory for the query stream only
    if (A.kd_lds_points > 0) {
      const int fc = pick_cloud(S.fixed, a), nf = S.fixed.count[fc], fb = S.fixed.start[fc];
      kd_leaves_lds = !kKdTop && nf <= A.kd_lds_points && kd_lds == km.n_nodes;      // workgroup-uniform
      if (kd_leaves_lds) for (int i = tid; i < nf; i += kAlignBlock) { l_kxy[i] = S.fixed.kd.leaf_xy[fb + i]; l_knr[i] = S.fixed.kd.leaf_nrm[fb + i]; }
    }
  }
  if (pq_on) {      // bounding box of the fixed cloud (finite points only)
    const SliceDev& S = A.s[0];
    const int fc = pick_cloud(S.fixed, a), nf = S.fixed.count[fc];
    const float2* fp = S.fixed.xy + S.fixed.start[fc];
    for (int i = tid; i < nf; i += kAlignBlock) {
      const float2 p = fp[i];
      if (__builtin_fabsf(p.x) < 1e30f && __builtin_fabsf(p.y) < 1e30f) {
        atomicMin(&s_pqbb[0], ordered(p.x)); atomicMin(&s_pqbb[1], ordered(p.y)); atomicMax(&s_pqbb[2], ordered(p.x)); atomicMax(&s_pqbb[3], ordered(p.y));
      }
    }
  }
  if (kKdAllLds && !kd_leaves_lds) {      // cannot happen (the host sized the staging for the set's largest tree): refuse loudly, as above
    if (tid == 0) {
      A.out_pose[3 * a + 0] = s_pose[0]; A.out_pose[3 * a + 1] = s_pose[1]; A.out_pose[3 * a + 2] = s_pose[2];
      if (A.out_H) for (int k = 0; k < 9; ++k) A.out_H[9 * a + k] = 0.0f;      // (no iteration ran: the information matrix is the zero the regular path would hand back)
      if (A.out_its) A.out_its[a] = 0;
      if (A.host_polls) { __threadfence_system(); __hip_atomic_store(&A.out_status[a], (int) LSM2D_CAPACITY_EXCEEDED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
      else A.out_status[a] = LSM2D_CAPACITY_EXCEEDED;
    }
    return;
  }
  const Iso ident = {1.0f, 0.0f, 0.0f, 0.0f};
  for (int s = 0; s < A.n_slices; ++s) {
    const SliceDev& S = A.s[s];
    if (!kHasProj || S.finder != LSM2D_FINDER_PROJECTIVE) continue;
    const int fc = pick_cloud(S.fixed, a);
    // (a set unpacked by this launch: its size comes with the arguments -- the scalar cache may not have seen the count written above)
    project_cloud(S.fixed.xy + S.fixed.start[fc], (A.inline_n1 && S.unpack_src) ? S.unpack_n : S.fixed.count[fc], ident, S.proj, fcan + S.fcan_offset, tid, kW);
  }
  __syncthreads();
  if (pq_on) {      // the occupancy bitmap: every thread derives the same cell size and origin from the box, then stamps its points
    const SliceDev& S = A.s[0];
    const int fc = pick_cloud(S.fixed, a), nf = S.fixed.count[fc];
    const float2* fp = S.fixed.xy + S.fixed.start[fc];
    const unsigned k0 = s_pqbb[0], k1 = s_pqbb[1], k2 = s_pqbb[2], k3 = s_pqbb[3];
    const float minx = unordered(k0), miny = unordered(k1), maxx = unordered(k2), maxy = unordered(k3);
    const bool have = k0 <= k2 && k1 <= k3;                           // at least one finite point
    // a skipped tile's points stay farther than max_distance from every fixed point; the distance map pairs a query with the point of a PIXEL whose
    // centre is within max_distance of the query's pixel centre: two pixel diagonals more
    float reach = S.max_distance * 1.002f + 2e-3f;
    if (kHasDist && S.finder == LSM2D_FINDER_DISTMAP) reach += 3.0f / S.fixed.dist.meta[fc].inv_res;
    const float ext = __builtin_fmaxf(maxx - minx, maxy - miny);
    const float g = __builtin_fmaxf((reach + 0.1f) * (1.0f / 6.0f), ext * (1.0f / 125.0f));
    const float ox = minx - g, oy = miny - g, inv_g = 1.0f / g;
    const bool usable = have && g > 0.0f && g < 1e30f && S.max_distance >= 0.0f;
    if (tid == 0) { s_pq[0] = ox; s_pq[1] = oy; s_pq[2] = inv_g; s_pq[3] = usable ? reach : -1.0f; }
    if (usable) for (int i = tid; i < nf; i += kAlignBlock) {
      const float2 p = fp[i];
      if (!(__builtin_fabsf(p.x) < 1e30f && __builtin_fabsf(p.y) < 1e30f)) continue;
      const int cx = (int) ((p.x - ox) * inv_g), cy = (int) ((p.y - oy) * inv_g);       // 1 .. 126 by construction
      if ((unsigned) cx < 128u && (unsigned) cy < 128u) atomicOr(&l_occ[cy * kPqRowWords + (cx >> 5)], 1u << (cx & 31));
    }
  }
  __syncthreads();

  int it = __builtin_amdgcn_readfirstlane(s_it0);
  const int it_cap = __builtin_amdgcn_readfirstlane(A.inlier_runs ? 2 * A.max_it : A.max_it);      // (a scalar: as a select and a shift it lived in a vector register, spilled for the loop's back edge)
  const bool want_dig = A.out_stats != nullptr;      // the digest leaves the kernel through the statistics only
  for (; it < it_cap; ++it) {
    const bool lists_only = kFirstStage && it == A.stage_split;      // the first of two launches enters this iteration for the LENGTH of its unit lists alone
    const bool inl_only = A.inlier_runs && __builtin_amdgcn_readfirstlane(s_phase) != 0;
#if LSM2D_PRIO_BY_PROGRESS == 1
    { const int q = (4 * it) / A.max_it; if (q == 0) __builtin_amdgcn_s_setprio(3); else if (q == 1) __builtin_amdgcn_s_setprio(2); else if (q == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#elif LSM2D_PRIO_BY_PROGRESS == 2
    { if (2 * it < A.max_it) __builtin_amdgcn_s_setprio(3); else if (4 * it < 3 * A.max_it) __builtin_amdgcn_s_setprio(2); else if (8 * it < 7 * A.max_it) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
    for (int s = 0; s < A.n_slices; ++s) {
      const SliceDev& S = A.s[s];
      const Iso T = s_iso[s];
      const uint32_t salt = (uint32_t) s * 0x632BE5ABu;
      Accum acc; accum_zero(acc);
      float seq_acc = 0.0f;      // kSeq: wave 0's lanes carry the slice's eleven running sums (in Accum's order) from trip to trip: seq_walk / seq_total; the counts stay in acc
      // kSeq: a matched pair becomes a record instead of being added into the thread's partial sums
      bool seq_have = false;      // kSeq: this thread holds a pair of the current trip
      auto seq_pair = [&](float2 pf, float2 nf, float2 pm, float2 nm, float (&t)[kSeqFields]) {
        bool inl; pair_terms(T, pf, nf, pm, nm, S.cauchy != 0, S.tau, inl_only, t, inl);
        ++acc.n_corr; acc.n_in += inl ? 1 : 0; acc.n_out += inl ? 0 : 1;
        seq_have = true;
      };
      // kSeq: the end of a trip.  The trip's PAIRS are compacted in thread order -- ascending column for the projective walk, ascending moving index for the queries: the
      // reference's order (a slot without a pair would add fma(+0, +0, h) == h: leaving it out changes no bit, and on the tracker's wiring of the point-query finders a
      // trip of 512 queries holds a handful of pairs) -- and APPENDED to the records in LDS (kSeqHalf of them: 14 KB); wave 0 walks them, in ascending position, whenever
      // the buffer is full and once more at the end of the slice (seq_flush): a walk costs two barriers, and the queries' 196 trips per iteration on configs[1] fill
      // the buffer three or four times.  One barrier per trip (the compaction's); the stores of later trips go to other slots than those of earlier ones, and every walk
      // stands between two barriers of its own.
      int seq_fill = 0;      // records waiting in LDS (the same in every thread)
      auto seq_walk_buffer = [&](int n) {      // n records in LDS, "no pair" records up to the next multiple of eight (seq_walk takes eight at a time)
        if (tid < ((n + 7) & ~7) - n) { float z[kSeqFields]; seq_zero(z); seq_store(l_rec, n + tid, z); }
        __syncthreads();
        // (LSM2D_SEQ_WALK_PRIO: the walking wave ahead of the other workgroups' streams on its SIMD -- its workgroup's other seven waves wait for it, and for the
        // solve, which is thread 0's: the wave keeps the priority until the next iteration sets the one its progress earns)
        if (tid < 64) seq_acc = seq_walk<LSM2D_SEQ_WALK_PRIO != 0>(l_rec, n, tid, seq_acc);
        __syncthreads();
      };
      auto seq_trip = [&](const float (&t)[kSeqFields], int parity /* 0, 1, 0, ... from trip to trip: s_wcnt's two buffers */) {
        int n_rec = 0;
        const int pos = block_compact_pos(seq_have, s_wcnt, parity, n_rec, tid, kAlignBlock / 64);      // (one barrier; n_rec: the trip's pairs, the same in every thread)
        for (int done = 0; done < n_rec; ) {
          const int room = kSeqHalf - seq_fill, take = n_rec - done < room ? n_rec - done : room;
          if (seq_have && pos >= done && pos < done + take) seq_store(l_rec, seq_fill + pos - done, t);
          seq_fill += take; done += take;
          if (seq_fill == kSeqHalf) { seq_walk_buffer(kSeqHalf); seq_fill = 0; }
        }
        seq_have = false;
      };
      auto seq_flush = [&]() { if (seq_fill > 0) { seq_walk_buffer(seq_fill); seq_fill = 0; } };      // the end of the slice's pairs: what is still waiting
      LSM2D_PH(2);
      if (kHasProj && ((!kHasNN && !kHasDist && !kHasKd) || S.finder == LSM2D_FINDER_PROJECTIVE)) {
        {
          // HOT: every moving point, every iteration (correspondence_finder_projective_2d.cpp:47-48)
          const int mc = pick_cloud(S.moving, a);
          // clouds of at most one pair per thread (the tracker's clipped scenes) need no lane-chunked copy
          if (kProjCulled) {
            // Round 4: the survivors of the BLOCK-level test as a list in LDS, kept across iterations (see AlignArgs::units_off).  Build, when thread 0 found the
            // slice's transform outside the list's neighbourhood: (A) every thread tests the chunk it owns -- as the per-iteration test of round 3 did, with the
            // margins -- and the surviving chunks are compacted; (B) the nb blocks of every surviving chunk are tested the same way, dealt to the threads in
            // block-major order and compacted in that order: the list.  2 + ceil(nb s / 512) + 1 barriers, a few times per alignment.
            const int lane = tid & 63, wave = tid >> 6;
            uint16_t* units = l_units + s * A.units_stride;
            const int Tm = S.moving.lane_T[mc];
            const int nbs = __builtin_amdgcn_readfirstlane(S.moving.block_stride);
            const int B = cull_block_steps(Tm, nbs), nb = (Tm + B - 1) / B;
            if (__builtin_amdgcn_readfirstlane(s_rebuild[s]) || lists_only) {      // (lists_only: the list the second launch will build first, whatever the kept one covers)
              typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
              const float m_t = A.cull_mt, m_th = A.cull_mth;      // (0 when the lists are not kept: the host sees to that -- a select made here was a vector register held across the iteration)
              const unsigned long long bb = reinterpret_cast<unsigned long long>(S.moving.lane_bounds + (size_t) mc * kAlignBlock);
              float4* bbase = reinterpret_cast<float4*>(((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (bb >> 32)) << 32) |
                                                        (unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) bb));
              int n_surv = 0;
              if constexpr (kW == kAlignBlock) {
              const u32x4 bw = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(bbase, (short) 0, kAlignBlock * 16, 0x00020000), tid * 16, 0, 0);
              const bool keep = chunk_may_matter(T, S.proj, make_float4(__uint_as_float(bw.x), __uint_as_float(bw.y), __uint_as_float(bw.z), 0.0f), fcan + S.fcan_offset, S.point_distance, m_t, m_th);
              const u64 bal = __ballot(keep);
              { int wv = tid >> 6; asm volatile("" : "+v"(wv)); if (lane == 0) s_wcnt[wv] = __popcll(bal); }      // (the address made here, not in front of the iteration loop and kept)
              __syncthreads();
              int before = 0;
#pragma unroll
              for (int w = 0; w < nwaves; ++w) { const int cw = s_wcnt[w]; before += w < wave ? cw : 0; n_surv += cw; }
              n_surv = __builtin_amdgcn_readfirstlane(n_surv);
              if (keep) s_surv[before + (int) __builtin_amdgcn_mbcnt_hi((unsigned) (bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) bal, 0u))] = (uint16_t) tid;
              __syncthreads();
              } else {      // narrow workgroup: the 512 chunks in rounds of kW, compacted round by round (any order of the survivors serves: the z-buffer's minimum is the same)
                const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(bbase, (short) 0, kAlignBlock * 16, 0x00020000);
                int par = 0;
                for (int v0 = 0; v0 < kAlignBlock; v0 += kW, par ^= 1) {
                  const int v = v0 + tid; bool keep = false;
                  if (v < kAlignBlock) {
                    const u32x4 bw = __builtin_amdgcn_raw_buffer_load_b128(brs, v * 16, 0, 0);
                    keep = chunk_may_matter(T, S.proj, make_float4(__uint_as_float(bw.x), __uint_as_float(bw.y), __uint_as_float(bw.z), 0.0f), fcan + S.fcan_offset, S.point_distance, m_t, m_th);
                  }
                  const int pos = block_compact_pos(keep, s_wcnt, par, n_surv, tid, nwaves);
                  if (keep) s_surv[pos] = (uint16_t) v;
                }
                __syncthreads();
              }
              // (B) test v = (block v / n_surv, survivor v mod n_surv), v = tid, tid + 512, ...; block_compact_pos: one barrier per round, buffers alternating
              const unsigned long long kb = reinterpret_cast<unsigned long long>(S.moving.block_bounds + (size_t) mc * nbs * kAlignBlock);
              float4* kbase = reinterpret_cast<float4*>(((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (kb >> 32)) << 32) |
                                                        (unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) kb));
              const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(kbase, (short) 0, nbs * kAlignBlock * 16, 0x00020000);
              int n_units = 0, parity = 1, i = tid, blk = 0;
              while (n_surv > 0 && i >= n_surv && blk < nb) { i -= n_surv; ++blk; }
              const int n_tests = nb * n_surv;
              for (int v0 = 0; v0 < n_tests; v0 += kW, parity ^= 1) {
                bool k2 = false; int code = 0;
                if (blk < nb) {
                  const int g = (int) s_surv[i];
                  code = (blk << 9) | g;
                  const u32x4 w4 = __builtin_amdgcn_raw_buffer_load_b128(krs, code * 16, 0, 0);      // entry (blk * 512 + g) of the cloud's block circles
                  k2 = chunk_may_matter(T, S.proj, make_float4(__uint_as_float(w4.x), __uint_as_float(w4.y), __uint_as_float(w4.z), 0.0f), fcan + S.fcan_offset, S.point_distance, m_t, m_th);
                }
                const int pos = block_compact_pos(k2, s_wcnt, parity, n_units, tid, nwaves);
                if (k2) units[pos] = (uint16_t) code;
                i += kW;
                while (n_surv > 0 && i >= n_surv && blk < nb) { i -= n_surv; ++blk; }
              }
              if (tid == 0) { s_nunits[s] = n_units; s_list_iso[s] = T; }
              __syncthreads();
            }
            if (lists_only) continue;
            LSM2D_PH_LISTS();
            const int n_units = __builtin_amdgcn_readfirstlane(s_nunits[s]);
            if (n_units > 0) project_cloud_list<kW>(S.moving.lane_xy + S.moving.lane_start[mc], Tm, T, S.proj, mcan, tid, kAlignBlock, units, n_units, B);
          }
          else if (S.moving.lane_xy && S.moving.lane_bounds && A.cull) {
            // exact culling against the fixed canvas (chunk_may_matter): every thread tests the chunk it would stream, the survivors are
            // compacted (two barriers: counts, then the list) and their points spread evenly over the workgroup (project_cloud_units)
            const int lane = tid & 63, wave = tid >> 6;
            // (the chunk's circle through a buffer resource: base in SGPRs, one 32-bit lane offset -- a per-thread 64-bit pointer would be
            // hoisted out of the iteration loop and spilled: 64 registers)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const unsigned long long bb = reinterpret_cast<unsigned long long>(S.moving.lane_bounds + (size_t) mc * kAlignBlock);
            float4* bbase = reinterpret_cast<float4*>(((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (bb >> 32)) << 32) |
                                                      (unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) bb));
            const u32x4 bw = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(bbase, (short) 0, kAlignBlock * 16, 0x00020000), tid * 16, 0, 0);
            const bool keep = chunk_may_matter(T, S.proj, make_float4(__uint_as_float(bw.x), __uint_as_float(bw.y), __uint_as_float(bw.z), 0.0f), fcan + S.fcan_offset, S.point_distance);
            const u64 bal = __ballot(keep);
            if (lane == 0) s_wcnt[wave] = __popcll(bal);
            __syncthreads();
            int before = 0, n_surv = 0;
#pragma unroll
            for (int w = 0; w < nwaves; ++w) { const int cw = s_wcnt[w]; before += w < wave ? cw : 0; n_surv += cw; }
            // (rank below the lane by v_mbcnt: a hoisted 64-bit lane mask would be two more registers held -- and spilled -- across the loops)
            if (keep) s_surv[before + (int) __builtin_amdgcn_mbcnt_hi((unsigned) (bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) bal, 0u))] = (uint16_t) tid;
            __syncthreads();
            const int Tm = S.moving.lane_T[mc];
            // blocks of an even number of steps, 7 per chunk (measured on configs[1], T = 98: blocks of 2 / 4 / 6 / 8 / 14 steps 1.098 / 1.017 /
            // 0.991 / 0.983 / 0.969 ms -- what a unit costs to set up outweighs the better balance of smaller ones; element-wise row-major
            // order, balanced to one step, 1.066: project_cloud_rows, "cull" 2)
            const int B = (kExperiments && A.cull_block > 0) ? A.cull_block : cull_block_steps(Tm), nb = (Tm + B - 1) / B;
            if (n_surv > 0) {
#ifdef LSM2D_EXPERIMENTS
              if (!kProjCulled && A.cull == 2) project_cloud_rows(S.moving.lane_xy + S.moving.lane_start[mc], Tm, T, S.proj, mcan, tid, kAlignBlock, s_surv, n_surv);
              else
#endif
              project_cloud_units(S.moving.lane_xy + S.moving.lane_start[mc], Tm, T, S.proj, mcan, tid, kAlignBlock, s_surv, n_surv, B, nb);
            }
          }
          else if (S.moving.lane_xy) project_cloud_lanes(S.moving.lane_xy + S.moving.lane_start[mc], S.moving.lane_T[mc], T, S.proj, mcan, tid, kAlignBlock);
          else project_cloud(S.moving.xy + S.moving.start[mc], S.moving.count[mc], T, S.proj, mcan, tid, kAlignBlock);
        }
        __syncthreads();
        LSM2D_PH_STREAM();
        if (kXcdWindow && xsync && tid == 0 && it * A.n_slices + s < A.xcd_positions)      // this workgroup's pass (it, s) over the map is behind all its waves
          __hip_atomic_fetch_add(&xsync[16 + it * A.n_slices + s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // bin walk (correspondence_finder_projective_2d.cpp:55-74): the fixed side comes from LDS, the two gathers of the
        // moving winner are issued together, and every cell read is reset for the next projection
        const int mbase = S.moving.start[pick_cloud(S.moving, a)], fbase = S.fixed.start[pick_cloud(S.fixed, a)];
        const float2* mn = S.moving.nrm + mbase; const float2* mp = S.moving.xy + mbase;
        const float2* fnr = S.fixed.nrm + fbase; const float2* fpp = S.fixed.xy + fbase;
        const float4* maos = S.moving.aos ? S.moving.aos + mbase : nullptr;      // (a set without its AoS copy -- sizes still pending on the device, or
        const float4* faos = S.fixed.aos ? S.fixed.aos + fbase : nullptr;        //  unpacked by this launch -- is gathered from its split arrays)
        const u64* fcs = fcan + S.fcan_offset;
#ifndef LSM2D_BINWALK_BATCHED
#define LSM2D_BINWALK_BATCHED 1
#endif
        if constexpr (kSeq) {
          // "sum_order" 1: trips of kAlignBlock consecutive columns (every thread takes part in every trip: the barriers), slot = column - first column of the trip
          for (int col0 = 0; col0 < S.proj.cols; col0 += kAlignBlock) {
            const int col = col0 + tid;
            float t[kSeqFields]; seq_zero(t);
            if (col < S.proj.cols) {
              const u64 fk = fcs[col], mk = mcan[col];
              mcan[col] = kEmptyCell;
              const float fd = __uint_as_float((uint32_t) (fk >> 32)), md = __uint_as_float((uint32_t) (mk >> 32));
              if (mk != kEmptyCell && fk != kEmptyCell && !(__builtin_fabsf(fd - md) > S.point_distance)) {
                const int mi = (int) (uint32_t) mk, fi = (int) (uint32_t) fk;
                float2 nm, pm; float4 f;
                if (maos) { const float4 m4 = maos[mi]; pm = make_float2(m4.x, m4.y); nm = make_float2(m4.z, m4.w); } else { nm = mn[mi]; pm = mp[mi]; }
                if (faos) f = faos[fi]; else { const float2 p2 = fpp[fi], n2 = fnr[fi]; f = make_float4(p2.x, p2.y, n2.x, n2.y); }
                float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
                if (!(__builtin_fmaf(nqx, f.z, nqy * f.w) < S.normal_cos)) {
                  if (want_dig) digest_add(&s_dig, salt, fi, mi);
                  seq_pair(make_float2(f.x, f.y), make_float2(f.z, f.w), pm, nm, t);
                }
              }
            }
            seq_trip(t, (col0 / kAlignBlock) & 1);
          }
        }
        else if constexpr (kW != kAlignBlock) {
          // narrow workgroup: this thread plays the VIRTUAL threads tid, tid + kW, ... (< 512) of the wide kernel one after the other -- each with its own partial sums
          // over the wide kernel's columns (vt, vt + 512, ...) in its order, each virtual wave's tree into that wave's row of `red` (kW is a multiple of 64: a
          // physical wave plays whole virtual waves)
          const int bits = 32 - __builtin_clz(((S.proj.cols + kAlignBlock - 1) / kAlignBlock) | 1);
          for (int vt = tid; vt < kAlignBlock; vt += kW) {
            Accum av; accum_zero(av);
            for (int col = vt; col < S.proj.cols; col += kAlignBlock) {
              const u64 fk = fcs[col], mk = mcan[col];
              mcan[col] = kEmptyCell;
              if (mk == kEmptyCell || fk == kEmptyCell) continue;
              const float fd = __uint_as_float((uint32_t) (fk >> 32)), md = __uint_as_float((uint32_t) (mk >> 32));
              if (__builtin_fabsf(fd - md) > S.point_distance) continue;
              const int mi = (int) (uint32_t) mk, fi = (int) (uint32_t) fk;
              float2 nm, pm; float4 f;
              if (maos) { const float4 m4 = maos[mi]; pm = make_float2(m4.x, m4.y); nm = make_float2(m4.z, m4.w); } else { nm = mn[mi]; pm = mp[mi]; }
              if (faos) f = faos[fi]; else { const float2 p2 = fpp[fi], n2 = fnr[fi]; f = make_float4(p2.x, p2.y, n2.x, n2.y); }
              float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
              if (__builtin_fmaf(nqx, f.z, nqy * f.w) < S.normal_cos) continue;
              if (want_dig) digest_add(&s_dig, salt, fi, mi);
              accumulate_pair(T, make_float2(f.x, f.y), make_float2(f.z, f.w), pm, nm, S.cauchy != 0, S.tau, av, inl_only);
            }
            block_reduce_store(av, red, vt, bits);
          }
        }
        else
        if (LSM2D_BINWALK_BATCHED && kProjCulled && maos && faos) {
          // Round 5: the walk as a two-deep pipeline -- the NEXT column's cells are read and gated (LDS) and its winners' two 16-byte rows asked for BEFORE this
          // column's factor terms are formed; the terms are added in the same column order as before (the sums keep their bits).  A trip used to end in two
          // dependent gathers from L2 that nothing covered: three exposed round trips per thread, iteration and slice; now the second and third travel under
          // the arithmetic of the one before.  (All three trips asked for up front needed 24 registers more than the kernel has: 96 bytes of scratch.)
          auto gate = [&](int col, int& fi, int& mi) -> bool {
            if (col >= S.proj.cols) return false;
            const u64 fk = fcs[col], mk = mcan[col];
            mcan[col] = kEmptyCell;
            if (mk == kEmptyCell || fk == kEmptyCell) return false;
            const float fd = __uint_as_float((uint32_t) (fk >> 32)), md = __uint_as_float((uint32_t) (mk >> 32));
            if (__builtin_fabsf(fd - md) > S.point_distance) return false;
            mi = (int) (uint32_t) mk; fi = (int) (uint32_t) fk;
            return true;
          };
          int fi_a = 0, mi_a = 0, fi_b = 0, mi_b = 0;
          float4 m_a = make_float4(0.f, 0.f, 0.f, 0.f), f_a = m_a, m_b = m_a, f_b = m_a;
          bool ok_a = gate(tid, fi_a, mi_a);
          if (ok_a) { m_a = maos[mi_a]; f_a = faos[fi_a]; }
          for (int col = tid; col < S.proj.cols; col += kAlignBlock) {
            const bool ok_b = gate(col + kAlignBlock, fi_b, mi_b);
            if (ok_b) { m_b = maos[mi_b]; f_b = faos[fi_b]; }
            if (ok_a) {
              const float2 pm = make_float2(m_a.x, m_a.y), nm = make_float2(m_a.z, m_a.w);
              float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
              if (!(__builtin_fmaf(nqx, f_a.z, nqy * f_a.w) < S.normal_cos)) {
                if (want_dig) digest_add(&s_dig, salt, fi_a, mi_a);
                accumulate_pair(T, make_float2(f_a.x, f_a.y), make_float2(f_a.z, f_a.w), pm, nm, S.cauchy != 0, S.tau, acc, inl_only);
              }
            }
            ok_a = ok_b; fi_a = fi_b; mi_a = mi_b; m_a = m_b; f_a = f_b;
          }
        }
        else
        for (int col = tid; col < S.proj.cols; col += kAlignBlock) {
          const u64 fk = fcs[col], mk = mcan[col];
          mcan[col] = kEmptyCell;
          if (mk == kEmptyCell || fk == kEmptyCell) continue;
          const float fd = __uint_as_float((uint32_t) (fk >> 32)), md = __uint_as_float((uint32_t) (mk >> 32));
          if (__builtin_fabsf(fd - md) > S.point_distance) continue;
          const int mi = (int) (uint32_t) mk, fi = (int) (uint32_t) fk;
          float2 nm, pm; float4 f;
          if (maos) { const float4 m4 = maos[mi]; pm = make_float2(m4.x, m4.y); nm = make_float2(m4.z, m4.w); } else { nm = mn[mi]; pm = mp[mi]; }
          if (faos) f = faos[fi]; else { const float2 p2 = fpp[fi], n2 = fnr[fi]; f = make_float4(p2.x, p2.y, n2.x, n2.y); }
          float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
          if (__builtin_fmaf(nqx, f.z, nqy * f.w) < S.normal_cos) continue;
          if (want_dig) digest_add(&s_dig, salt, fi, mi);
          accumulate_pair(T, make_float2(f.x, f.y), make_float2(f.z, f.w), pm, nm, S.cauchy != 0, S.tau, acc, inl_only);
        }
      } else if (kHasNN || kHasDist || kHasKd) {
        // NN finder fused with the factor (correspondence_finder_kd_tree_2d.cpp:12-27): every moving point is
        // transformed, matched to its exact nearest fixed point within max_distance, normal-gated, accumulated
        const int fc = pick_cloud(S.fixed, a), mc = pick_cloud(S.moving, a);
        // (the clouds' first indices declared wave-uniform -- they are -- so that the six pointers formed from them live in scalar registers: round 6, k_align<0,1,0,0,1>
        // 48 B of scratch -> none, role B / exact NN 1.608 -> 1.393 ms on configs[1]; the other point-query modes unchanged: profiles/r06/pq_uniform_bases_ab_r06o.txt)
        const int mbase = __builtin_amdgcn_readfirstlane(S.moving.start[mc]), fbase = __builtin_amdgcn_readfirstlane(S.fixed.start[fc]);
        const float2* fn = S.fixed.nrm + fbase; const float2* mn = S.moving.nrm + mbase;
        const float2* fp = S.fixed.xy + fbase;  const float2* mp = S.moving.xy + mbase;
        const bool use_grid = kHasNN && ((!kHasDist && !kHasKd) || S.finder == LSM2D_FINDER_NN);
        const bool use_kd = kHasKd && ((!kHasNN && !kHasDist) || S.finder == LSM2D_FINDER_KDTREE);
        GridMeta g; DistMeta dm;
        const int32_t* cst = nullptr; const int32_t* sidx = nullptr; const float2* sxy = nullptr;
        const KdNode* knd = nullptr; const float2* knr = nullptr;
        if (use_kd) {      // the reference's tree over the fixed cloud (correspondence_finder_kd_tree_2d.cpp:18-19): one descent + one leaf per query
          knd = S.fixed.kd.nodes + __builtin_amdgcn_readfirstlane(S.fixed.kd.meta[fc].node_base);
          sxy = S.fixed.kd.leaf_xy + fbase; knr = S.fixed.kd.leaf_nrm + fbase;
        } else if (use_grid) {
          g = S.fixed.grid.meta[fc];
          // the meta comes through a vector load: tell the compiler it is wave-uniform -- seven VGPRs fewer across the query loops, which
          // takes the last spills out of them (NN role B 2.27 -> 2.06 ms, role A 7.91 -> 7.82; variants_r02v_nn_scalar_meta.log)
          g.minx = uniform_f(g.minx); g.miny = uniform_f(g.miny); g.h = uniform_f(g.h); g.inv_h = uniform_f(g.inv_h);
          g.gw = __builtin_amdgcn_readfirstlane(g.gw); g.gh = __builtin_amdgcn_readfirstlane(g.gh); g.cell_base = __builtin_amdgcn_readfirstlane(g.cell_base);
          cst = S.fixed.grid.cell_start + g.cell_base;
          sidx = S.fixed.grid.sorted_idx + fbase; sxy = S.fixed.grid.sorted_xy + fbase;
          if (kNNGlobal) knr = S.fixed.grid.sorted_nrm + fbase;
        } else {
          dm = S.fixed.dist.meta[fc];       // distance-map finder: one lookup per query (correspondence_finder_nn_2d.cpp:63-80)
        }
        const float md2 = S.max_distance * S.max_distance;
        const int nm_pts = S.moving.count[mc];
        // cooperative search (kNNGroup lanes per query) when THIS alignment's fixed cloud is at least four times its moving one --
        // decided per alignment from the device-side counts, so ragged batches get the right loop for each cloud (the oracle's
        // device-order mode applies the same rule)
        const bool coop = use_grid && (long long) S.fixed.count[fc] >= 4ll * nm_pts;
        // this iteration's keep bits, one per tile of 64 moving points (see the prologue): bit t of l_keep <-> tile t
        const int n_tiles = (nm_pts + 63) >> 6;
        const bool pq_cull = pq_on && S.moving.tile_bounds && !coop && ((n_tiles + kAlignBlock - 1) / kAlignBlock) * (kAlignBlock / 64) <= A.pq_keep_words;
        if (pq_cull) {
          const float4* tb = S.moving.tile_bounds + S.moving.tile_start[mc];
          const float ox = s_pq[0], oy = s_pq[1], inv_g = s_pq[2], reach = s_pq[3];
          for (int t0 = 0; t0 < n_tiles; t0 += kAlignBlock) {
            const int t = t0 + tid; bool keep = false;
            if (t < n_tiles) {
              const float4 b = tb[t];
              keep = true;
              const float kf = (b.z * 1.002f + reach) * inv_g;      // cells the tile's reach spans (rho = +inf, a tile with a non-finite point: no claim)
              if (reach >= 0.0f && kf < 10.0f) {
                float qx, qy; xf_point(T, b.x, b.y, qx, qy);
                const float fx = __builtin_floorf((qx - ox) * inv_g), fy = __builtin_floorf((qy - oy) * inv_g);
                if (fx > -64.0f && fx < 192.0f && fy > -64.0f && fy < 192.0f) {
                  const int k = (int) kf + 1, cx = (int) fx, cy = (int) fy;
                  const int x0 = cx - k < 0 ? 0 : cx - k, x1 = cx + k > 127 ? 127 : cx + k, y0 = cy - k < 0 ? 0 : cy - k, y1 = cy + k > 127 ? 127 : cy + k;
                  keep = false;
                  if (x0 <= x1) {
                    const u64 mask = (~0ull >> (63 - (x1 - x0))) << (x0 & 31);      // at most 21 bits, from bit x0 of the two-word window
                    for (int y = y0; y <= y1; ++y) {
                      const uint32_t* row = l_occ + y * kPqRowWords + (x0 >> 5);
                      if ((((u64) row[1] << 32) | (u64) row[0]) & mask) { keep = true; break; }
                    }
                  }
                }
                else keep = !(fx == fx && fy == fy);      // far beyond the bitmap: nothing within reach; not a number: no claim
              }
            }
            const u64 bal = __ballot(keep);
            if ((tid & 63) == 0) l_keep[(t0 >> 6) + (tid >> 6)] = bal;
          }
          __syncthreads();
        }
        // the grid search is cooperative on dense fixed clouds (kNNGroup lanes per query); the distance map is one lookup
        auto query_loop = [&](auto group_tag) {
          constexpr int group = decltype(group_tag)::value;
          const int sub = tid & (group - 1);
          constexpr int per_step = kAlignBlock / group;
          for (int j0 = 0; j0 < nm_pts; j0 += per_step) {
            bool skip = false;                       // kSeq: a culled tile's wave still takes part in the trip (its records are zeros, the barriers are everybody's)
            float t[kSeqFields];                     // kSeq: this thread's record of the trip
            if constexpr (kSeq) seq_zero(t);
            if (group == 1 && pq_cull) {             // this wave's 64 queries of the trip are one tile
              const int tile = (j0 >> 6) + __builtin_amdgcn_readfirstlane(tid >> 6);
              const u64 w = l_keep[tile >> 6];
              const unsigned half = (tile & 32) ? (unsigned) (w >> 32) : (unsigned) w;
              if (!((__builtin_amdgcn_readfirstlane((int) half) >> (tile & 31)) & 1)) { if constexpr (kSeq) skip = true; else continue; }
            }
            const int j = j0 + tid / group;
            const bool live = j < nm_pts && !skip;           // whole groups are live or not: the shuffles inside stay uniform
            const float2 pm = live ? mp[j] : make_float2(0.0f, 0.0f);
            float qx, qy; xf_point(T, pm.x, pm.y, qx, qy);
            int best = -1;
            if (use_kd) {      // the match's point and normal come from the leaf arrays, where the scan found it: the original index is never needed
              if (live) {
                float2 bxy;
                const int pos = kKdAllLds ? kd_query_pos<true>(knd, l_kxy, qx, qy, md2, bxy, l_kpl, l_klk, kd_lds)
                              : kKdTop ? kd_query_pos(knd, sxy, qx, qy, md2, bxy, l_kpl, l_klk, kd_lds)
                              : kd_leaves_lds ? kd_query_pos<true>(knd, l_kxy, qx, qy, md2, bxy, l_kpl, l_klk, kd_lds) : kd_query_pos(knd, sxy, qx, qy, md2, bxy, l_kpl, l_klk, kd_lds);
                if (pos >= 0) {
                  const float2 nm = mn[j], nf = kKdAllLds ? l_knr[pos] : (kKdTop ? knr[pos] : (kd_leaves_lds ? l_knr[pos] : knr[pos]));
                  float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
                  const float dot = __builtin_fmaf(nqx, nf.x, nqy * nf.y);
                  if (!(dot < S.normal_cos)) {
                    if (want_dig) digest_add(&s_dig, salt, S.fixed.kd.leaf_idx[fbase + pos], j);      // the original index: only the digest asks for it
                    if constexpr (kSeq) seq_pair(bxy, nf, pm, nm, t); else
                    accumulate_pair(T, bxy, nf, pm, nm, S.cauchy != 0, S.tau, acc, inl_only);
                  }
                }
              }
            } else
            if (kNNGlobal) {      // the match's point and normal come from where the search found it: no detour through the original index
              const int pos = live ? nn_query_pos<group>(g, cst, sidx, sxy, qx, qy, S.max_distance, md2, sub, j < A.nn_qcache ? l_qc + 8 * j : nullptr) : -1;
              if (pos >= 0 && sub == 0) {                    // one lane per query accumulates
                const float2 nm = mn[j], nf = knr[pos], pf = sxy[pos];
                float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
                const float dot = __builtin_fmaf(nqx, nf.x, nqy * nf.y);
                if (!(dot < S.normal_cos)) {
                  if (want_dig) digest_add(&s_dig, salt, sidx[pos], j);
                  if constexpr (kSeq) seq_pair(pf, nf, pm, nm, t); else
                  accumulate_pair(T, pf, nf, pm, nm, S.cauchy != 0, S.tau, acc, inl_only);
                }
              }
            } else
            if (kNNLds) { if (live) best = nn_query<1, uint16_t, uint16_t>(g, l_cst, l_sidx, l_sxy, qx, qy, S.max_distance, md2, 0); }
            else
            if (use_grid) {
              if (live) best = (group == 1 && nn_lds) ? nn_query<1, uint16_t, uint16_t>(g, l_cst, l_sidx, l_sxy, qx, qy, S.max_distance, md2, 0)
                                                      : nn_query<group>(g, cst, sidx, sxy, qx, qy, S.max_distance, md2, sub);
            }
            else if (live) best = distmap_lookup(dm, S.fixed.dist.parent, qx, qy);
            if (best >= 0 && sub == 0) {                     // one lane per query accumulates
              const float2 nm = mn[j], nf = fn[best];
              float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
              const float dot = __builtin_fmaf(nqx, nf.x, nqy * nf.y);
              if (!(dot < S.normal_cos)) {
                if (want_dig) digest_add(&s_dig, salt, best, j);
                if constexpr (kSeq) seq_pair(fp[best], nf, pm, nm, t); else
                accumulate_pair(T, fp[best], nf, pm, nm, S.cauchy != 0, S.tau, acc, inl_only);
              }
            }
            if constexpr (kSeq) seq_trip(t, (j0 / per_step) & 1);      // "sum_order" 1: the trip's queries are per_step consecutive moving indices (one lane per query holds its pair: ascending thread = ascending query)
          }
        };
        // (the cooperative search on a scan-sized fixed cloud with its tables in LDS: 2 / 4 / 8 lanes per query take 21 / 41 / 90 ms against
        // 8.1 ms with one lane per query -- the time goes with the number of wave-queries, i.e. into the fixed cost of a query, not its candidates)
        // (several queries of a thread in flight together -- all points, then all pixels, then all parents -- measured with the registers
        // for it: 4 waves per SIMD and 3-4 trips tie with this loop at 8 waves per SIMD on role B and lose 10-50 % elsewhere; at 8 waves per SIMD
        // with only the parents' indices kept live, 2 / 3 trips take 0.29 / 0.38 ms against 0.21 on role B and lose on role A too; DESIGN App. A)
        // (matched pairs queued per wave in LDS and added up 64 at a time with every lane busy, instead of ~70 instructions of accumulate_pair on
        // every trip for the quarter of the lanes that matched: slower everywhere -- distance map role A 5.21 -> 5.87 ms, NN role A 7.95 -> 8.61,
        // distance map role B 0.21 -> 0.30: the ballot, the queue and the reloads cost more than the idle lanes; DESIGN App. A)
        if (!kNNLds && coop) query_loop(std::integral_constant<int, kNNGroup>{});
        else query_loop(std::integral_constant<int, 1>{});
      }
      if constexpr (kSeq) seq_flush();
      LSM2D_PH(0);
      // (a projective slice's thread accumulates at most ceil(cols / block) pairs: its counts are a few bits, summed by ballots)
      if constexpr (kW == kAlignBlock)      // (the narrow workgroups' virtual waves have written their rows in the walk)
      block_reduce_store(acc, red, tid, (kHasProj && !kHasNN && !kHasDist && !kHasKd) ? 32 - __builtin_clz(((S.proj.cols + kAlignBlock - 1) / kAlignBlock) | 1) : 0);
      __syncthreads();
      if (tid < 64) {
        // lanes 0..13 of wave 0 each add one quantity over the waves (wave order) and then into the iteration's sum themselves
        float v; int vi; block_reduce_gather_lane(red, kAlignBlock / 64, tid, v, vi);
        if constexpr (kSeq) v = seq_total(seq_acc, tid);      // (the partial sums' floats were never touched: the eleven quantities are the walker's)
        const int n_corr = __builtin_amdgcn_readlane(vi, 13);
        if (tid == 0) s_n_corr += n_corr;
        if (n_corr > S.min_corr) {     // slices with #pairs <= min_num_correspondences are skipped
          int ts = tid; asm volatile("" : "+v"(ts));      // (s_sum's address for this lane made here: hoisted, it was spilled across the iteration)
          if (tid < 11) s_sum[ts] += v;
          else if (tid < 13) s_sum[ts] = __int_as_float(__float_as_int(s_sum[ts]) + vi);
          if (tid == 0) ++s_active;
        }
      }
      LSM2D_PH(1);
      // pure projective kernels need no barrier here: the other waves go on to the next slice's projection (the cells it
      // writes were reset by the bin walk) and touch `red` again only after the barrier that follows it, which lane 0
      // joins once it is done with the partials.  The point-query branches write `red` without such a barrier in between.
      if (kHasNN || kHasDist || kHasKd) __syncthreads();
    }
    if (lists_only) {      // what the second launch will stream per iteration -> its placement; the state it goes on from
      if (tid == 0) {
        ResumeDev R;
        R.pose[0] = s_pose[0]; R.pose[1] = s_pose[1]; R.pose[2] = s_pose[2];
        for (int k = 0; k < 9; ++k) R.H[k] = s_H[k];
        R.prev_chi = s_prev_chi; R.phase = s_phase; R.phase_start = s_phase_start; R.phase_end = s_phase_end;
        R.last_n_in = s_last_n_in; R.status = s_status; R.done = 0; R.it = it;
        A.resume[a] = R;
        int units = 0;
        for (int s = 0; s < A.n_slices; ++s) units += s_nunits[s];
        const int w = (units + 7 * A.n_slices - 1) / (7 * A.n_slices);      // <= 512: a slice's list holds at most kCullBlocks x 512 units
        A.stage_work[a] = w < 1 ? 1 : (w > kAlignBlock ? kAlignBlock : w);
      }
      return;
    }
    if (tid == 0) {
      // (thread 0's serial state lives in LDS, not in registers every thread would carry -- and spill -- across the loops)
      StatsDev last; last.n_corr = s_n_corr; last.n_in = __float_as_int(s_sum[11]); last.n_out = __float_as_int(s_sum[12]); last.chi_in = s_sum[9]; last.chi_out = s_sum[10];
      s_last_n_in = last.n_in;
#ifdef LSM2D_DEBUG_UNITS      // diagnostics build: the culled stream's list length and whether it was rebuilt, in place of the outlier statistics
      if (kProjCulled) { last.n_out = s_nunits[0]; last.chi_out = (float) s_rebuild[0]; }
#endif
      if (A.out_stats) { const u64 dg = s_dig; last.dig_lo = (uint32_t) dg; last.dig_hi = (uint32_t) (dg >> 32); A.out_stats[(size_t) a * A.stats_stride + it] = last; }
      if (!s_active) { s_status = LSM2D_NOT_ENOUGH_CORRESPONDENCES; s_done = 1; }
      else {
        // information matrix = H of the last iteration: assembled, given its prior and solved where it lies
        s_H[0] = s_sum[0]; s_H[1] = s_sum[1]; s_H[2] = s_sum[2]; s_H[3] = s_sum[1]; s_H[4] = s_sum[3]; s_H[5] = s_sum[4];
        s_H[6] = s_sum[2]; s_H[7] = s_sum[4]; s_H[8] = s_sum[5];
        s_rhs[0] = s_sum[6]; s_rhs[1] = s_sum[7]; s_rhs[2] = s_sum[8];
        if (A.prior) add_prior(s_prior, s_pose, s_H, s_rhs);
        float dmp = A.damping;
        asm volatile("" : "+v"(dmp));      // (not a loop invariant to hoist -- as a double it was kept, and spilled, across the whole kernel)
        if (!solve_update(s_H, s_rhs, dmp, s_pose)) { s_status = LSM2D_SINGULAR_H; s_done = 1; }
        else {
          bool phase_over = it + 1 >= s_phase_end;
          if (A.term_eps > 0.0f) {      // the aligner's termination criterion: relative decay of the total chi^2 (lsm2d.h), afresh in every phase
            const float chi_now = s_sum[9] + s_sum[10];      // (= last.chi_in + last.chi_out, read again: kept in registers across the solve they were spilled)
            if (it > s_phase_start && __builtin_fabsf(s_prev_chi - chi_now) < A.term_eps * chi_now) phase_over = true;      // status stays RUNNING: decided below as after max_iterations
            s_prev_chi = chi_now;
          }
          if (phase_over) {
            // enable_inlier_only_runs (lsm2d.h): the regular loop ended without a failure and with enough inliers -> up to max_it iterations over inliers only
            if (A.inlier_runs && s_phase == 0 && last.n_in >= A.min_inliers) { s_phase = 1; s_phase_start = it + 1; s_phase_end = it + 1 + A.max_it; }
            else s_done = 1;
          }
        }
      }
      if (!s_done) begin_iteration();        // next iteration's transforms and zeroed sums, under the same barrier
      // the XCD lockstep: nobody starts its next pass before everybody on this XCD has finished the pass xcd_window back (the others stand at the barrier below anyway)
      if (kXcdWindow && xsync && !s_done) xcd_wait(xsync, (it + 1) * A.n_slices - 1 - A.xcd_window * A.n_slices, A.xcd_positions);
    }
    LSM2D_PH(2);
    __syncthreads();
    if (s_done) { ++it; break; }
  }
  if (kXcdWindow && xsync && tid == 0) __hip_atomic_fetch_add(&xsync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // gone: nobody waits for this workgroup any more
  if (tid == 0) {
    int st = s_status;
    if (st == LSM2D_RUNNING) st = (A.max_it > 0 && s_last_n_in < A.min_inliers) ? LSM2D_NOT_ENOUGH_INLIERS : LSM2D_SUCCESS;
    A.out_pose[3 * a + 0] = s_pose[0]; A.out_pose[3 * a + 1] = s_pose[1]; A.out_pose[3 * a + 2] = s_pose[2];
    if (A.out_H) for (int k = 0; k < 9; ++k) A.out_H[9 * a + k] = s_H[k];
    if (A.out_its) A.out_its[a] = it;
    if (kFirstStage) { A.resume[a].done = 1; A.stage_work[a] = 0; }      // finished before the second launch: its workgroup there leaves at once
    // the status goes last, behind a system-scope release: with results written straight to pinned host memory the host polls
    // this word instead of waiting for the stream (lsm2d_align_batch), and whoever sees it sees everything above
    if (stamp) {
      unsigned long long* co = A.clock_out + 4 * (a / A.clock_stride);
      co[0] = __builtin_amdgcn_s_memtime() - s_clk[0];
      co[1] = __builtin_amdgcn_s_memrealtime() - s_clk[1];
      co[2] = s_clk[1];                                                            // when it started (100 MHz ticks): which dispatch round it was in
#ifdef LSM2D_PHASE_PROBE
      co[1] = s_ph[0]; co[2] = s_ph[1]; co[3] = s_ph[2];      // cycles: query phase, barrier + reduction, solve + the rest (co[0] stays the lifetime)
      if (A.host_polls) { __threadfence_system(); __hip_atomic_store(&A.out_status[a], st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); } else A.out_status[a] = st;
      return;
#endif
      co[3] = (unsigned long long) __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) |      // HW_REG_HW_ID (id 4): wave / SIMD / CU / SE it ran on
              ((unsigned long long) __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32);  // HW_REG_XCC_ID (id 20)
    }
    if (A.host_polls) { __threadfence_system(); __hip_atomic_store(&A.out_status[a], st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    else A.out_status[a] = st;
  }
}
// Registers: 8 waves per SIMD (64 VGPRs, four workgroups per CU) for every single-finder instantiation.  The MIXED instantiations (a projective slice next to a
// point-query slice in one aligner: all forms of all finders in one body) spilled 176 bytes per thread at that budget; they are given 4 waves per SIMD
// (128 VGPRs, two workgroups per CU) and have no private segment -- a configuration no BASELINE workload uses (round 5; tools/isa_dump.sh prints every kernel's frame).
#ifndef LSM2D_MIXED_MIN_WAVES
#define LSM2D_MIXED_MIN_WAVES 4
#endif
#ifndef LSM2D_NNGLOBAL_MIN_WAVES
#define LSM2D_NNGLOBAL_MIN_WAVES LSM2D_QUERY_MIN_WAVES      // k_align<0,1,0,0,1>: A/B knob (6: 80 VGPRs, no frame, three workgroups per CU)
#endif
template <bool kHasProj, bool kHasNN, bool kHasDist, bool kHasKd, int kNNMode>
constexpr int align_min_waves() {
  return (kHasProj && (kHasNN || kHasDist || kHasKd)) ? LSM2D_MIXED_MIN_WAVES : kHasProj ? LSM2D_ALIGN_MIN_WAVES : kNNMode == 1 ? LSM2D_NNGLOBAL_MIN_WAVES : LSM2D_QUERY_MIN_WAVES;
}
template <bool kHasProj, bool kHasNN, bool kHasDist, bool kHasKd = false, int kNNMode = 0>
__global__ __launch_bounds__(kAlignBlock, (align_min_waves<kHasProj, kHasNN, kHasDist, kHasKd, kNNMode>())) void k_align(const AlignArgs A) {
  align_body<kHasProj, kHasNN, kHasDist, kHasKd, kNNMode, false>(A);
}
#ifndef LSM2D_SEQ_MIN_WAVES
#define LSM2D_SEQ_MIN_WAVES 8
#endif
// The culled projective stream with up to TWO alignments per workgroup, one after the other (AlignArgs::order2): the whole body again, from its prologue -- an
// instantiation of its own (two copies of the body), so that the headline's kernel does not carry a loop around it (in one kernel: 80 bytes of scratch).
__global__ __launch_bounds__(kAlignBlock, LSM2D_ALIGN_MIN_WAVES) void k_align_two(const AlignArgs A) {
  align_body<true, false, false, false, 5, false>(A, __builtin_amdgcn_readfirstlane(A.order[blockIdx.x]));
  const int a2 = __builtin_amdgcn_readfirstlane(A.order2[blockIdx.x]);
  if (a2 < 0) return;
  __syncthreads();      // (thread 0 is done with the first one's results before anybody overwrites the state they came from)
  align_body<true, false, false, false, 5, false>(A, a2);
}
// ... and the same with the reference's order of summation ("sum_order" 1: no narrow form exists for it, so every batch between one and two rounds is packed)
__global__ __launch_bounds__(kAlignBlock, LSM2D_SEQ_MIN_WAVES) void k_align_seq_two(const AlignArgs A) {
  align_body<true, false, false, false, 5, false, true>(A, __builtin_amdgcn_readfirstlane(A.order[blockIdx.x]));
  const int a2 = __builtin_amdgcn_readfirstlane(A.order2[blockIdx.x]);
  if (a2 < 0) return;
  __syncthreads();
  align_body<true, false, false, false, 5, false, true>(A, a2);
}
// The culled projective stream in NARROW workgroups (align_body's kW): 256 threads, six workgroups -- 1536 alignments -- resident per round where the wide kernel
// holds 1024.  The same results as k_align<1,0,0,0,5>, bit for bit; chosen by the host (align_width_for) for batches just above a multiple of 1024 alignments.
template <int kW>
__global__ __launch_bounds__(kW, LSM2D_ALIGN_MIN_WAVES) void k_align_narrow(const AlignArgs A) {
  align_body<true, false, false, false, 5, false, false, kW>(A);
}
// "sum_order" 1: the same kernel with the reference's order of summation (align_body<.., kSeq = true>).  14 KB of pair records per workgroup beside the canvases (four workgroups per CU still fit the headline's shape):
// the register budget stays that of 8 waves per SIMD -- 4 for the mixed instantiations, as above
template <bool kHasProj, bool kHasNN, bool kHasDist, bool kHasKd = false, int kNNMode = 0>
__global__ __launch_bounds__(kAlignBlock, ((kHasProj && (kHasNN || kHasDist || kHasKd)) ? LSM2D_MIXED_MIN_WAVES : LSM2D_SEQ_MIN_WAVES)) void k_align_seq(const AlignArgs A) {
  align_body<kHasProj, kHasNN, kHasDist, kHasKd, kNNMode, false, true>(A);
}
// Round 4 (late): TWO launches for a culled batch of about one dispatch round.  The placement of such a batch decides its tail (the launch lasts as long as
// the CU with the largest sum of work), and what an alignment will stream is known badly at its START pose -- the estimate of k_cull_estimate left a tail of
// 10 % -- but well after ONE Gauss-Newton iteration, which takes most of the start error out.  So: this kernel runs iteration 0 of every alignment in any
// order (a twentieth of the work: its own tail does not matter), builds the unit lists of iteration 1 for their length alone, and leaves pose, phase and
// statistics state in ResumeDev; k_balance_only deals the alignments out by those lengths; k_align runs the other nineteen iterations from the saved state.
// The same arithmetic on the same values in the same order: every result keeps its bits (the second launch rebuilds its lists; a list is a superset of
// what can pair, whatever pose within its margins it was built at).  MEASURED AND NOT SHIPPED ("two_stage" 0 by default): the second launch takes 0.698 ms
// instead of 0.745 -- but a twentieth of that is the iteration it no longer runs, its tail is still 7 % (the length of a list is not the whole of an
// alignment's cost), and this kernel takes 95 us for its twentieth of the work: all thousand workgroups are in the same phase at the same time, and the
// phases that wait (prologue, list building, barriers) have no other workgroup's stream to hide under.  0.861 vs 0.836 ms per step.
#ifdef LSM2D_EXPERIMENTS
__global__ __launch_bounds__(kAlignBlock, LSM2D_ALIGN_MIN_WAVES) void k_first_iteration(const AlignArgs A) {
  align_body<true, false, false, false, 5, true>(A);
}
#endif
