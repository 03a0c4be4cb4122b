// lsm2d_k_mapping.h -- the steps either side of the aligner: scene clipper, merger, voxelisation, raw-data preprocessor (mapping/scene_clipper_projective_2d.cpp:24-62, mapping/merger_projective_2d.cpp:16-97, sensor_processing/raw_data_preprocessor_projective_2d.cpp:23-104).
// Part of lsm2d_kernels.h (included there, inside namespace lsm2d, in this order); not a translation unit of its own.
// ---- mapping kernels around the aligner: the same polar z-buffer, spread over many
//      workgroups for one big cloud, then an O(Bins) pass.  -----------------------------------------------
struct ProjectSplitArgs {
  const float2* xy; int32_t n; Iso T; ProjK proj;
  u64* gcanvas;            // [cols], pre-filled with kEmptyCell
};

// each workgroup z-buffers a contiguous slice of the cloud in LDS, then folds its canvas into the global one
__global__ __launch_bounds__(512) void k_project_split(const ProjectSplitArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* can = reinterpret_cast<u64*>(smem);
  const int tid = threadIdx.x;
  for (int i = tid; i < A.proj.cols; i += 512) can[i] = kEmptyCell;
  __syncthreads();
  const Iso T = A.T; const ProjK P = A.proj;
  const int npairs = (A.n + 1) >> 1;
  const int per = (npairs + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = lo + per < npairs ? lo + per : npairs;
  const float4* xy4 = reinterpret_cast<const float4*>(A.xy);
  for (int j = lo + tid; j < hi; j += 512) {
    const float4 v = xy4[j];
    project_point(T, P, v.x, v.y, 2 * j, can);
    if (2 * j + 1 < A.n) project_point(T, P, v.z, v.w, 2 * j + 1, can);
  }
  __syncthreads();
  for (int i = tid; i < P.cols; i += 512) { const u64 k = can[i]; if (k != kEmptyCell) atomicMin(&A.gcanvas[i], k); }
}

LSM2D_DEV int block_compact_offset(bool flag, int* s_wave_tot, int* s_base, int tid, int nwaves) {
  // order-preserving position of this thread's element among the flagged ones (all threads must call)
  const int lane = tid & 63, wave = tid >> 6;
  const u64 bal = __ballot(flag);
  const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) s_wave_tot[wave] = __popcll(bal);
  __syncthreads();
  int before = *s_base, total = 0;
  for (int w = 0; w < nwaves; ++w) { const int t = s_wave_tot[w]; if (w < wave) before += t; total += t; }
  __syncthreads();
  if (tid == 0) *s_base += total;
  __syncthreads();
  return before + prefix;
}

// the same with ONE barrier per call: the per-wave totals alternate between two buffers (`parity`: 0, 1, 0, ... from call to call; the
// barrier of call i + 1 separates the reads of call i from the writes of call i + 2), and every thread keeps the running base itself
// (`base`, the same value in all threads; in: flagged elements so far, out: including this call's)
LSM2D_DEV int block_compact_pos(bool flag, int* s_tot /* [2][nwaves] */, int parity, int& base, int tid, int nwaves) {
  const int lane = tid & 63, wave = tid >> 6;
  const u64 bal = __ballot(flag);
  const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
  int* t = s_tot + parity * nwaves;
  if (lane == 0) t[wave] = __popcll(bal);
  __syncthreads();
  int before = base, total = 0;
  for (int w = 0; w < nwaves; ++w) { const int v = t[w]; if (w < wave) before += v; total += v; }
  // the running base is the same in every lane: say so (a count that came out of LDS reads is a per-lane value to the compiler, and loops
  // bounded by it compile to per-lane forms -- the preprocessor's window walks ran 58 % slower over a batch before this line)
  base = __builtin_amdgcn_readfirstlane(base + total);
  return before + prefix;
}

// SceneClipperProjective2D::compute tail (mapping/scene_clipper_projective_2d.cpp:53-63): filled cells in ascending
// column -> transformed point (sensor frame), then moved to the robot frame by sensor_in_robot
struct ClipEmitArgs {
  const u64* gcanvas; int32_t cols;
  const float2* xy; const float2* nrm;       // full scene
  Iso T;                                      // sensor_in_local_map^-1
  Iso S; int32_t s_identity;                  // sensor_in_robot
  float2* out_xy; float2* out_nrm; int32_t* out_src; int32_t* out_count_dev /* count[0] of the clipped set */; int32_t* out_count;
  int32_t host_polls;                          // out_src / out_count are pinned host memory and the host polls out_count: write it last, released to the system
};

__global__ __launch_bounds__(kFindBlock) void k_clip_emit(const ClipEmitArgs A);

// small scenes (the tracker's local map between key frames): clipper and merger as ONE workgroup-resident kernel each --
// z-buffers in LDS, no global canvas, no memsets, one launch instead of three resp. six
// n_dev: when non-null the scene's size is only known on the device (its set was last written by an asynchronous clip / merge)
struct ClipSmallArgs { const float2* xy; const float2* nrm; int32_t n; const int32_t* n_dev; ProjK proj; ClipEmitArgs emit; };

LSM2D_DEV void clip_emit_body(const ClipEmitArgs& A, const u64* canvas, int* s_tot /* [2][kFindBlock / 64] */, int tid) {
  int base = 0, parity = 0;
  for (int c0 = 0; c0 < A.cols; c0 += kFindBlock, parity ^= 1) {
    const int col = c0 + tid;
    const u64 k = col < A.cols ? canvas[col] : kEmptyCell;
    const bool ok = k != kEmptyCell;
    const int pos = block_compact_pos(ok, s_tot, parity, base, tid, kFindBlock / 64);
    if (ok) {
      const int src = (int) (uint32_t) k;
      const float2 p = A.xy[src], n = A.nrm[src];
      float x, y, nx, ny;
      xf_point(A.T, p.x, p.y, x, y); xf_normal(A.T, n.x, n.y, nx, ny);
      if (!A.s_identity) {
        float tx, ty, tnx, tny;
        xf_point(A.S, x, y, tx, ty); xf_normal(A.S, nx, ny, tnx, tny);
        x = tx; y = ty; nx = tnx; ny = tny;
      }
      A.out_xy[pos] = make_float2(x, y); A.out_nrm[pos] = make_float2(nx, ny);
      if (A.out_src) A.out_src[pos] = src;
    }
  }
  if (A.host_polls) {      // the count goes last, behind every thread's system-scope release of its rows: the synchronous form's host side polls it
    __threadfence_system();
    __syncthreads();
    if (tid == 0) { *A.out_count_dev = base; __hip_atomic_store(A.out_count, base, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  } else if (tid == 0) { *A.out_count = base; *A.out_count_dev = base; }
}

__global__ __launch_bounds__(kFindBlock) void k_clip_small(const ClipSmallArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* can = reinterpret_cast<u64*>(smem);
  __shared__ int s_tot[2 * (kFindBlock / 64)];
  const int tid = threadIdx.x;
  for (int i = tid; i < A.proj.cols; i += kFindBlock) can[i] = kEmptyCell;
  __syncthreads();
  project_cloud(A.xy, A.n_dev ? *A.n_dev : A.n, A.emit.T, A.proj, can, tid, kFindBlock);
  __syncthreads();
  clip_emit_body(A.emit, can, s_tot, tid);
}

__global__ __launch_bounds__(kFindBlock) void k_clip_emit(const ClipEmitArgs A) {
  __shared__ int s_tot[2 * (kFindBlock / 64)];
  clip_emit_body(A, A.gcanvas, s_tot, threadIdx.x);
}

// transform a cloud (measurement -> scene frame, mapping/merger_projective_2d.cpp:22-23)
__global__ void k_transform_cloud(const float2* __restrict__ xy, const float2* __restrict__ nrm, int n, const Iso T,
                                  float2* __restrict__ oxy, float2* __restrict__ onrm) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float2 p = xy[i], q = nrm[i];
    float x, y, nx, ny;
    xf_point(T, p.x, p.y, x, y); xf_normal(T, q.x, q.y, nx, ny);
    oxy[i] = make_float2(x, y); onrm[i] = make_float2(nx, ny);
  }
}

// MergerProjective2D::compute column walk (mapping/merger_projective_2d.cpp:39-95)
struct MergeArgs {
  const u64* scanvas; const u64* mcanvas; int32_t cols;
  float2* sxy; float2* snrm; int32_t n_scene;        // scene, updated in place and appended to
  const float2* mxy; const float2* mnrm;             // measurement already in the scene frame
  float far_limit, merge_threshold;
  int32_t* out;                                       // [4]: new size, new, merged, replaced
  int32_t* count_dev;                                 // count[0] of the scene set
  int32_t host_polls;                                 // out is pinned host memory and the host polls out[0]: write it last, released to the system
};

// mkT: when non-null the measurement is still in its own frame and is moved by *mkT on the fly (fused small-scene kernel);
// the transform is the same operation sequence as k_transform_cloud, so both forms give the same bits
// returns the number of appended points (the same value in every thread); s_tot: [2][kFindBlock / 64]; s_cnt: [1] new, [2] merged, [3] replaced,
// zeroed by the caller, read here by thread 0 behind a barrier of its own
LSM2D_DEV int merge_apply_body(const MergeArgs& A, const u64* scanvas, const u64* mcanvas, const Iso* mkT, int* s_tot, int* s_cnt, int tid) {
  int appended = 0, parity = 0;
  for (int c0 = 0; c0 < A.cols; c0 += kFindBlock, parity ^= 1) {
    const int col = c0 + tid;
    bool append = false; float2 mp = make_float2(0.f, 0.f), mn = mp;
    if (col < A.cols) {
      const u64 mk = mcanvas[col], sk = scanvas[col];
      const float md = __uint_as_float((uint32_t) (mk >> 32));
      if (mk != kEmptyCell && !(md > A.far_limit)) {
        const int mi = (int) (uint32_t) mk;
        mp = A.mxy[mi]; mn = A.mnrm[mi];
        if (mkT) {
          float x, y, nx, ny;
          xf_point(*mkT, mp.x, mp.y, x, y); xf_normal(*mkT, mn.x, mn.y, nx, ny);
          mp = make_float2(x, y); mn = make_float2(nx, ny);
        }
        if (sk == kEmptyCell) { append = true; atomicAdd(&s_cnt[1], 1); }
        else {
          const int si = (int) (uint32_t) sk;
          const float dr = md - __uint_as_float((uint32_t) (sk >> 32));
          if (__builtin_fabsf(dr) < A.merge_threshold) {
            const float2 sp = A.sxy[si], sn = A.snrm[si];
            const float x = (sp.x + mp.x) * 0.5f, y = (sp.y + mp.y) * 0.5f;
            float nx = (sn.x + mn.x) * 0.5f, ny = (sn.y + mn.y) * 0.5f;
            const float nn = __builtin_sqrtf(__builtin_fmaf(nx, nx, ny * ny));
            if (nn > 0.0f) { nx = nx / nn; ny = ny / nn; }
            A.sxy[si] = make_float2(x, y); A.snrm[si] = make_float2(nx, ny);
            atomicAdd(&s_cnt[2], 1);
          } else if (dr > 0.0f) { A.sxy[si] = mp; A.snrm[si] = mn; atomicAdd(&s_cnt[3], 1); }
          else append = true;
        }
      }
    }
    const int pos = block_compact_pos(append, s_tot, parity, appended, tid, kFindBlock / 64);
    if (append) { A.sxy[A.n_scene + pos] = mp; A.snrm[A.n_scene + pos] = mn; }
  }
  if (A.host_polls) __threadfence_system();      // every thread's rows, ahead of the size the host polls
  __syncthreads();                                // the counters are final (and, with host_polls, every thread's rows are released)
  if (tid == 0) {      // the new size goes last; the synchronous form's host side polls it in pinned memory: released to the system then
    A.out[1] = s_cnt[1]; A.out[2] = s_cnt[2]; A.out[3] = s_cnt[3]; *A.count_dev = A.n_scene + appended;
    if (A.host_polls) { __threadfence_system(); __hip_atomic_store(&A.out[0], A.n_scene + appended, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    else A.out[0] = A.n_scene + appended;
  }
  return appended;
}

__global__ __launch_bounds__(kFindBlock) void k_merge_apply(const MergeArgs A) {
  __shared__ int s_tot[2 * (kFindBlock / 64)];
  __shared__ int s_cnt[4];
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  merge_apply_body(A, A.scanvas, A.mcanvas, nullptr, s_tot, s_cnt, threadIdx.x);
}

// small scene: transform + both projections + column walk in one workgroup (mxy / mnrm hold the measurement in ITS frame)
struct MergeSmallArgs { MergeArgs m; ProjK proj; Iso Tinv, M; int32_t n_meas; const int32_t* n_scene_dev; const int32_t* n_meas_dev; };   // *_dev: see ClipSmallArgs

__global__ __launch_bounds__(kFindBlock) void k_merge_small(const MergeSmallArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* scan = reinterpret_cast<u64*>(smem);
  u64* mcan = scan + A.proj.cols;
  __shared__ int s_tot[2 * (kFindBlock / 64)];
  __shared__ int s_cnt[4];
  const int tid = threadIdx.x;
  for (int i = tid; i < A.proj.cols; i += kFindBlock) { scan[i] = kEmptyCell; mcan[i] = kEmptyCell; }
  if (tid < 4) s_cnt[tid] = 0;
  __syncthreads();
  MergeArgs m = A.m;
  if (A.n_scene_dev) m.n_scene = *A.n_scene_dev;
  const int n_meas = A.n_meas_dev ? *A.n_meas_dev : A.n_meas;
  project_cloud(m.sxy, m.n_scene, A.Tinv, A.proj, scan, tid, kFindBlock);
  for (int i = tid; i < n_meas; i += kFindBlock) {               // measurement -> scene frame -> camera frame
    const float2 p = m.mxy[i];
    float x, y; xf_point(A.M, p.x, p.y, x, y);
    project_point(A.Tinv, A.proj, x, y, i, mcan);
  }
  __syncthreads();
  merge_apply_body(m, scan, mcan, &A.M, s_tot, s_cnt, tid);
}

// several measurements merged into the scene one after the other by ONE launch (lsm2d_merge_scenes: the live tracker's front and
// rear scan): the same passes as k_merge_small per measurement, the scene's new size carried from one to the next in the workgroup
static constexpr int kMergeMulti = 4;
struct MergeMultiArgs { MergeSmallArgs a[kMergeMulti]; int32_t n; };
__global__ __launch_bounds__(kFindBlock) void k_merge_multi(const MergeMultiArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* scan = reinterpret_cast<u64*>(smem);
  u64* mcan = scan + A.a[0].proj.cols;                             // one projector for all of them
  __shared__ int s_tot[2 * (kFindBlock / 64)];
  __shared__ int s_cnt[4];
  const int tid = threadIdx.x;
  // sizes only the device knows: all of them up front, the loads in flight together (not one round trip per measurement)
  int n_scene = A.a[0].n_scene_dev ? *A.a[0].n_scene_dev : A.a[0].m.n_scene;
  int n_meas_of[kMergeMulti];
#pragma unroll
  for (int k = 0; k < kMergeMulti; ++k) n_meas_of[k] = k < A.n ? (A.a[k].n_meas_dev ? *A.a[k].n_meas_dev : A.a[k].n_meas) : 0;
  for (int k = 0; k < A.n; ++k) {
    const MergeSmallArgs& S = A.a[k];
    for (int i = tid; i < S.proj.cols; i += kFindBlock) { scan[i] = kEmptyCell; mcan[i] = kEmptyCell; }
    if (tid < 4) s_cnt[tid] = 0;
    __syncthreads();
    MergeArgs m = S.m;
    m.n_scene = n_scene;
    const int n_meas = k == 0 ? n_meas_of[0] : (k == 1 ? n_meas_of[1] : (k == 2 ? n_meas_of[2] : n_meas_of[3]));
    project_cloud(m.sxy, m.n_scene, S.Tinv, S.proj, scan, tid, kFindBlock);
    for (int i = tid; i < n_meas; i += kFindBlock) {
      const float2 p = m.mxy[i];
      float x, y; xf_point(S.M, p.x, p.y, x, y);
      project_point(S.Tinv, S.proj, x, y, i, mcan);
    }
    __syncthreads();
    n_scene = m.n_scene + merge_apply_body(m, scan, mcan, &S.M, s_tot, s_cnt, tid);      // what the next measurement is merged into
    __syncthreads();                         // thread 0 has read the counters; rows and canvases are free for the next measurement
  }
}

// split a single device cloud back into AoS (download)
__global__ void k_pack_aos(const float2* __restrict__ xy, const float2* __restrict__ nrm, int n, float4* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float2 p = xy[i], q = nrm[i];
    out[i] = make_float4(p.x, p.y, q.x, q.y);
  }
}

// ---- PointCloud::voxelize as the reference uses it (sensor_processing/raw_data_preprocessor_projective_2d.cpp:38-41 with
// coefficients (res, res, 1, 1); mapping/scene_clipper_projective_2d.cpp:44-48 with (res, res, 0.1, 0.1); assumption F2.3,
// PARITY.md section 3): k points staged in LDS (s_q coordinates, s_n normals), key = floor of (x, y) * inv_rx and of the normal
// components * inv_rn, equal keys averaged (normal re-normalised), voxels in ascending lexicographic key order.  One workgroup of
// kVoxBlock threads, k <= kVoxMax; emit(pos, x, y, nx, ny) is called once per voxel; returns the number of voxels (every thread).
static constexpr int kVoxBlock = 1024;
static constexpr int kVoxMax = 2048;
template <int kBlock = kVoxBlock, typename Emit>
LSM2D_DEV int voxelize_lds(const float2* s_q, const float2* s_n, u64* s_key, int k, float inv_rx, float inv_rn, int* s_tot /* [2][kBlock / 64] */,
                           int tid, Emit emit) {
  int np2 = 1; while (np2 < k) np2 <<= 1;
  for (int i = tid; i < np2; i += kBlock) {
    u64 key = ~0ull;
    if (i < k) {
      const float kx = __builtin_floorf(s_q[i].x * inv_rx), ky = __builtin_floorf(s_q[i].y * inv_rx);
      const float knx = __builtin_floorf(s_n[i].x * inv_rn), kny = __builtin_floorf(s_n[i].y * inv_rn);
      if (kx >= -32768.0f && kx < 32768.0f && ky >= -32768.0f && ky < 32768.0f && knx >= -16.0f && knx <= 15.0f && kny >= -16.0f && kny <= 15.0f) {
        const u64 v = ((u64) ((int) kx + 32768) << 26) | ((u64) ((int) ky + 32768) << 10) | ((u64) ((int) knx + 16) << 5) | (u64) ((int) kny + 16);
        key = (v << 16) | (u64) i;
      }
    }
    s_key[i] = key;
  }
  __syncthreads();
  // bitonic network, one compare-exchange per thread and step (np2 / 2 <= kVoxBlock).  Pair t touches elements inside the
  // aligned 128-element block of its wave whenever stride <= 64, so those steps need no workgroup barrier -- LDS operations
  // of one wave complete in order -- only the compiler must keep them in order (wavefront fence).  6 of the 55 steps of a
  // 1024-key sort cross waves.
  // (a smaller workgroup -- the batch preprocessor's 512 threads -- takes its pairs t = tid, tid + kBlock, ...: pair t of a wave still lies in ONE aligned
  // 128-element block, the same one in every step, so the wave-local ordering holds per trip)
  static_assert(kVoxMax / 2 <= kVoxBlock && kBlock % 64 == 0, "one compare-exchange per thread at the full block size");
  for (int size = 2; size <= np2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma nounroll
      for (int t = tid; t < (np2 >> 1); t += kBlock) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const u64 a = s_key[lo], b = s_key[hi];
        if ((a > b) == up) { s_key[lo] = b; s_key[hi] = a; }
      }
      // the next step's stride is stride / 2, or `size` when this was the last step of its stage
      if (stride > 64 || (stride == 1 && size > 64)) __syncthreads();
      else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    }
  }
  __syncthreads();
  int nv = 0, parity = 0;
  for (int t0 = 0; t0 < np2; t0 += kBlock, parity ^= 1) {
    const int t = t0 + tid;
    bool head = false; u64 key = ~0ull;
    if (t < np2) { key = s_key[t]; head = key != ~0ull && (t == 0 || (s_key[t - 1] >> 16) != (key >> 16)); }
    const int pos = block_compact_pos(head, s_tot, parity, nv, tid, kBlock / 64);
    if (head) {
      float ax = 0.0f, ay = 0.0f, anx = 0.0f, any_ = 0.0f; int cnt = 0;
      for (int e = t; e < np2 && (s_key[e] >> 16) == (key >> 16); ++e) {
        const int i = (int) (s_key[e] & 0xFFFFull);
        ax += s_q[i].x; ay += s_q[i].y; anx += s_n[i].x; any_ += s_n[i].y; ++cnt;
      }
      const float inv = 1.0f / (float) cnt;
      ax *= inv; ay *= inv; anx *= inv; any_ *= inv;
      const float nn = __builtin_sqrtf(__builtin_fmaf(anx, anx, any_ * any_));
      if (nn > 0.0f) { anx = anx / nn; any_ = any_ / nn; }
      emit(pos, ax, ay, anx, any_);
    }
  }
  return nv;
}

// the clipper's voxelize_resolution > 0 branch (mapping/scene_clipper_projective_2d.cpp:36-48,60-62): the clipped cloud -- written
// by the clip kernels in the SENSOR frame, ascending column -- is voxelised with coefficients (res, res, 0.1, 0.1) and only then
// moved to the robot frame by sensor_in_robot.  In place: everything is staged in LDS first and a voxelised cloud never grows.
struct VoxArgs {
  float2* xy; float2* nrm; int32_t* count_dev;      // the clipped set (one cloud)
  float inv_rx, inv_rn; Iso S; int32_t s_identity;
  int32_t* out_count; int32_t host_polls;            // the synchronous form's count, in pinned memory, written last
};
__global__ __launch_bounds__(kVoxBlock) void k_voxelize_clipped(const VoxArgs A) {
  __shared__ float2 s_q[kVoxMax];
  __shared__ float2 s_n[kVoxMax];
  __shared__ u64 s_key[kVoxMax];
  __shared__ int s_tot[2 * (kVoxBlock / 64)];
  const int tid = threadIdx.x;
  int k = *A.count_dev; if (k > kVoxMax) k = kVoxMax;      // the host refuses canvases beyond kVoxMax columns
  for (int i = tid; i < k; i += kVoxBlock) { s_q[i] = A.xy[i]; s_n[i] = A.nrm[i]; }
  __syncthreads();
  const int nv = voxelize_lds(s_q, s_n, s_key, k, A.inv_rx, A.inv_rn, s_tot, tid, [&](int pos, float x, float y, float nx, float ny) {
    if (!A.s_identity) {
      float tx, ty, tnx, tny;
      xf_point(A.S, x, y, tx, ty); xf_normal(A.S, nx, ny, tnx, tny);
      x = tx; y = ty; nx = tnx; ny = tny;
    }
    A.xy[pos] = make_float2(x, y); A.nrm[pos] = make_float2(nx, ny);
  });
  if (A.host_polls) {
    __threadfence_system();
    __syncthreads();
    if (tid == 0) { *A.count_dev = nv; __hip_atomic_store(A.out_count, nv, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  } else if (tid == 0) { *A.count_dev = nv; if (A.out_count) *A.out_count = nv; }
}

// ---- RawDataPreprocessorProjective2D (row f2): one workgroup per scan, everything in LDS -----------------------
static constexpr int kPrepBlock = kVoxBlock;     // one beam per thread for the window walks; one compare-exchange per thread in the sort
static constexpr int kPrepMaxBeams = kVoxMax;
struct PrepArgs {
  const float* ranges; const float2* beam_dir;      // [n_scans][n_beams]; (cos, sin) per beam, host-computed
  int32_t n_beams, stride;                           // stride: points reserved per output cloud (even)
  float rmin, rmax, d2max; int32_t min_points; float inv_res;   // inv_res <= 0: no voxelisation
  float2* out_xy; float2* out_nrm; int32_t* out_count;
  float4* out_aos = nullptr;                         // the set's (x, y, nx, ny) rows, rewritten in place by a refill (lsm2d_preprocess_scans_refill), or nullptr
};

// kBlock threads, room for kCap beams.  (1024, 2048): one beam per thread, what a scan alone on the chip wants (the live tracker).  (512, 1152), round 5: a
// BATCH of scans preprocessed beside a k_align launch in flight -- a workgroup of 512 threads and 37 KB is exactly what one retiring k_align workgroup leaves
// free, where the 1024-thread, 64 KB form had to wait for two slots of one CU to come free together; four of them per CU when the chip is theirs.
template <int kBlock, int kCap>
LSM2D_DEV void preprocess_scan_body(const PrepArgs& A, const int scan) {
  constexpr int kPrepBlock = kBlock;         // (shadows the full-size constant: every loop below strides by the workgroup's own size)
  // (the sort pads to a power of two: its keys need 2048 entries as soon as more than 1024 points carry a normal.  The small form cannot afford them beside the
  // three point arrays -- so its keys LIVE where the unprojected points were: those are dead once the normals are out, a barrier before the first key is written)
  constexpr int kKeyCap = kCap <= 1024 ? 1024 : 2048;
  constexpr bool kKeysOverPoints = kCap < kPrepMaxBeams;
  __shared__ u64 s_key[kKeyCap];             // (voxel key << 16) | index, bitonic-sorted
  __shared__ float2 s_p_own[kKeysOverPoints ? 1 : kCap];
  float2* const s_p = kKeysOverPoints ? reinterpret_cast<float2*>(s_key) : s_p_own;      // unprojected points, beam order
  __shared__ float2 s_q[kCap];               // points that got a normal
  __shared__ float2 s_n[kCap];               // their normals
  __shared__ int s_tot[2 * (kPrepBlock / 64)];
  const int tid = threadIdx.x, nb = A.n_beams;
#ifdef LSM2D_PHASE_CLOCKS
  unsigned long long pc_t = __builtin_amdgcn_s_memrealtime(), pc_acc[6] = {0, 0, 0, 0, 0, 0};
#define LSM2D_PC(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); pc_acc[k] += n_ - pc_t; pc_t = n_; } while (0)
#else
#define LSM2D_PC(k) do { } while (0)
#endif
  const float* rg = A.ranges + (size_t) scan * nb;
  float2* oxy = A.out_xy + (size_t) scan * A.stride; float2* onr = A.out_nrm + (size_t) scan * A.stride;
  float4* oaos = A.out_aos ? A.out_aos + (size_t) scan * A.stride : nullptr;
  // ---- F2.1 unprojection, valid beams compacted in beam order
  int m = 0, parity = 0;
  for (int c0 = 0; c0 < nb; c0 += kPrepBlock, parity ^= 1) {
    const int c = c0 + tid;
    float r = 0.0f; bool ok = false; float2 d = make_float2(0.0f, 0.0f);
    if (c < nb) { r = rg[c]; d = A.beam_dir[c]; ok = r >= A.rmin && r <= A.rmax; }
    const int pos = block_compact_pos(ok, s_tot, parity, m, tid, kPrepBlock / 64);
    if (ok) s_p[pos] = make_float2(r * d.x, r * d.y);
  }
  __syncthreads();
  LSM2D_PC(0);
  // ---- F2.2 sliding-window normals
  int k = 0;
  for (int i0 = 0; i0 < m; i0 += kPrepBlock, parity ^= 1) {
    const int i = i0 + tid;
    bool ok = false; float vx = 0.0f, vy = 0.0f; float2 pi = make_float2(0.0f, 0.0f);
    if (i < m) {
      pi = s_p[i];
      // Round 5: the window walks and the two sums were one LDS round trip per step (load -> compare -> branch; load -> add), 13-33 us of a scan's ~35 -- the
      // longest window of the workgroup (hundreds of points at close range) sets the pace.  Four neighbours per trip now: the loads of a trip are independent, the
      // tests and the additions keep the reference's order (first neighbour that fails ends the walk; sums run j = lo .. hi one after the other): same bits.
      const auto within = [&](int j) { const float2 q = s_p[j]; const float dx = q.x - pi.x, dy = q.y - pi.y; return __builtin_fmaf(dx, dx, dy * dy) <= A.d2max; };
      int lo = i, hi = i;
      while (lo > 0) {
        const int j1 = lo - 1, j2 = lo >= 2 ? lo - 2 : 0, j3 = lo >= 3 ? lo - 3 : 0, j4 = lo >= 4 ? lo - 4 : 0;      // (clamped: the extra loads are of valid cells and never counted)
        const bool f1 = within(j1), f2 = within(j2), f3 = within(j3), f4 = within(j4);
        const int room = lo < 4 ? lo : 4;
        const int adv = !f1 ? 0 : (room < 2 || !f2) ? 1 : (room < 3 || !f3) ? 2 : (room < 4 || !f4) ? 3 : 4;
        lo -= adv;
        if (adv < 4) break;
      }
      while (hi < m - 1) {
        const int last = m - 1, j1 = hi + 1, j2 = hi + 2 <= last ? hi + 2 : last, j3 = hi + 3 <= last ? hi + 3 : last, j4 = hi + 4 <= last ? hi + 4 : last;
        const bool f1 = within(j1), f2 = within(j2), f3 = within(j3), f4 = within(j4);
        const int room = last - hi < 4 ? last - hi : 4;
        const int adv = !f1 ? 0 : (room < 2 || !f2) ? 1 : (room < 3 || !f3) ? 2 : (room < 4 || !f4) ? 3 : 4;
        hi += adv;
        if (adv < 4) break;
      }
      const int cnt = hi - lo + 1;
      if (cnt >= A.min_points) {
        float sx = 0.0f, sy = 0.0f;
        int j = lo;
        for (; j + 3 <= hi; j += 4) {
          const float2 a = s_p[j], b = s_p[j + 1], c = s_p[j + 2], d = s_p[j + 3];
          sx += a.x; sy += a.y; sx += b.x; sy += b.y; sx += c.x; sy += c.y; sx += d.x; sy += d.y;
        }
        for (; j <= hi; ++j) { sx += s_p[j].x; sy += s_p[j].y; }
        const float inv = 1.0f / (float) cnt, mx = sx * inv, my = sy * inv;
        float sxx = 0.0f, sxy = 0.0f, syy = 0.0f;
        const auto cov = [&](const float2 q) {
          const float dx = q.x - mx, dy = q.y - my;
          sxx = __builtin_fmaf(dx, dx, sxx); sxy = __builtin_fmaf(dx, dy, sxy); syy = __builtin_fmaf(dy, dy, syy);
        };
        for (j = lo; j + 3 <= hi; j += 4) {
          const float2 a = s_p[j], b = s_p[j + 1], c = s_p[j + 2], d = s_p[j + 3];
          cov(a); cov(b); cov(c); cov(d);
        }
        for (; j <= hi; ++j) cov(s_p[j]);
        const float tr = sxx + syy, df = sxx - syy;
        const float disc = __builtin_sqrtf(__builtin_fmaf(df, df, 4.0f * (sxy * sxy)));
        const float lmin = 0.5f * (tr - disc);
        const float v1x = sxy, v1y = lmin - sxx, v2x = lmin - syy, v2y = sxy;
        const float n1 = __builtin_fmaf(v1x, v1x, v1y * v1y), n2 = __builtin_fmaf(v2x, v2x, v2y * v2y);
        float nn = n1; vx = v1x; vy = v1y;
        if (n2 > n1) { vx = v2x; vy = v2y; nn = n2; }
        if (nn > 0.0f) {
          const float s = __builtin_sqrtf(nn);
          vx = vx / s; vy = vy / s;
          if (__builtin_fmaf(vx, pi.x, vy * pi.y) > 0.0f) { vx = -vx; vy = -vy; }
          ok = true;
        }
      }
    }
    LSM2D_PC(1);
    const int pos = block_compact_pos(ok, s_tot, parity, k, tid, kPrepBlock / 64);
    if (ok) { s_q[pos] = pi; s_n[pos] = make_float2(vx, vy); }
  }
  __syncthreads();
  LSM2D_PC(2);
  if (!(A.inv_res > 0.0f)) {                       // no voxelisation: every valid point, beam order
    for (int i = tid; i < k; i += kPrepBlock) { oxy[i] = s_q[i]; onr[i] = s_n[i]; if (oaos) oaos[i] = make_float4(s_q[i].x, s_q[i].y, s_n[i].x, s_n[i].y); }
    if (tid == 0) A.out_count[scan] = k;
    return;
  }
  // ---- F2.3 voxelisation: sort (key, index), average equal-key runs, ascending key order
  const int nv = voxelize_lds<kPrepBlock>(s_q, s_n, s_key, k, A.inv_res, 1.0f, s_tot, tid,
                                          [&](int pos, float x, float y, float nx, float ny) { oxy[pos] = make_float2(x, y); onr[pos] = make_float2(nx, ny); if (oaos) oaos[pos] = make_float4(x, y, nx, ny); });
  if (tid == 0) A.out_count[scan] = nv;
  LSM2D_PC(3);
#ifdef LSM2D_PHASE_CLOCKS
  if (tid == 0) printf("preprocess ticks(10ns): unproject %llu normals %llu compaction %llu voxelise %llu (beams %d valid %d normals %d voxels %d)\n", pc_acc[0], pc_acc[1], pc_acc[2], pc_acc[3], nb, m, k, nv);
#endif
#undef LSM2D_PC
}
__global__ __launch_bounds__(kPrepBlock) void k_preprocess_scans(const PrepArgs A) { preprocess_scan_body<kPrepBlock, kPrepMaxBeams>(A, blockIdx.x); }
static constexpr int kPrepSmallBlock = 512, kPrepSmallBeams = 1152;      // 4 x 8 B x 1152 = 36 KB + the wave totals: beside three k_align workgroups of a CU
__global__ __launch_bounds__(kPrepSmallBlock) void k_preprocess_scans_small(const PrepArgs A) { preprocess_scan_body<kPrepSmallBlock, kPrepSmallBeams>(A, blockIdx.x); }
// several scans, each with its own sensor geometry and its own output set, side by side (the live tracker's front and rear scanner:
// lsm2d_preprocess_scan_into defers its launch, the aligner call that reads both sets queues them together)
static constexpr int kPrepMulti = 4;
struct PrepMultiArgs { PrepArgs a[kPrepMulti]; };
__global__ __launch_bounds__(kPrepBlock) void k_preprocess_multi(const PrepMultiArgs M) { preprocess_scan_body<kPrepBlock, kPrepMaxBeams>(M.a[blockIdx.x], 0); }
