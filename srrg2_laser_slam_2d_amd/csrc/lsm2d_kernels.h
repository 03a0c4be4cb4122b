// lsm2d_kernels.h -- HIP kernels of the scan-matching hot path for gfx950 (CDNA4, wave64).
//
//   k_align            the product kernel: ONE workgroup owns ONE alignment for all max_iterations
//                      (MultiAligner2D::compute restated): per iteration and slice it
//                      streams the moving cloud through an LDS polar z-buffer (64-bit ds_min keys),
//                      walks the two canvases bin by bin (registration/correspondence_finder_projective_2d.cpp:55-74),
//                      accumulates the plane-to-plane factor (octave/solver/nicp_post.m:4-26,69-90) in
//                      registers, reduces with wave shuffles + one LDS hop, and lane 0 solves the 3x3
//                      system and right-updates the pose -- no host round trip, no global atomics.
//   k_find_projective  CorrespondenceFinderProjective2f::compute for one (fixed, moving, pose): ordered pairs.
//   k_project_canvas   PointNormal2fProjectorPolar::compute: source index / depth / transformed point per column.
//   k_linearize_*      SE2Plane2PlaneErrorFactor over a correspondence vector (two-stage deterministic reduce).
//   k_repack_cloud     AoS float4 (x,y,nx,ny) -> split xy / normal arrays with even-aligned cloud starts.
#pragma once
#include "lsm2d_device.h"

namespace lsm2d {

static constexpr int kMaxSlices = 4;
#ifndef LSM2D_ALIGN_BLOCK
#define LSM2D_ALIGN_BLOCK 512
#endif
static constexpr int kAlignBlock = LSM2D_ALIGN_BLOCK;
static constexpr int kFindBlock = 1024;

struct CloudDev {            // device view of a cloud set
  const float2* xy;          // [padded total] coordinates
  const float2* nrm;         // [padded total] normals
  const int32_t* start;      // [n_clouds] first (even) padded index of each cloud
  const int32_t* count;      // [n_clouds] points per cloud
  const int32_t* index;      // [n_alignments] cloud chosen per alignment, or nullptr
  int32_t n_clouds;
};

struct SliceDev {
  CloudDev fixed, moving;
  int32_t finder;
  ProjK   proj;
  float   point_distance, normal_cos, max_distance;
  int32_t cauchy;
  float   tau;
  int32_t min_corr;
  int32_t has_sensor;        // X_eff = S^-1 * X
  float   Sinv[3], cSinv, sSinv;
  int32_t fcan_offset;       // start of this slice's fixed canvas, in cells
};

struct PriorDev { float z_inv[3], cz, sz, omega[9]; };   // Z^-1 and cos/sin of its angle, host-computed
struct StatsDev { int32_t n_corr, n_in, n_out; float chi_in, chi_out; };

struct AlignArgs {
  int32_t n_align, n_slices, max_it, min_inliers;
  float   damping;
  int32_t cols_max, fcan_total;
  const float* init_pose;
  const PriorDev* prior;
  float* out_pose; float* out_H; int32_t* out_status; int32_t* out_its; StatsDev* out_stats;
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

#ifndef LSM2D_ALIGN_MIN_WAVES
#define LSM2D_ALIGN_MIN_WAVES 8      // waves per SIMD the register allocator must leave room for
#endif
__global__ __launch_bounds__(kAlignBlock, LSM2D_ALIGN_MIN_WAVES) void k_align(const AlignArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* mcan = reinterpret_cast<u64*>(smem);
  u64* fcan = mcan + A.cols_max;
  float* red = reinterpret_cast<float*>(fcan + A.fcan_total);     // [nwaves][kAccumWords]
  __shared__ float s_pose[3];
  __shared__ Iso   s_iso[kMaxSlices];
  __shared__ float s_H[9], s_b[3];
  __shared__ int   s_n_in, s_n_out, s_n_corr, s_active, s_done, s_status;
  __shared__ float s_chi_in, s_chi_out;

  const int a = blockIdx.x, tid = threadIdx.x;
  constexpr int nwaves = kAlignBlock / 64;

  // ---- prologue: fixed canvases, camera at identity (correspondence_finder_projective_2d.cpp:37-44)
  for (int i = tid; i < A.fcan_total; i += kAlignBlock) fcan[i] = kEmptyCell;
  if (tid == 0) {
    s_pose[0] = A.init_pose[3 * a + 0]; s_pose[1] = A.init_pose[3 * a + 1]; s_pose[2] = A.init_pose[3 * a + 2];
    s_done = 0; s_status = LSM2D_RUNNING;
    for (int k = 0; k < 9; ++k) s_H[k] = 0.0f;
  }
  __syncthreads();
  const Iso ident = {1.0f, 0.0f, 0.0f, 0.0f};
  for (int s = 0; s < A.n_slices; ++s) {
    const SliceDev& S = A.s[s];
    const int fc = pick_cloud(S.fixed, a);
    project_cloud(S.fixed.xy + S.fixed.start[fc], S.fixed.count[fc], ident, S.proj, fcan + S.fcan_offset, tid, kAlignBlock);
  }
  __syncthreads();

  int it = 0;
  StatsDev last = {0, 0, 0, 0.0f, 0.0f};
  for (; it < A.max_it; ++it) {
    if (tid == 0) {
      // X_eff = S^-1 X per slice (AlignerSliceProcessorLaser2DWithSensor), then cos/sin once per slice
      for (int s = 0; s < A.n_slices; ++s) {
        float Xe[3] = {s_pose[0], s_pose[1], s_pose[2]};
        if (A.s[s].has_sensor) compose(A.s[s].cSinv, A.s[s].sSinv, A.s[s].Sinv, s_pose, Xe);
        s_iso[s].c = cosf(Xe[2]); s_iso[s].s = sinf(Xe[2]); s_iso[s].tx = Xe[0]; s_iso[s].ty = Xe[1];
      }
      for (int k = 0; k < 9; ++k) s_H[k] = 0.0f;
      s_b[0] = s_b[1] = s_b[2] = 0.0f;
      s_n_in = s_n_out = s_n_corr = s_active = 0; s_chi_in = s_chi_out = 0.0f;
    }
    for (int s = 0; s < A.n_slices; ++s) {
      const SliceDev& S = A.s[s];
      for (int i = tid; i < S.proj.cols; i += kAlignBlock) mcan[i] = kEmptyCell;
      __syncthreads();
      const Iso T = s_iso[s];
      const int fc = pick_cloud(S.fixed, a), mc = pick_cloud(S.moving, a);
      const int mbase = S.moving.start[mc], fbase = S.fixed.start[fc];
      // HOT: every moving point, every iteration (correspondence_finder_projective_2d.cpp:47-48)
      project_cloud(S.moving.xy + mbase, S.moving.count[mc], T, S.proj, mcan, tid, kAlignBlock);
      __syncthreads();
      Accum acc; accum_zero(acc);
      const u64* fcs = fcan + S.fcan_offset;
      const float2* fn = S.fixed.nrm + fbase; const float2* mn = S.moving.nrm + mbase;
      const float2* fp = S.fixed.xy + fbase;  const float2* mp = S.moving.xy + mbase;
      for (int col = tid; col < S.proj.cols; col += kAlignBlock) {
        int fi, mi; float2 nf, nm;
        if (match_bin(fcs[col], mcan[col], S, T, fn, mn, fi, mi, nf, nm))
          accumulate_pair(T, fp[fi], nf, mp[mi], nm, S.cauchy != 0, S.tau, acc);
      }
      block_reduce_store(acc, red, tid);
      __syncthreads();
      if (tid == 0) {
        Accum t; block_reduce_gather(red, nwaves, t);
        s_n_corr += t.n_corr;
        if (t.n_corr > S.min_corr) {   // slices with #pairs <= min_num_correspondences are skipped
          ++s_active;
          s_H[0] += t.h00; s_H[1] += t.h01; s_H[2] += t.h02; s_H[3] += t.h01; s_H[4] += t.h11; s_H[5] += t.h12;
          s_H[6] += t.h02; s_H[7] += t.h12; s_H[8] += t.h22;
          s_b[0] += t.b0; s_b[1] += t.b1; s_b[2] += t.b2;
          s_n_in += t.n_in; s_n_out += t.n_out; s_chi_in += t.chi_in; s_chi_out += t.chi_out;
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
      last.n_corr = s_n_corr; last.n_in = s_n_in; last.n_out = s_n_out; last.chi_in = s_chi_in; last.chi_out = s_chi_out;
      if (A.out_stats) A.out_stats[(size_t) a * A.max_it + it] = last;
      if (!s_active) { s_status = LSM2D_NOT_ENOUGH_CORRESPONDENCES; s_done = 1; }
      else {
        float H[9], b[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) H[k] = s_H[k];
        b[0] = s_b[0]; b[1] = s_b[1]; b[2] = s_b[2];
        if (A.prior) {
          // SE2 prior: e = t2v(Z^-1 X), J = blkdiag(R_e, 1) for the right perturbation
          const PriorDev& Pz = A.prior[a];
          float E[3]; compose(Pz.cz, Pz.sz, Pz.z_inv, s_pose, E);
          const float c = cosf(E[2]), s_ = sinf(E[2]);
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
              H[3 * r + cc] += v;
            }
            float v = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) v += Jp[3 * k + r] * Oe[k];
            b[r] += v;
          }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) s_H[k] = H[k];     // information matrix = H of the last iteration
        float X[3] = {s_pose[0], s_pose[1], s_pose[2]};
        if (!solve_update(H, b, A.damping, X)) { s_status = LSM2D_SINGULAR_H; s_done = 1; }
        else { s_pose[0] = X[0]; s_pose[1] = X[1]; s_pose[2] = X[2]; }
      }
    }
    __syncthreads();
    if (s_done) { ++it; break; }
  }
  if (tid == 0) {
    int st = s_status;
    if (st == LSM2D_RUNNING) st = (A.max_it > 0 && last.n_in < A.min_inliers) ? LSM2D_NOT_ENOUGH_INLIERS : LSM2D_SUCCESS;
    A.out_status[a] = st;
    A.out_pose[3 * a + 0] = s_pose[0]; A.out_pose[3 * a + 1] = s_pose[1]; A.out_pose[3 * a + 2] = s_pose[2];
    if (A.out_H) for (int k = 0; k < 9; ++k) A.out_H[9 * a + k] = s_H[k];
    if (A.out_its) A.out_its[a] = it;
  }
}

// ---- finder-level: one (fixed, moving, pose) -> pairs in ascending column ------------------------
struct FindArgs {
  CloudDev fixed, moving; int32_t fc, mc;
  ProjK proj; float point_distance, normal_cos;
  Iso T;
  int32_t* out_pairs;  // [cols][2]
  int32_t* out_count;
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
  project_cloud(A.fixed.xy + fbase, A.fixed.count[A.fc], ident, A.proj, fcan, tid, kFindBlock);
  project_cloud(A.moving.xy + mbase, A.moving.count[A.mc], A.T, A.proj, mcan, tid, kFindBlock);
  __syncthreads();
  SliceDev S; S.point_distance = A.point_distance; S.normal_cos = A.normal_cos;
  const int lane = tid & 63, wave = tid >> 6;
  for (int c0 = 0; c0 < A.proj.cols; c0 += kFindBlock) {
    const int col = c0 + tid;
    int fi = -1, mi = -1; float2 nf, nm; bool ok = false;
    if (col < A.proj.cols) ok = match_bin(fcan[col], mcan[col], S, A.T, A.fixed.nrm + fbase, A.moving.nrm + mbase, fi, mi, nf, nm);
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
};

__global__ __launch_bounds__(256) void k_linearize_partial(const LinArgs A) {
  __shared__ float red[4 * kAccumWords];
  const int tid = threadIdx.x;
  const int fbase = A.fixed.start[A.fc], mbase = A.moving.start[A.mc];
  Accum acc; accum_zero(acc);
  for (int k = blockIdx.x * 256 + tid; k < A.n_pairs; k += gridDim.x * 256) {
    const int fi = A.pairs[2 * k], mi = A.pairs[2 * k + 1];
    accumulate_pair(A.T, A.fixed.xy[fbase + fi], A.fixed.nrm[fbase + fi], A.moving.xy[mbase + mi], A.moving.nrm[mbase + mi],
                    A.cauchy != 0, A.tau, acc);
  }
  block_reduce_store(acc, red, tid);
  __syncthreads();
  if (tid == 0) {
    Accum t; block_reduce_gather(red, 4, t);
    float* p = A.partial + (size_t) blockIdx.x * kAccumWords;
    p[0] = t.h00; p[1] = t.h01; p[2] = t.h02; p[3] = t.h11; p[4] = t.h12; p[5] = t.h22; p[6] = t.b0; p[7] = t.b1; p[8] = t.b2;
    p[9] = t.chi_in; p[10] = t.chi_out; p[11] = __int_as_float(t.n_in); p[12] = __int_as_float(t.n_out); p[13] = __int_as_float(t.n_corr);
  }
}

__global__ void k_linearize_final(const float* partial, int n_blocks, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Accum t; block_reduce_gather(partial, n_blocks, t);     // fixed block order => deterministic
  out[0] = t.h00; out[1] = t.h01; out[2] = t.h02; out[3] = t.h11; out[4] = t.h12; out[5] = t.h22; out[6] = t.b0; out[7] = t.b1; out[8] = t.b2;
  out[9] = t.chi_in; out[10] = t.chi_out; out[11] = __int_as_float(t.n_in); out[12] = __int_as_float(t.n_out); out[13] = __int_as_float(t.n_corr);
}

// ---- cloud repack: AoS float4 -> xy / normal arrays, cloud c starting at padded index pstart[c] ----
__global__ void k_repack_cloud(const float4* __restrict__ src, const int32_t* __restrict__ offsets, const int32_t* __restrict__ pstart,
                               int n_clouds, long long total, float2* __restrict__ xy, float2* __restrict__ nrm) {
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < total; i += (long long) gridDim.x * blockDim.x) {
    int lo = 0, hi = n_clouds - 1;               // last cloud with offsets[c] <= i
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((long long) offsets[mid] <= i) lo = mid; else hi = mid - 1; }
    const long long d = (long long) pstart[lo] + (i - offsets[lo]);
    const float4 v = src[i];
    xy[d] = make_float2(v.x, v.y); nrm[d] = make_float2(v.z, v.w);
  }
}

}  // namespace lsm2d
