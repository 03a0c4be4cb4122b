// Exhaustive check, on the GPU, of the short correctly-rounded fp32 sequences the kernels use in place of the
// compiler's general divide / sqrt (csrc/lsm2d_device.h: div_rn_unit, sqrt_rn_normal).  The CPU oracle uses the
// plain IEEE '/' and sqrtf, so bit parity of columns and depths needs these sequences to be EXACT on this hardware's
// v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 for every input the projector's range gate lets through.
//
//   division  n/d, 0 < n <= d: every step scales exactly with powers of two (no subnormals inside the gate), so all
//             2^23 x 2^23 mantissa pairs cover every admissible input; n = 1.m_n (or half of it when m_n > m_d).
//   sincos    device vs host evaluation of the fixed sin / cos sequence (bitwise) on 6.5e6 arguments, and its error vs libm.
//   sqrt      every fp32 bit pattern in [1e-30, FLT_MAX] (the gate's r2 lies in [1e-30, 1e36]; the lane-chunked stream also
//             forms depths of points beyond range_max, which only have to stay above it).
//
// Column 0 of each table is the PRODUCTION function (lsm2d::div_rn_unit / lsm2d::sqrt_rn_normal, included from
// csrc/lsm2d_device.h); the others are the longer sequence it replaced and the shorter ones that turn out not to be exact.
//
// build + run:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -Isrrg2_laser_slam_2d_amd/csrc -Iinclude \
//                     -o /tmp/fp_exact_check tools/fp_exact_check.hip
//               /tmp/fp_exact_check [div_rows]      (div_rows: how many m_d values to cover, default all 2^23; exit code 1 when
//                                                    a production function mismatches)
#include <hip/hip_runtime.h>
#include "lsm2d_device.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define NVAR 4
struct Report { unsigned long long fails[NVAR]; uint32_t first_n[NVAR], first_d[NVAR]; };

__device__ __forceinline__ float f_from(uint32_t b) { return __uint_as_float(b); }

// S7: the sequence hipcc emits for '/', minus v_div_scale / v_div_fixup (two operations longer than production)
__device__ __forceinline__ float div_s7(float n, float d, float r0) {
  const float e0 = __builtin_fmaf(-d, r0, 1.0f), r1 = __builtin_fmaf(e0, r0, r0);
  const float q0 = n * r1, e1 = __builtin_fmaf(-d, q0, n), q1 = __builtin_fmaf(e1, r1, q0);
  const float e2 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(e2, r1, q1);
}
// S5: raw reciprocal, two quotient corrections
__device__ __forceinline__ float div_s5(float n, float d, float r0) {
  const float q0 = n * r0, e0 = __builtin_fmaf(-d, q0, n), q1 = __builtin_fmaf(e0, r0, q0);
  const float e1 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(e1, r0, q1);
}
// S3: raw reciprocal, one quotient correction
__device__ __forceinline__ float div_s3(float n, float d, float r0) {
  const float q0 = n * r0, e0 = __builtin_fmaf(-d, q0, n);
  return __builtin_fmaf(e0, r0, q0);
}

__global__ void k_div(uint32_t md0, Report* rep) {
  const uint32_t md = md0 + blockIdx.y;
  const float d = f_from(0x3f800000u | md);
  const float r0 = __builtin_amdgcn_rcpf(d);
  unsigned long long bad[NVAR] = {0, 0, 0, 0};
  uint32_t firstn[NVAR] = {0, 0, 0, 0};
  for (uint32_t mn = blockIdx.x * blockDim.x + threadIdx.x; mn < (1u << 23); mn += gridDim.x * blockDim.x) {
    const float n = f_from((mn <= md ? 0x3f800000u : 0x3f000000u) | mn);
    const float t = n / d;                                   // hipcc default: correctly rounded
    const float v[NVAR] = {lsm2d::div_rn_unit(n, d), div_s7(n, d, r0), div_s5(n, d, r0), div_s3(n, d, r0)};
#pragma unroll
    for (int k = 0; k < NVAR; ++k)
      if (__float_as_uint(v[k]) != __float_as_uint(t)) { if (!bad[k]) firstn[k] = __float_as_uint(n); ++bad[k]; }
  }
#pragma unroll
  for (int k = 0; k < NVAR; ++k)
    if (bad[k]) { if (atomicAdd(&rep->fails[k], bad[k]) == 0) { rep->first_n[k] = firstn[k]; rep->first_d[k] = __float_as_uint(d); } }
}

// v_sqrt_f32 + two-sided residual test: what hipcc emits for sqrtf() minus its denormal pre-scaling
__device__ __forceinline__ float sqrt_two_sided(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
  const float em = __builtin_fmaf(-sm, s, x), ep = __builtin_fmaf(-sp, s, x);
  float r = em <= 0.0f ? sm : s;
  r = ep > 0.0f ? sp : r;
  return r;
}
// R8: a second residual correction
__device__ __forceinline__ float sqrt_r8(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y, h = 0.5f * y;
  const float e = __builtin_fmaf(-s0, s0, x), s1 = __builtin_fmaf(e, h, s0);
  const float e2 = __builtin_fmaf(-s1, s1, x);
  return __builtin_fmaf(e2, h, s1);
}
// Q6: v_sqrt_f32 + v_rcp-free correction  s + e * (0.5/s) with 0.5/s from v_rsq
__device__ __forceinline__ float sqrt_q6(float x) {
  const float s0 = __builtin_amdgcn_sqrtf(x), h = 0.5f * __builtin_amdgcn_rsqf(x);
  const float e = __builtin_fmaf(-s0, s0, x);
  return __builtin_fmaf(e, h, s0);
}

__global__ void k_sqrt(uint32_t lo, uint32_t hi, Report* rep) {
  unsigned long long bad[NVAR] = {0, 0, 0, 0};
  uint32_t firstx[NVAR] = {0, 0, 0, 0};
  for (unsigned long long b = lo + (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += (unsigned long long) gridDim.x * blockDim.x) {
    const float x = f_from((uint32_t) b);
    const float t = sqrtf(x);                                // hipcc default: correctly rounded
    const float v[NVAR] = {lsm2d::sqrt_rn_normal(x), sqrt_two_sided(x), sqrt_r8(x), sqrt_q6(x)};
#pragma unroll
    for (int k = 0; k < NVAR; ++k)
      if (__float_as_uint(v[k]) != __float_as_uint(t)) { if (!bad[k]) firstx[k] = (uint32_t) b; ++bad[k]; }
  }
#pragma unroll
  for (int k = 0; k < NVAR; ++k)
    if (bad[k]) { if (atomicAdd(&rep->fails[k], bad[k]) == 0) rep->first_n[k] = firstx[k]; }
}

__global__ void k_sincos(const float* x, float2* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
    float s, c; lsm2d::sincos_fixed(x[i], s, c);
    out[i] = make_float2(s, c);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
  const uint32_t rows = argc > 1 ? (uint32_t) strtoul(argv[1], nullptr, 0) : (1u << 23);
  Report* d_rep; Report h; unsigned long long sincos_fails = 0;
  CK(hipMalloc(&d_rep, sizeof(Report)));

  // ---- sqrt
  CK(hipMemset(d_rep, 0, sizeof(Report)));
  float flo = 1e-30f, fhi = 3.402823466e+38f; uint32_t lo, hi; memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
  hipLaunchKernelGGL(k_sqrt, dim3(8192), dim3(256), 0, 0, lo, hi, d_rep);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(&h, d_rep, sizeof(h), hipMemcpyDeviceToHost));
  const char* sn[NVAR] = {"sqrt_rn_normal", "two_sided(v_sqrt)", "R8(v_rsq,2 corr)", "Q6(v_sqrt+v_rsq)"};
  printf("sqrt: %llu inputs, bit patterns [0x%08x, 0x%08x]\n", (unsigned long long) hi - lo + 1, lo, hi);
  for (int k = 0; k < NVAR; ++k) printf("  %-20s mismatches vs sqrtf: %llu  first x=0x%08x\n", sn[k], h.fails[k], h.first_n[k]);
  const unsigned long long sqrt_prod_fails = h.fails[0];
  fflush(stdout);

  // ---- sincos_fixed: the kernels' rotation of a pose must have the bits the host code (and the CPU oracle, which evaluates the
  // same fmaf sequence) gives it.  Every 97th fp32 bit pattern with |x| in [2^-20, 64]: ~6.5e6 arguments of either sign.
  {
    std::vector<float> xs;
    float a = 9.5367431640625e-07f, b = 64.0f; uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
    for (uint32_t u = ua; u <= ub; u += 97) { float v; memcpy(&v, &u, 4); xs.push_back(v); xs.push_back(-v); }
    const size_t n = xs.size();
    float* d_x; float2* d_sc;
    CK(hipMalloc(&d_x, n * sizeof(float))); CK(hipMalloc(&d_sc, n * sizeof(float2)));
    CK(hipMemcpy(d_x, xs.data(), n * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sincos, dim3(4096), dim3(256), 0, 0, d_x, d_sc, n);
    CK(hipDeviceSynchronize());
    std::vector<float2> sc(n);
    CK(hipMemcpy(sc.data(), d_sc, n * sizeof(float2), hipMemcpyDeviceToHost));
    unsigned long long bad = 0; double worst = 0.0;
    for (size_t i = 0; i < n; ++i) {
      float s, c; lsm2d::sincos_fixed(xs[i], s, c);
      if (memcmp(&s, &sc[i].x, 4) || memcmp(&c, &sc[i].y, 4)) ++bad;
      const double es = fabs((double) s - sin((double) xs[i])), ec = fabs((double) c - cos((double) xs[i]));
      if (es > worst) worst = es; if (ec > worst) worst = ec;
    }
    printf("sincos: %zu arguments, |x| in [2^-20, 64]\n  %-20s device vs host bit mismatches: %llu  max abs error vs libm (double): %.3e\n", n, "sincos_fixed", bad, worst);
    sincos_fails = bad + (worst > 1.5e-7 ? 1 : 0);
    CK(hipFree(d_x)); CK(hipFree(d_sc));
    fflush(stdout);
  }

  // ---- division: rows of m_d, every m_n
  CK(hipMemset(d_rep, 0, sizeof(Report)));
  const uint32_t rows_per_launch = 2048;
  unsigned long long pairs = 0;
  // stride over the m_d range so a partial run still samples all of it
  const uint32_t total_launches = (rows + rows_per_launch - 1) / rows_per_launch;
  for (uint32_t l = 0; l < total_launches; ++l) {
    const uint32_t md0 = rows == (1u << 23) ? l * rows_per_launch : (uint32_t) (((unsigned long long) l * ((1u << 23) - rows_per_launch)) / (total_launches > 1 ? total_launches - 1 : 1));
    hipLaunchKernelGGL(k_div, dim3(64, rows_per_launch), dim3(256), 0, 0, md0, d_rep);
    pairs += (unsigned long long) rows_per_launch << 23;
    if ((l & 127) == 127 || l + 1 == total_launches) {
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(&h, d_rep, sizeof(h), hipMemcpyDeviceToHost));
      printf("div: %u/%u launches, %.3e pairs, mismatches div_rn_unit=%llu S7=%llu S5=%llu S3=%llu\n", l + 1, total_launches, (double) pairs,
             h.fails[0], h.fails[1], h.fails[2], h.fails[3]);
      fflush(stdout);
    }
  }
  const char* dn[NVAR] = {"div_rn_unit", "S7(2 corrections)", "S5(raw rcp,2 corr)", "S3(raw rcp,1 corr)"};
  printf("division: %.4e (n,d) mantissa pairs\n", (double) pairs);
  for (int k = 0; k < NVAR; ++k) printf("  %-20s mismatches vs n/d: %llu  first n=0x%08x d=0x%08x\n", dn[k], h.fails[k], h.first_n[k], h.first_d[k]);
  CK(hipFree(d_rep));
  return (h.fails[0] || sqrt_prod_fails || sincos_fails) ? 1 : 0;
}
