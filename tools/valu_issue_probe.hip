// valu_issue_probe.hip -- what "VALU-issue bound" means on this chip (gfx950), measured: cycles per wave64 VALU instruction on one SIMD
// as a function of (i) how many waves share the SIMD, (ii) whether each wave's instructions depend on one another (the z-buffer stream of
// k_align is ONE dependent chain per point), (iii) the instruction kind (plain v_fma_f32, v_fmaak_f32 with a 32-bit literal, v_rcp_f32,
// v_cmp + v_cndmask through vcc).  bench.py's roofline prices k_align against the saturated rate this prints.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -Isrrg2_laser_slam_2d_amd/csrc -Iinclude -o /tmp/valu_probe tools/valu_issue_probe.hip && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND>
__global__ __launch_bounds__(1024) void probe(unsigned long long* out, int iters, float seed) {
  extern __shared__ char pad[];      // the whole LDS: one workgroup per CU
  float a = seed + threadIdx.x, b = 1.0001f, c = 0.5f, d = seed * 2.0f, e = seed * 3.0f, f = seed * 5.0f;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f pa = {a, d}, pb = {b, b}, pc = {c, c};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));                       // 64 dependent
    if (KIND == 1) asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                                : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));                                  // 4 independent chains
    if (KIND == 2) asm volatile(REP64("v_fmaak_f32 %0, %0, %1, 0x3d286f0c\n") : "+v"(a) : "v"(b));                      // literal, dependent
    if (KIND == 3) asm volatile(REP16("v_rcp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n")
                                : "+v"(a) : "v"(b), "v"(c));                                                             // 1 transcendental in 4, dependent
    if (KIND == 4) asm volatile(REP16("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n")
                                : "+v"(a) : "v"(b), "v"(c) : "vcc");                                                     // compare + select through vcc
    if (KIND == 5) asm volatile(REP16("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n")
                                : "+v"(a), "+v"(d) : "v"(b), "v"(c));                                                    // 2 interleaved chains
    if (KIND == 6) asm volatile(REP16("v_rcp_f32 %1, %0\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %0, %1, %3, %4\n")
                                : "+v"(a), "+v"(d), "+v"(e) : "v"(b), "v"(c));                                          // rcp, 3 unrelated fma, then its consumer
    if (KIND == 7) asm volatile(REP64("v_rcp_f32 %0, %0\n") : "+v"(a));                                                 // rcp chain
    if (KIND == 8) asm volatile(REP16("v_rcp_f32 %1, %2\n v_fma_f32 %0, %0, %3, %4\n v_fma_f32 %0, %0, %3, %4\n v_fma_f32 %0, %0, %3, %4\n")
                                : "+v"(a), "+v"(d) : "v"(e), "v"(b), "v"(c));                                            // rcp whose result nobody waits for
    if (KIND == 9) asm volatile(REP16("v_rcp_f32 %1, %0\n s_nop 0\n v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n")
                                : "+v"(a), "+v"(d) : "v"(b), "v"(c));                                                    // as the compiler emits it: s_nop 0 behind the rcp
    if (KIND == 10) asm volatile(REP16("v_rcp_f32 %1, %0\n v_rsq_f32 %2, %0\n v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %0, %0, %3, %4\n")
                                : "+v"(a), "+v"(d), "+v"(e) : "v"(b), "v"(c));                                           // two transcendentals back to back
    if (KIND == 11) asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(pa) : "v"(pb), "v"(pc));                  // packed: two fp32 FMAs per lane per instruction
    if (KIND == 12) asm volatile(REP64("v_pk_mul_f32 %0, %0, %1\n") : "+v"(pa) : "v"(pb));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (a + d + e + f + pa.x + pa.y == 12345.678f) out[0] = 0;
}

// the real per-point instruction stream of k_align (csrc/lsm2d_device.h: project_point_stream, incl. its exec masking and the
// fire-and-forget ds_min_u64) on register-resident points: no global loads, no barriers, no bin walk, no solve -- what the 1024 SIMDs
// can do with THIS instruction mix when nothing else is in the way.  4 workgroups of 512 threads per CU, as k_align runs.
#include "lsm2d_device.h"
// ablations of the same stream: kAtomic = 0 drops the ds_min_u64 (the key is folded into a register instead), kGates = 0 drops the two
// exec-masked regions (range gate, column check: straight-line code, the column clamped, the update always issued), kTrans = 0
// replaces v_rsq_f32 by a plain multiply (wrong values, same dependencies).  Differences against the full stream price each part.
template <int kAtomic, int kGates, int kTrans>
__device__ __forceinline__ void point_ablation(const lsm2d::Iso& T, const lsm2d::ProjK& P, float px, float py, int idx, unsigned long long* canvas, unsigned long long& sink) {
  using namespace lsm2d;
  float qx, qy; xf_point(T, px, py, qx, qy);
  const float r2 = __builtin_fmaf(qx, qx, qy * qy);
  if (!kGates || (r2 >= P.r2lo && r2 <= P.r2hi)) {
    const float y0 = kTrans ? __builtin_amdgcn_rsqf(r2) : r2 * 0.11f;
    const float s0 = r2 * y0, h = 0.5f * y0, e = __builtin_fmaf(-s0, s0, r2), r = __builtin_fmaf(e, h, s0);
    const float ax = __builtin_fabsf(qx), ay = __builtin_fabsf(qy);
    const bool swap = ay > ax;
    const float mn = swap ? ax : ay;
    const float e0 = __builtin_fmaf(-r, y0, 1.0f), y1 = __builtin_fmaf(e0, y0, y0), q0 = mn * y1, e1 = __builtin_fmaf(-r, q0, mn);
    const float t = __builtin_fmaf(e1, y1, q0), s = t * t;
    float p = 1.237212196e-01f;
    p = __builtin_fmaf(p, s, -1.153038889e-01f); p = __builtin_fmaf(p, s, 9.340071678e-02f); p = __builtin_fmaf(p, s, 1.043075230e-02f);
    p = __builtin_fmaf(p, s, 4.762428626e-02f); p = __builtin_fmaf(p, s, 7.478348911e-02f); p = __builtin_fmaf(p, s, 1.666723490e-01f);
    float phi = __builtin_fmaf(t * s, p, t);
    if (swap) phi = 1.57079637050628662f - phi;
    if (qx < 0.0f) phi = 3.14159274101257324f - phi;
    const float u = __builtin_fmaf(P.K00, __builtin_copysignf(phi, qy), P.K01);
    int col; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(col) : "v"(u));
    if (!kGates) col = (int) min((unsigned) col, (unsigned) P.cols - 1u);
    if (!kGates || (unsigned) col < (unsigned) P.cols) {
      const unsigned long long key = ((unsigned long long) __float_as_uint(r) << 32) | (unsigned long long) (unsigned) idx;
      if (kAtomic) atomicMin(&canvas[col], key); else sink ^= key + (unsigned long long) (unsigned) col;
    }
  }
}
template <int kAtomic, int kGates, int kTrans>
__global__ __launch_bounds__(512, 8) void probe_ablation(unsigned long long* out, int iters, float seed, int cols) {
  extern __shared__ unsigned long long canvas[];
  using namespace lsm2d;
  for (int i = threadIdx.x; i < cols; i += 512) canvas[i] = kEmptyCell;
  __syncthreads();
  ProjK P; P.cols = cols; P.K00 = (float) cols / 6.28318548f; P.K01 = 0.5f * (float) cols; P.r2lo = 0.09f; P.r2hi = 900.0f; P.rmin = 0.3f; P.rmax = 30.f;
  P.colsf = (float) cols; P.tiny_ok = 1;
  asm volatile("" : "+v"(P.K01));
  Iso T; T.c = 0.8f; T.s = 0.6f; T.tx = seed; T.ty = -seed;
  const float a = 0.37f * (float) (threadIdx.x * 131 % 509), b = 0.23f * (float) (blockIdx.x % 97);
  float4 v0 = make_float4(10.f * __cosf(a) + 0.01f * b, 8.f * __sinf(a), 12.f * __cosf(a + 1.f), 9.f * __sinf(a + 1.f) - 0.01f * b);
  float4 v1 = make_float4(7.f * __cosf(a + 2.f), 11.f * __sinf(a + 2.f) + 0.02f * b, 14.f * __cosf(a + 3.f) - 0.02f * b, 6.f * __sinf(a + 3.f));
  unsigned long long sink = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    point_ablation<kAtomic, kGates, kTrans>(T, P, v0.x, v0.y, 4 * i, canvas, sink);
    point_ablation<kAtomic, kGates, kTrans>(T, P, v0.z, v0.w, 4 * i + 1, canvas, sink);
    point_ablation<kAtomic, kGates, kTrans>(T, P, v1.x, v1.y, 4 * i + 2, canvas, sink);
    point_ablation<kAtomic, kGates, kTrans>(T, P, v1.z, v1.w, 4 * i + 3, canvas, sink);
    T.tx += 1e-4f; T.ty -= 1e-4f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (sink == 0x1234567ull) out[0] = sink;
}
template <int kAtomic, int kGates, int kTrans>
static void run_ablation(const char* name, unsigned long long* d_out, int n_cu) {
  const int iters = 4000, cols = 1081, nb = 4 * n_cu, lds = 36 * 1024;
  hipFuncSetAttribute((const void*) probe_ablation<kAtomic, kGates, kTrans>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipMemset(d_out, 0, sizeof(unsigned long long) * 16 * nb);
  probe_ablation<kAtomic, kGates, kTrans><<<nb, 512, lds>>>(d_out, iters, 0.25f, cols);
  probe_ablation<kAtomic, kGates, kTrans><<<nb, 512, lds>>>(d_out, iters, 0.25f, cols);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(16 * nb);
  hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
  std::vector<double> per;
  for (int b = 0; b < nb; ++b) for (int k = 0; k < 8; ++k) per.push_back((double) h[b * 16 + k]);
  std::sort(per.begin(), per.end());
  printf("  %-58s %.1f cycles of a SIMD per point of a wave\n", name, per[per.size() * 99 / 100] / ((double) iters * 4.0 * 8.0));
}

__global__ __launch_bounds__(512, 8) void probe_stream(unsigned long long* out, int iters, float seed, int cols) {
  extern __shared__ unsigned long long canvas[];
  using namespace lsm2d;
  for (int i = threadIdx.x; i < cols; i += 512) canvas[i] = kEmptyCell;
  __syncthreads();
  ProjK P; P.cols = cols; P.K00 = (float) cols / 6.28318548f; P.K01 = 0.5f * (float) cols; P.r2lo = 0.09f; P.r2hi = 900.0f; P.rmin = 0.3f; P.rmax = 30.f;
  P.colsf = (float) cols; P.tiny_ok = 1;
  asm volatile("" : "+v"(P.K01));
  Iso T; T.c = 0.8f; T.s = 0.6f; T.tx = seed; T.ty = -seed;
  // four points per thread, spread over the room like the lanes of a lane-chunked wave (decorrelated columns)
  const float a = 0.37f * (float) (threadIdx.x * 131 % 509), b = 0.23f * (float) (blockIdx.x % 97);
  float4 v0 = make_float4(10.f * __cosf(a) + 0.01f * b, 8.f * __sinf(a), 12.f * __cosf(a + 1.f), 9.f * __sinf(a + 1.f) - 0.01f * b);
  float4 v1 = make_float4(7.f * __cosf(a + 2.f), 11.f * __sinf(a + 2.f) + 0.02f * b, 14.f * __cosf(a + 3.f) - 0.02f * b, 6.f * __sinf(a + 3.f));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    project_point_stream<false>(T, P, v0.x, v0.y, 4 * i, canvas);
    project_point_stream<false>(T, P, v0.z, v0.w, 4 * i + 1, canvas);
    project_point_stream<false>(T, P, v1.x, v1.y, 4 * i + 2, canvas);
    project_point_stream<false>(T, P, v1.z, v1.w, 4 * i + 3, canvas);
    T.tx += 1e-4f; T.ty -= 1e-4f;                        // the points move: nothing is loop-invariant
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
static void run_stream(unsigned long long* d_out, int n_cu) {
  const int iters = 4000, cols = 1081, nb = 4 * n_cu, lds = 36 * 1024;        // 36 KB per workgroup: four per CU, like k_align
  hipFuncSetAttribute((const void*) probe_stream, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipMemset(d_out, 0, sizeof(unsigned long long) * 16 * nb);
  probe_stream<<<nb, 512, lds>>>(d_out, iters, 0.25f, cols);
  probe_stream<<<nb, 512, lds>>>(d_out, iters, 0.25f, cols);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(16 * nb);
  hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
  std::vector<double> per;
  for (int b = 0; b < nb; ++b) for (int k = 0; k < 8; ++k) per.push_back((double) h[b * 16 + k]);
  std::sort(per.begin(), per.end());
  const double mx = per[per.size() * 99 / 100], mn = per[per.size() / 100];
  // a SIMD ran 8 waves x iters x 4 points
  printf("k_align's point stream (project_point_stream, 8 waves per SIMD): %.1f cycles of a SIMD per point of a wave (waves %.0f..%.0f kcyc)\n",
         mx / ((double) iters * 4.0 * 8.0), mn * 1e-3, mx * 1e-3);
}

template <int KIND>
static void run(const char* name, unsigned long long* d_out, int n_cu) {
  const int iters = 2000;
  printf("%-44s", name);
  // {threads per workgroup, workgroups per CU}: 1-4 waves per SIMD from one workgroup, then 8 per SIMD as 2 x 1024 and as 4 x 512 (k_align's shape)
  const int cfg[6][2] = {{256, 1}, {512, 1}, {768, 1}, {1024, 1}, {1024, 2}, {512, 4}};
  for (int c = 0; c < 6; ++c) {
    const int threads = cfg[c][0], per_cu = cfg[c][1], w = threads / 256 * per_cu, nb = n_cu * per_cu;
    const int lds = 160 * 1024 / per_cu - 1024;
    hipFuncSetAttribute((const void*) probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipMemset(d_out, 0, sizeof(unsigned long long) * 16 * nb);
    probe<KIND><<<nb, threads, lds>>>(d_out, iters, 1.0f);
    probe<KIND><<<nb, threads, lds>>>(d_out, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(16 * nb);
    hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    std::vector<double> per;
    for (int b = 0; b < nb; ++b) for (int k = 0; k < threads / 64; ++k) per.push_back((double) h[b * 16 + k]);
    std::sort(per.begin(), per.end());
    const double med = per[per.size() / 2], mx = per[per.size() * 99 / 100], mn = per[per.size() / 100];
    // a SIMD ran w waves x iters x 64 instructions; with every wave resident from the start the slowest wave's time is the SIMD's time
    printf(" | %dx%d: %.2f cyc/inst/SIMD (waves %.0f..%.0f..%.0f kcyc)", per_cu, threads, mx / ((double) iters * 64.0 * w), mn * 1e-3, med * 1e-3, mx * 1e-3);
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  unsigned long long* d_out; hipMalloc(&d_out, sizeof(unsigned long long) * 16 * n_cu * 4);
  printf("%s, %d CUs; one workgroup per CU, w waves per SIMD; cycles = s_memtime ticks\n", p.gcnArchName, n_cu);
  run<0>("v_fma_f32, one dependent chain", d_out, n_cu);
  run<5>("v_fma_f32, two interleaved chains", d_out, n_cu);
  run<1>("v_fma_f32, four interleaved chains", d_out, n_cu);
  run<2>("v_fmaak_f32 (32-bit literal), dependent", d_out, n_cu);
  run<3>("1 v_rcp_f32 + 3 v_fma_f32, dependent", d_out, n_cu);
  run<4>("v_cmp + v_cndmask(vcc) + 2 v_fma, dependent", d_out, n_cu);
  run<7>("v_rcp_f32 chain (x64 counted as 64 inst)", d_out, n_cu);
  run<8>("1 v_rcp (result unused) + 3 fma", d_out, n_cu);
  run<9>("1 v_rcp + s_nop 0 + 3 dependent fma (4 VALU counted as 4)", d_out, n_cu);
  run<6>("v_rcp, 3 unrelated fma, consumer (5 per group, counted as 4)", d_out, n_cu);
  run<10>("v_rcp + v_rsq + 2 fma", d_out, n_cu);
  run<11>("v_pk_fma_f32 (2 FMAs per lane), dependent", d_out, n_cu);
  run<12>("v_pk_mul_f32, dependent", d_out, n_cu);
  run_stream(d_out, n_cu);
  printf("ablations of that stream (same launch shape):\n");
  run_ablation<1, 1, 1>("full stream (copy of project_point_stream)", d_out, n_cu);
  run_ablation<0, 1, 1>("without the ds_min_u64", d_out, n_cu);
  run_ablation<1, 0, 1>("without the exec-masked gates (straight-line)", d_out, n_cu);
  run_ablation<1, 1, 0>("v_rsq_f32 replaced by a plain multiply", d_out, n_cu);
  run_ablation<0, 0, 0>("arithmetic only: no atomic, no gates, no transcendentals", d_out, n_cu);
  return 0;
}
