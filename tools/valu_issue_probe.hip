// valu_issue_probe.hip -- what "VALU-issue bound" means on this chip (gfx950), measured: cycles per wave64 VALU instruction on one SIMD
// as a function of (i) how many waves share the SIMD, (ii) whether each wave's instructions depend on one another (the z-buffer stream of
// k_align is ONE dependent chain per point), (iii) the instruction kind (plain v_fma_f32, v_fmaak_f32 with a 32-bit literal, v_rcp_f32,
// v_cmp + v_cndmask through vcc).  bench.py's roofline prices k_align against the saturated rate this prints.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/valu_issue_probe.hip && /tmp/valu_probe
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
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (a + d + e + f == 12345.678f) out[0] = 0;
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
  return 0;
}
