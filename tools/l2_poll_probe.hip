// l2_poll_probe.hip -- which kinds of read SEE another CU's L2 atomics on the same XCD (gfx950)?  (round 5: the XCD window of k_align polls a counter that
// other CUs of the same XCD advance with workgroup-scope atomic adds, i.e. adds performed in that XCD's L2.)
// Sixteen workgroups; workgroup w runs on XCD w % 8 under round-robin placement (checked: every workgroup reports its XCC_ID).  Workgroup 8 is the WRITER: it
// adds 1 to a counter N times, ~1 us apart.  Workgroup 0 is the READER: it polls the counter until it reads N (or gives up) with one of
//   0  s_load_dword glc            (scalar, "globally coherent")
//   1  s_dcache_inv + s_load_dword
//   2  global_load_dword sc0       (what an atomic add of 0 at workgroup scope compiles to)
//   3  global_load_dword sc1
//   4  global_load_dword sc0 sc1
//   5  global_atomic_add 0, returning
// and reports: the last value it saw, the number of polls, the 100 MHz ticks it took, and how many ticks one poll costs.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_poll_probe tools/l2_poll_probe.hip && /tmp/l2_poll_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int kMethod>
__device__ __forceinline__ unsigned read_counter(const uint32_t* p) {
  unsigned r = 0;
  if (kMethod == 0) asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  if (kMethod == 1) asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  if (kMethod == 2) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  if (kMethod == 3) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  if (kMethod == 4) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  if (kMethod == 5) { const unsigned z = 0; asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(z) : "memory"); }
  return r;
}

template <int kMethod>
__global__ void k_probe(uint32_t* counter, unsigned long long* out, int n_adds) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) & 15u;
  if (threadIdx.x != 0) return;
  out[8 + blockIdx.x] = xcc;
  if (blockIdx.x == 8) {                                   // writer
    for (int i = 0; i < n_adds; ++i) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __builtin_amdgcn_s_sleep(32);
    }
  } else if (blockIdx.x == 0) {                            // reader
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned v = 0; unsigned long long polls = 0;
    while (polls < 4000000ull) {
      v = read_counter<kMethod>(counter); ++polls;
      if (v >= (unsigned) n_adds) break;
    }
    out[0] = v; out[1] = polls; out[2] = __builtin_amdgcn_s_memrealtime() - t0;
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();      // the cost of a poll once the value is there
    for (int i = 0; i < 1000; ++i) v += read_counter<kMethod>(counter);
    out[3] = __builtin_amdgcn_s_memrealtime() - t1; out[4] = v;
  }
}

template <int kMethod> static void run(const char* name, uint32_t* d_counter, unsigned long long* d_out) {
  const int n_adds = 2000;
  unsigned long long h[32] = {0};
  hipMemset(d_counter, 0, 256); hipMemset(d_out, 0, sizeof h);
  hipLaunchKernelGGL(k_probe<kMethod>, dim3(16), dim3(64), 0, 0, d_counter, d_out, n_adds);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
  printf("%-32s saw %4llu of %d after %8llu polls, %9.1f us; a poll costs %6.1f ns; reader on XCC %llu, writer on XCC %llu%s %s\n", name, h[0], n_adds, h[1], h[2] * 0.01,
         h[3] * 10.0 / 1000.0, h[8], h[16], h[8] == h[16] ? "" : "  (NOT the same XCD: rerun)", e == hipSuccess ? "" : hipGetErrorString(e));
}

int main() {
  uint32_t* d_counter; unsigned long long* d_out;
  hipMalloc((void**) &d_counter, 256); hipMalloc((void**) &d_out, 256);
  run<0>("s_load_dword glc", d_counter, d_out);
  run<1>("s_dcache_inv + s_load_dword", d_counter, d_out);
  run<2>("global_load_dword sc0", d_counter, d_out);
  run<3>("global_load_dword sc1", d_counter, d_out);
  run<4>("global_load_dword sc0 sc1", d_counter, d_out);
  run<5>("global_atomic_add 0 (returning)", d_counter, d_out);
  return 0;
}
