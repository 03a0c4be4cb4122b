// lsm2d_device.h -- device-side arithmetic of the scan-matching hot path (gfx950, wave64).
//
// Every routine here is a FIXED sequence of IEEE-754 fp32 operations (explicit fmaf, IEEE divide and
// sqrt, no fast-math, built with -ffp-contract=off) so that column indices, z-buffer winners and
// per-pair factor terms are reproducible bit-for-bit on a CPU; only the order of the H/b sums is
// device-specific.  Reference semantics per routine:
//   transform   PointNormal2f::transform<Isometry> as used at
//               registration/correspondence_finder_kd_tree_2d.cpp:15-16
//   projection  PointNormal2fProjectorPolar::compute as used at
//               registration/correspondence_finder_projective_2d.cpp:40-41,47-48
//   factor      SE2Plane2PlaneErrorFactor = 2-D restriction of octave/solver/nicp_post.m:4-26
//   robustifier RobustifierCauchy (configurations/stage_segway_double_config_MULTI.json:153-158)
//   step        octave/solver/nicp_post.m:92-97 (dx = -H\b ; T = T*v2t(dx))
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsm2d {

typedef unsigned long long u64;
static constexpr u64 kEmptyCell = ~0ull;   // depth bits 0xFFFFFFFF (a NaN pattern no valid depth reaches), idx -1

#define LSM2D_DEV __device__ __forceinline__
#define LSM2D_HD  __host__ __device__ inline

struct Iso { float c, s, tx, ty; };          // R = [[c,-s],[s,c]], t = (tx,ty)

// projector constants, precomputed once on the host in fp32
struct ProjK {
  float K00, K01;       // column = floor(K00*theta + K01)
  float rmin, rmax;     // range gate on the depth r = sqrt_rn(r2) ...
  float r2lo, r2hi;     // ... restated on r2: rmin <= sqrt_rn(r2) <= rmax  <=>  r2lo <= r2 <= r2hi (host-proved, sqrt_rn is monotone)
  float colsf;          // (float) canvas_cols
  int   cols;
  int   tiny_ok;        // host-proved: a quotient min / r below 1e-12 / r cannot move a column, whatever tiny value it gets (make_projk)
};

// cos / sin of a pose angle as a fixed operation sequence (the CPU restatement evaluates the same sequence, operation for operation; coefficients from
// tools/fit_sincos.py, max abs error 9.3e-8): host code, kernels and the CPU oracle give a rotation the same bits -- glibc's and
// ocml's cosf / sinf differ in the last place now and then, and one such bit moves a point across a z-buffer column edge.
LSM2D_HD void sincos_fixed(float x, float& sn, float& cs) {
  const float kf = __builtin_rintf(x * 6.3661974669e-01f);
  float r = __builtin_fmaf(-kf, 1.5707963705e+00f, x);
  r = __builtin_fmaf(-kf, -4.3711388287e-08f, r);
  const float z = r * r;
  float ps = -1.9495635934e-04f; ps = __builtin_fmaf(ps, z, 8.3319786936e-03f); ps = __builtin_fmaf(ps, z, -1.6666650772e-01f);
  const float s = __builtin_fmaf(r * z, ps, r);
  float pc = 2.4438450055e-05f; pc = __builtin_fmaf(pc, z, -1.3887367677e-03f); pc = __builtin_fmaf(pc, z, 4.1666645557e-02f);
  const float c = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
  const int q = (int) kf & 3;
  sn = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
  cs = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}

// log of a positive normal number as a fixed operation sequence (the Cauchy kernel's statistic tau * log(1 + chi/tau) was the last
// libm call on the path; the CPU restatement evaluates the same sequence; tools/fit_log.py, max relative error 1.3e-7)
LSM2D_HD float log_fixed_inline(float x) {
  const uint32_t bits = __float_as_uint(x);
  int e = (int) (bits >> 23) - 127;
  float m = __uint_as_float((bits & 0x7FFFFFu) | 0x3F800000u);
  if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
  const float f = m - 1.0f, z = f * f;
  float p = -7.6311752200e-02f;
  p = __builtin_fmaf(p, f, 1.2854909897e-01f); p = __builtin_fmaf(p, f, -1.3207894564e-01f); p = __builtin_fmaf(p, f, 1.4190009236e-01f);
  p = __builtin_fmaf(p, f, -1.6616521776e-01f); p = __builtin_fmaf(p, f, 2.0001615584e-01f); p = __builtin_fmaf(p, f, -2.5001060963e-01f);
  p = __builtin_fmaf(p, f, 3.3333328366e-01f);
  const float r = __builtin_fmaf(z * f, p, __builtin_fmaf(-0.5f, z, f));
  return __builtin_fmaf((float) e, 6.9314718246e-01f, r);
}
// NOT inlined in the throughput kernels: a real call keeps its temporaries out of the register allocation of the loops around
// accumulate_pair (measured: k_align 2.09 -> 2.03 ms projective, 2.75 -> 2.58 ms NN role B against the inlined form); only
// Cauchy slices ever take the call.  The latency kernel (k_align_pair, registers to spare) inlines it.
__device__ __noinline__ float log_fixed(float x) { return log_fixed_inline(x); }

LSM2D_DEV void xf_point(const Iso& T, float px, float py, float& qx, float& qy) {
  qx = __builtin_fmaf(T.c, px, __builtin_fmaf(-T.s, py, T.tx));
  qy = __builtin_fmaf(T.s, px, __builtin_fmaf(T.c, py, T.ty));
}
LSM2D_DEV void xf_normal(const Iso& T, float nx, float ny, float& ox, float& oy) {
  ox = __builtin_fmaf(T.c, nx, (-T.s) * ny);
  oy = __builtin_fmaf(T.s, nx, T.c * ny);
}

LSM2D_DEV float wrap_angle(float a) {
  while (a > 3.14159274101257324f) a -= 6.28318548202514648f;
  while (a <= -3.14159274101257324f) a += 6.28318548202514648f;
  return a;
}

// ---- depth and bearing of a point from ONE transcendental ------------------------------------------------------------------
// tools/valu_issue_probe.hip: in this stream a transcendental costs ~18 cycles of a saturated SIMD (a plain instruction 2.15), and
// round 1 spent two per point -- v_rcp_f32 for atan(min / max) and v_rsq_f32 for the depth.  Both now grow from the depth's seed:
//   r   = sqrt_rn(r2)           correctly rounded square root:  y0 = v_rsq_f32(r2), s0 = r2 y0, r = s0 + (r2 - s0 s0)(y0 / 2)
//   t   = RN(min / r)           correctly rounded quotient = sin of the octant angle, t in [0, sqrt(1/2)]: one Newton step takes the
//                                seed to 1/r, then Markstein's residual correction
//   phi = t + t s P(s), s = t t  asin on [0, sqrt(1/2)], degree-6 P (tools/fit_asin.py: max abs error 4.6e-8 rad)
// Every step is IEEE fp32 (fmaf, sqrt, divide), so the CPU restatement reproduces r, t and phi bit for bit with sqrtf and '/'
// -- with ONE enumerated exception: tools/fp_exact_check.hip runs the two short sequences on this chip's v_rsq_f32 over EVERY input the
// range gate lets through (all r2 bit patterns in [1e-30, 1e37]; for the quotient all 2^23 mantissas of `min` against each of them):
// the square root is correctly rounded everywhere, the quotient everywhere except the exact ties its last fused step cannot see (r with
// an all-ones mantissa, `min` a power of two: the float BELOW the IEEE quotient comes out -- 4 inputs per binade).  The oracle's
// definition of the quotient follows the device on those ties (the CPU restatement's atan2 says so in its source), so CPU / GPU parity on them holds
// because the restatement mirrors the sequence, not because the sequence is exact.

// Correctly rounded sqrt for inputs in [1e-30, FLT_MAX] (the range gate's r2 and beyond); y0 returns the v_rsq_f32 seed.
LSM2D_DEV float sqrt_rn_seed(float x, float& y0) {
  y0 = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y0, h = 0.5f * y0;
  const float e  = __builtin_fmaf(-s0, s0, x);
  return __builtin_fmaf(e, h, s0);
}
LSM2D_DEV float sqrt_rn_normal(float x) { float y0; return sqrt_rn_seed(x, y0); }

// n / r for r = sqrt_rn(r2), y0 = v_rsq_f32(r2), 1e-12 <= n <= r: the IEEE-754 correctly rounded quotient except on the enumerated exact
// ties described above (one ulp low there; the oracle mirrors them).  n < 1e-12 (a quotient that could be
// subnormal, and n == 0) takes the compiler's full divide unless the host has proved that such a quotient cannot move a column
// (ProjK::tiny_ok; then kGuardTiny = false and the branch is gone).
template <bool kGuardTiny>
LSM2D_DEV float div_by_depth(float n, float r, float y0) {
  if (kGuardTiny && __builtin_expect(n < 1e-12f, 0)) return n / r;
  const float e0 = __builtin_fmaf(-r, y0, 1.0f);
  const float y1 = __builtin_fmaf(e0, y0, y0);
  const float q0 = n * y1;
  const float e1 = __builtin_fmaf(-r, q0, n);
  return __builtin_fmaf(e1, y1, q0);
}

// bearing of (x, y) given r = sqrt_rn(x*x + y*y) and its seed: exact octant fix-ups, sign of y copied onto the result (so the
// bearing of (-0, x < 0) is -pi like libm's atan2)
template <bool kGuardTiny>
LSM2D_DEV float bearing(float y, float x, float r, float y0) {
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const bool swap = ay > ax;
  const float mn = swap ? ax : ay;
  const float t = div_by_depth<kGuardTiny>(mn, r, y0);
  const float s = t * t;
  float p = 1.237212196e-01f;
  p = __builtin_fmaf(p, s, -1.153038889e-01f);
  p = __builtin_fmaf(p, s, 9.340071678e-02f);
  p = __builtin_fmaf(p, s, 1.043075230e-02f);
  p = __builtin_fmaf(p, s, 4.762428626e-02f);
  p = __builtin_fmaf(p, s, 7.478348911e-02f);
  p = __builtin_fmaf(p, s, 1.666723490e-01f);
  float phi = __builtin_fmaf(t * s, p, t);
  if (swap) phi = 1.57079637050628662f - phi;
  if (x < 0.0f) phi = 3.14159274101257324f - phi;
  return __builtin_copysignf(phi, y);
}

// One point of the polar z-buffer.  key = (bits(depth) << 32) | index: depth >= 0 so the IEEE bit
// pattern orders like the value, and the 64-bit unsigned min keeps the nearest point with ties going
// to the LOWEST index == "first point wins under strict <" of the sequential reference loop.
template <bool kReadFirst = true>
LSM2D_DEV void project_point(const Iso& T, const ProjK& P, float px, float py, int idx, u64* canvas) {
  float qx, qy;
  xf_point(T, px, py, qx, qy);
  const float r2 = __builtin_fmaf(qx, qx, qy * qy);
  if (r2 >= P.r2lo && r2 <= P.r2hi) {                      // r2 in [1e-30, 1e36]: normal numbers throughout
    float y0;
    const float r  = sqrt_rn_seed(r2, y0);
    const float th = bearing<true>(qy, qx, r, y0);
    const float u  = __builtin_fmaf(P.K00, th, P.K01);
    int col;                                               // floor + convert in one instruction (u is finite here;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(col) : "v"(u));  // negative / too large -> rejected below)
    if ((unsigned) col < (unsigned) P.cols) {
      const u64 key = ((u64) __float_as_uint(r) << 32) | (u64) (uint32_t) idx;
      if (kReadFirst) {
        // the cell only ever decreases, so a plain read that already beats us makes the atomic a no-op: neighbouring lanes
        // holding neighbouring points hit the SAME cell, where reads broadcast and atomics serialise
        const u64 cur = canvas[col];
        if (key < cur) atomicMin(&canvas[col], key);
      } else {
        atomicMin(&canvas[col], key);      // lane-chunked streams: lanes rarely share a cell, and a no-return atomic needs no wait
      }
    }
  }
}

// One point of k_align's lane-chunked stream (fire-and-forget z-buffer update): project_point's operation sequence, hence its bits,
// without the read of the cell.  What tools/valu_issue_probe.hip measured on the MI355X (profiles/r02/valu_issue_probe.txt): a plain
// VALU instruction costs 2.15 cycles of a saturated SIMD, two waves saturate it, instruction-level parallelism inside a wave buys
// nothing, and the whole stream below runs at the rate of its own instruction mix -- k_align's launch time IS points x cycles per point.
// kGuarded = false: the host has proved that a quotient below 1e-12 / r cannot move a column (ProjK::tiny_ok): no branch for it.
// Measured and rejected on top of this (same box, profiles/r02/variants_r02e.log): dropping the column check as well (a spare cell
// behind the canvas takes column `cols`, a full-circle canvas cannot produce any other) 1.494 ms against 1.478 with the check; pinning
// the gate's bounds in SGPRs in front of the loop (which removes an s_waitcnt lgkmcnt(0) per trip) 1.495 against 1.494.
template <bool kGuarded>
LSM2D_DEV void project_point_stream(const Iso& T, const ProjK& P, float px, float py, int idx, u64* canvas) {
  float qx, qy;
  xf_point(T, px, py, qx, qy);
  const float r2 = __builtin_fmaf(qx, qx, qy * qy);
  if (r2 >= P.r2lo && r2 <= P.r2hi) {
    float y0;
    const float r  = sqrt_rn_seed(r2, y0);
    const float th = bearing<kGuarded>(qy, qx, r, y0);
    const float u  = __builtin_fmaf(P.K00, th, P.K01);
    int col;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(col) : "v"(u));
    if ((unsigned) col < (unsigned) P.cols) atomicMin(&canvas[col], ((u64) __float_as_uint(r) << 32) | (u64) (uint32_t) idx);
  }
}

// Stream one cloud through the z-buffer.  xy is 16-byte aligned (cloud starts are padded to an even
// point index by the host) so every lane loads two points with one 16-byte global_load_dwordx4; the next
// load is issued before the current pair is processed.
LSM2D_DEV void project_cloud(const float2* __restrict__ xy, int n, const Iso& Tin, const ProjK& Pin,
                             u64* canvas, int tid, int nthreads) {
  const Iso T = Tin; const ProjK P = Pin;              // registers, not kernarg re-loads inside the loop
  const float4* __restrict__ xy4 = reinterpret_cast<const float4*>(xy);
  const int nfull = n >> 1;
  int j = tid;
  if (j < nfull) {
    float4 v = xy4[j];
    for (; j < nfull; j += nthreads) {
      const int jn = j + nthreads;
      const float4 nx = xy4[jn < nfull ? jn : j];
      // two fused-pair variants were measured and dropped: both LDS reads up front 50 % slower
      // (the second point must see the first one's update), straight-line arithmetic + sequential LDS phases 3 % slower
      project_point(T, P, v.x, v.y, 2 * j, canvas);
      project_point(T, P, v.z, v.w, 2 * j + 1, canvas);
      v = nx;
    }
  }
  if ((n & 1) && tid == 0) { const float2 t = xy[n - 1]; project_point(T, P, t.x, t.y, n - 1, canvas); }
}

// The same pass over a LANE-CHUNKED copy of the cloud (k_lane_layout): thread g owns the contiguous pairs
// [g*T, (g+1)*T) and the copy is stored step-major (slot t*nthreads + g), so every load is still one coalesced
// 16-byte access per lane -- but the 64 lanes of a wave now sit T pairs apart along the map, so lanes of one
// wave-instruction almost never hit the same cell: the same-address serialisation of ds_min_u64 (~1/3 of the kernel with
// neighbouring points in neighbouring lanes) is gone, and with it the reason to read a cell before updating it -- the
// update is one fire-and-forget LDS atomic and the stream never waits on the LDS.  Padding slots hold +inf and fail the
// range gate.
template <bool kGuarded>
LSM2D_DEV void project_cloud_lanes_t(const float4* __restrict__ lane_xy, int T_steps, const Iso& Tin, const ProjK& Pin,
                                     u64* canvas, int tid, int nthreads) {
  const Iso T = Tin; ProjK P = Pin;
  asm volatile("" : "+v"(P.K01));        // keep K01 in a VGPR: fma(K00, th, K01) may read only one SGPR, the compiler would v_mov it per point
  if (T_steps <= 0) return;
  const int base = tid * T_steps;
  // Buffer addressing: the cloud's base sits in a 4-SGPR resource, the lane adds a constant byte offset (voffset) and the row
  // advances in the scalar offset -- the load address costs no vector instruction.  Two steps per trip with the two load
  // buffers swapping roles, so no register copies either; each load has one whole pair of points of cover.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long pb = reinterpret_cast<unsigned long long>(lane_xy);       // wave-uniform, but it reached us through a
  const unsigned pb_hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (pb >> 32));       // vector load: say so.  (The builtin returns
  const unsigned pb_lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) pb);     // int: widen through unsigned, no sign extension)
  float4* ubase = reinterpret_cast<float4*>(((unsigned long long) pb_hi << 32) | (unsigned long long) pb_lo);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ubase, (short) 0, 0x7fffffff, 0x00020000);
  const int lane_off = tid * (int) sizeof(float4), row_bytes = nthreads * (int) sizeof(float4);
  int row = 0;                                           // < 2^31: ensure_lane_layout() keeps larger clouds off this path
  auto load = [&](int r) {
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, r, 0);
    return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
  };
  int idx = 2 * base, t = 0;
  float4 va = load(row);
  auto pair = [&](const float4& v, int i) {
    project_point_stream<kGuarded>(T, P, v.x, v.y, i, canvas);
    project_point_stream<kGuarded>(T, P, v.z, v.w, i + 1, canvas);
  };
  for (; t + 2 <= T_steps; t += 2) {
    const float4 vb = load(row + row_bytes);
    pair(va, idx);
    if (t + 2 < T_steps) row += 2 * row_bytes;            // else: a harmless re-read of the current row, never used
    va = load(row);
    pair(vb, idx + 2);
    idx += 4;
  }
  if (t < T_steps) pair(va, idx);
}
// ---- exact culling of the moving cloud against the FIXED canvas ---------------------------------------------------------------------
// The bin walk pairs a column's moving winner with its fixed cell only when the fixed cell is filled and |depth_f - depth_m| <=
// point_distance (correspondence_finder_projective_2d.cpp:61-65).  A moving point whose column has an EMPTY fixed cell, or whose depth
// exceeds depth_f + point_distance, therefore never yields a pair -- and dropping ALL such points leaves every pair as it was: if the
// column's winner is one of them, the pair it would have formed is rejected by the depth gate, and every other point of the column that
// could take its place is farther still, i.e. dropped as well; if the winner is not one of them, it stays the winner.  The moving canvas
// differs, the correspondences (hence H, b, the statistics and the pose) do not: bit for bit.
// The test is made per CHUNK (the contiguous run of map points one thread of k_align owns), on its bounding circle (centre c, radius
// rho, computed once per cloud: k_lane_bounds): under the pose T every point of the chunk has depth >= |T c| - rho and a bearing within
// asin(rho / |T c|) of the centre's.  The chunk is dropped when that depth bound exceeds depth_f + point_distance in EVERY column its
// bearing range can reach (empty fixed cells count as "no depth at all"), or the range gate, with margins far above fp32 rounding.
// On configs[1] (a 270-degree scan in a room with pillars) 57 % of the chunks of a 100k-point map fall to it: blind sector 22 %, beyond
// range_max 16 %, behind what the scan saw 19 %.
// Round 4: the test serves a NEIGHBOURHOOD of poses.  Seen from the sensor, a change of the estimate from T0 to T is a rotation about the sensor's
// origin by dth followed by a translation d (q = R(dth) q0 + d, d = t - R(dth) t0).  With margins m_t >= |d| and m_th >= |dth| every point of the
// circle is, under T, at least D - m_t - rho deep and within m_th + asin(m_t / (D - m_t)) + asin(rho / (D - m_t)) of the centre's bearing under T0:
// a circle dropped with these margins cannot matter under any such T, so a survivor list built at T0 stays valid while the estimate stays within
// (m_t, m_th) of T0 -- k_align rebuilds it a few times per alignment instead of testing every chunk every iteration.  m_t = m_th = 0 is the
// round-3 test.  Keeping a circle that could have gone is always harmless (its points are the ones whose presence changes no pair).
LSM2D_DEV bool chunk_may_matter(const Iso& T, const ProjK& P, float4 bd /* cx, cy, rho (< 0: no points) */, const u64* fcan, float point_distance,
                                float m_t = 0.0f, float m_th = 0.0f) {
  if (bd.z < 0.0f) return false;
  float qx, qy;
  xf_point(T, bd.x, bd.y, qx, qy);
  const float r2 = __builtin_fmaf(qx, qx, qy * qy);
  if (!(r2 >= 1e-30f && r2 <= 1e37f)) return true;                  // degenerate: no claim
  float y0;
  const float Dc = sqrt_rn_seed(r2, y0), rho = bd.z;
  const float D = Dc - m_t;                                          // the centre is at least this deep under every pose the answer serves
  const float near = D - rho - (1e-3f + 1e-5f * Dc);                // every point of the chunk is at least this deep (triangle inequality, minus slack)
  if (near > P.rmax) return false;                                   // the range gate takes them all
  if (!(D > 2.0f * (rho + m_t) + 0.05f)) return true;               // the sensor is next to (or inside) the chunk: its bearings spread over more than 30 degrees
  const float th = bearing<true>(qy, qx, Dc, y0);
  const float u = __builtin_fmaf(P.K00, th, P.K01);
  // asin(x) <= 1.048 x on [0, 1/2].  The columns the circle's points can fall into are floor(u - w) .. floor(u + w) for the true half-width w: the two
  // floors below cover them, and what the arithmetic adds -- the bearing polynomial's 5e-8 rad, the roundings of u and of a point's own column -- is below
  // 1e-3 columns at 16 384 of them: 0.05 columns of slack.  (Round 3 allowed 1.5 columns each side: at BLOCK level -- a block is about one column wide at
  // 10 m -- that tripled every window; measured on configs[1]: 41 % of the point visits left with 1.5, 3x % with 0.05, the floor for 28-point blocks being 32 %.)
  const float dc = P.K00 * (((rho + m_t) / D) * 1.06f + m_th) + 0.05f;
  if (!(dc <= 24.0f)) return true;
  const int c_lo = (int) __builtin_floorf(u - dc), c_hi = (int) __builtin_floorf(u + dc);
  float deepest = -__builtin_huge_valf();                            // deepest fixed cell among the columns in reach; empty cells (a NaN pattern) do not count
  for (int c = c_lo; c <= c_hi; ++c) {
    const int cc = c < 0 ? c + P.cols : (c >= P.cols ? c - P.cols : c);      // a full-circle canvas wraps; for a partial one the wrapped column is a harmless extra
    if ((unsigned) cc < (unsigned) P.cols) deepest = __builtin_fmaxf(deepest, __uint_as_float((uint32_t) (fcan[cc] >> 32)));
  }
  return !(near > deepest + point_distance);
}

// The surviving chunks' points through the z-buffer, spread evenly over the workgroup: a chunk's T steps are cut into nb blocks of B
// steps; unit v = (block v / s, survivor v mod s); thread u takes the units u, u + nthreads, ...  Neighbouring lanes hold neighbouring
// SURVIVORS at the same block offset -- still a chunk (>= 43 cm of wall on configs[1]) apart, so a wave-instruction's ds_min_u64s almost
// never share a cell (the reason for the lane-chunked layout), and their 16-byte loads fall into one or two rows of the copy.
template <bool kGuarded>
LSM2D_DEV void project_cloud_units_t(const float4* __restrict__ lane_xy, int T_steps, const Iso& Tin, const ProjK& Pin, u64* canvas, int tid, int nthreads,
                                     const uint16_t* surv, int s, int B, int nb) {
  const Iso T = Tin; ProjK P = Pin;
  asm volatile("" : "+v"(P.K01));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long pb = reinterpret_cast<unsigned long long>(lane_xy);
  const unsigned pb_hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (pb >> 32));
  const unsigned pb_lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) pb);
  float4* ubase = reinterpret_cast<float4*>(((unsigned long long) pb_hi << 32) | (unsigned long long) pb_lo);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ubase, (short) 0, 0x7fffffff, 0x00020000);
  const int row_bytes = nthreads * (int) sizeof(float4);
  int i = tid, blk = 0;
  while (i >= s && blk < nb) { i -= s; ++blk; }
  while (blk < nb) {
    const int g = (int) surv[i];
    const int t0 = blk * B;
    const int nsteps = T_steps - t0 < B ? T_steps - t0 : B;
    const int voff = g * (int) sizeof(float4) + t0 * row_bytes;      // this lane's chunk and block; the step advances in the scalar offset
    int idx = 2 * (g * T_steps + t0);
    auto load = [&](int soff) {
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
      return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    };
    auto pair = [&](const float4& v, int k) {
      project_point_stream<kGuarded>(T, P, v.x, v.y, k, canvas);
      project_point_stream<kGuarded>(T, P, v.z, v.w, k + 1, canvas);
    };
    // two steps per trip, the two load buffers swapping roles.  The scalar offset advances unconditionally (it must stay wave-uniform: the
    // last block of a chunk is shorter in some lanes): the last trip's look-ahead load reads one row past the block -- the next block's,
    // the next cloud's, or one of the two spare rows ensure_lane_layout() keeps behind the last cloud -- and is never used.
    int t = 0, soff = 0;
    float4 va = load(0);
    for (; t + 2 <= nsteps; t += 2) {
      const float4 vb = load(soff + row_bytes);
      pair(va, idx);
      soff += 2 * row_bytes;
      va = load(soff);
      pair(vb, idx + 2);
      idx += 4;
    }
    if (t < nsteps) pair(va, idx);
    i += nthreads;
    while (i >= s && blk < nb) { i -= s; ++blk; }
  }
}
// Round 4: the same stream over an explicit LIST of surviving units in LDS (k_align, kProjCulled): entry = (block << 9) | chunk, block-major -- the
// lanes of a wave hold consecutive entries, i.e. neighbouring surviving chunks at the same block offset (the spacing that keeps a wave's ds_min_u64s
// off each other's cells, as above) -- and every entry is a unit that survived the BLOCK-level test (a block = B steps = 2 B consecutive map points,
// ~6 cm of wall on configs[1] and [4] alike), so a chunk that straddles the edge of what the scan saw is streamed in part only.  Thread u takes the
// entries u, u + nthreads, ...: whole waves run out of work together.
// Round 5: the XCD lockstep of k_align<1,0,0,0,6> (AlignArgs::xcd_sync in lsm2d_kernels.h) -- how a workgroup LOOKS at its XCD's counters: a SCALAR load with glc
// (past the scalar cache, served by the L2, where the other CUs' atomic adds are performed; tools/l2_poll_probe.hip: it sees them within a poll, a plain or
// sc0 vector load -- which is what the compiler makes of an atomic add of zero -- never does).  One request, no lane, no vector register, and a READ.
LSM2D_DEV unsigned xcd_look(const uint32_t* p) {
  unsigned v;
  asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(p) : "memory");
  return v;
}
// wait until every workgroup registered on this XCD has counted itself into done[need] (or has gone).  Called by ONE thread of a workgroup while the others
// stand at a barrier.  The counters only grow, and the workgroup that is furthest behind waits for nobody, so every wait ends; should that reasoning ever be
// wrong, ~100 ms end it too (the lockstep changes no result: giving up on it is always safe) -- counted in word 2 of the XCD's counters (LSM2D_DUMP_XCD).
LSM2D_DEV void xcd_wait(uint32_t* sync, int need, int limit) {
  if (need < 0 || need >= limit) return;
  int polls = 0;
#pragma nounroll
  for (;;) {
    const unsigned d = xcd_look(sync + 16 + need), reg = xcd_look(sync), gone = xcd_look(sync + 1);
    if ((int) (d + gone - reg) >= 0) return;
    if (polls < 8) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(48);      // ~0.25 us at first, ~1.3 us later: 125 workgroups per XCD, one looking thread each
    if (++polls > 60000) { __hip_atomic_fetch_add(&sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); sync[3] = (unsigned) need; sync[4] = d; sync[5] = reg; sync[6] = gone; return; }
  }
}

// (kStride: the threads that share the list -- the workgroup's width; nthreads: the chunks of a row of the lane-chunked copy, 512 whatever the width.  0: the same)
template <bool kGuarded, int kStride = 0>
LSM2D_DEV void project_cloud_list_t(const float4* __restrict__ lane_xy, int T_steps, const Iso& Tin, const ProjK& Pin, u64* canvas, int tid, int nthreads,
                                    const uint16_t* units, int n_units, int B) {
  const Iso T = Tin; ProjK P = Pin;
  asm volatile("" : "+v"(P.K01));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long pb = reinterpret_cast<unsigned long long>(lane_xy);
  const unsigned pb_hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (pb >> 32));
  const unsigned pb_lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) pb);
  float4* ubase = reinterpret_cast<float4*>(((unsigned long long) pb_hi << 32) | (unsigned long long) pb_lo);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ubase, (short) 0, 0x7fffffff, 0x00020000);
  const int row_bytes = nthreads * (int) sizeof(float4);
  for (int k = tid; k < n_units; k += (kStride > 0 ? kStride : nthreads)) {
    const int code = (int) units[k];
    const int g = code & 511, t0 = (code >> 9) * B;
    const int nsteps = T_steps - t0 < B ? T_steps - t0 : B;
    const int voff = g * (int) sizeof(float4) + t0 * row_bytes;      // this lane's chunk and block; the step advances in the scalar offset
    int idx = 2 * (g * T_steps + t0);
    auto load = [&](int soff) {
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
      return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    };
    auto pair = [&](const float4& v, int i) {
      project_point_stream<kGuarded>(T, P, v.x, v.y, i, canvas);
      project_point_stream<kGuarded>(T, P, v.z, v.w, i + 1, canvas);
    };
    // (two steps per trip, the two load buffers swapping roles; the scalar offset advances unconditionally and the last trip's look-ahead load reads
    // one row past the block -- never used: project_cloud_units_t)
    int t = 0, soff = 0;
    float4 va = load(0);
    for (; t + 2 <= nsteps; t += 2) {
      const float4 vb = load(soff + row_bytes);
      pair(va, idx);
      soff += 2 * row_bytes;
      va = load(soff);
      pair(vb, idx + 2);
      idx += 4;
    }
    if (t < nsteps) pair(va, idx);
  }
}
template <int kStride = 0>
LSM2D_DEV void project_cloud_list(const float4* __restrict__ lane_xy, int T_steps, const Iso& T, const ProjK& P, u64* canvas, int tid, int nthreads,
                                  const uint16_t* units, int n_units, int B) {
  if (P.tiny_ok) project_cloud_list_t<false, kStride>(lane_xy, T_steps, T, P, canvas, tid, nthreads, units, n_units, B);
  else project_cloud_list_t<true, kStride>(lane_xy, T_steps, T, P, canvas, tid, nthreads, units, n_units, B);
}

#ifdef LSM2D_EXPERIMENTS      // ("cull" 2: measured 3 % slower than the block units, DESIGN App. A; compiled into the experiments build only)
// The same survivors, balanced to within one step: the s surviving chunks x T steps form a matrix (row = step t, column = survivor i);
// thread u takes the elements u, u + nthreads, ... of its row-major order.  The lanes of a wave then hold consecutive survivors at the SAME
// step -- their 16-byte loads fall into one row of the copy, and they stay a chunk apart along the map -- and every thread gets
// floor or ceil of s T / nthreads steps.  One step (two points) per element: the column is advanced incrementally (no division in the
// loop), the survivor's chunk comes from the LDS list one element ahead, two loads are in flight.
template <bool kGuarded>
LSM2D_DEV void project_cloud_rows_t(const float4* __restrict__ lane_xy, int T_steps, const Iso& Tin, const ProjK& Pin, u64* canvas, int tid, int nthreads,
                                    const uint16_t* surv, int s) {
  const Iso T = Tin; ProjK P = Pin;
  asm volatile("" : "+v"(P.K01));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long pb = reinterpret_cast<unsigned long long>(lane_xy);
  const unsigned pb_hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (pb >> 32));
  const unsigned pb_lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) pb);
  float4* ubase = reinterpret_cast<float4*>(((unsigned long long) pb_hi << 32) | (unsigned long long) pb_lo);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ubase, (short) 0, 0x7fffffff, 0x00020000);
  const int row_bytes = nthreads * (int) sizeof(float4);
  const int q0 = nthreads / s, d = nthreads - q0 * s;          // one division per call, wave-uniform: an element's successor is q0 rows and d columns on
  // (t, i): step and column of the NEXT element to be loaded
  int t = 0, i = tid;
  if (i >= s) { const int q = i / s; t = q; i -= q * s; }
  auto advance = [&]() { t += q0; i += d; if (i >= s) { i -= s; ++t; } };
  auto issue = [&](int g, int tt, float4& buf, int& idx) {
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, g * (int) sizeof(float4) + tt * row_bytes, 0, 0);
    buf = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    idx = 2 * (g * T_steps + tt);
  };
  auto pair = [&](const float4& v, int k) {
    project_point_stream<kGuarded>(T, P, v.x, v.y, k, canvas);
    project_point_stream<kGuarded>(T, P, v.z, v.w, k + 1, canvas);
  };
  float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va; int ia = 0, ib = 0;
  bool ha = t < T_steps;
  if (ha) { issue((int) surv[i], t, va, ia); advance(); }
  bool hb = ha && t < T_steps;
  if (hb) { issue((int) surv[i], t, vb, ib); advance(); }
  int g_next = (hb && t < T_steps) ? (int) surv[i] : 0;        // the survivor of the element after those two, fetched ahead of its use
  while (ha) {
    pair(va, ia);
    ha = hb && t < T_steps;
    if (ha) { issue(g_next, t, va, ia); advance(); g_next = t < T_steps ? (int) surv[i] : 0; }
    if (!hb) break;
    pair(vb, ib);
    hb = ha && t < T_steps;
    if (hb) { issue(g_next, t, vb, ib); advance(); g_next = t < T_steps ? (int) surv[i] : 0; }
  }
}
LSM2D_DEV void project_cloud_rows(const float4* __restrict__ lane_xy, int T_steps, const Iso& T, const ProjK& P, u64* canvas, int tid, int nthreads,
                                  const uint16_t* surv, int s) {
  if (P.tiny_ok) project_cloud_rows_t<false>(lane_xy, T_steps, T, P, canvas, tid, nthreads, surv, s);
  else project_cloud_rows_t<true>(lane_xy, T_steps, T, P, canvas, tid, nthreads, surv, s);
}
#endif

LSM2D_DEV void project_cloud_units(const float4* __restrict__ lane_xy, int T_steps, const Iso& T, const ProjK& P, u64* canvas, int tid, int nthreads,
                                   const uint16_t* surv, int s, int B, int nb) {
  if (P.tiny_ok) project_cloud_units_t<false>(lane_xy, T_steps, T, P, canvas, tid, nthreads, surv, s, B, nb);
  else project_cloud_units_t<true>(lane_xy, T_steps, T, P, canvas, tid, nthreads, surv, s, B, nb);
}

LSM2D_DEV void project_cloud_lanes(const float4* __restrict__ lane_xy, int T_steps, const Iso& T, const ProjK& P,
                                   u64* canvas, int tid, int nthreads) {
  if (P.tiny_ok) project_cloud_lanes_t<false>(lane_xy, T_steps, T, P, canvas, tid, nthreads);
  else project_cloud_lanes_t<true>(lane_xy, T_steps, T, P, canvas, tid, nthreads);
}

// ---- factor ---------------------------------------------------------------------------------
struct Accum {      // one lane's partial sums of H (6 unique), b (3), chi and counts
  float h00, h01, h02, h11, h12, h22, b0, b1, b2, chi_in, chi_out;
  int n_in, n_out, n_corr;
};
LSM2D_DEV void accum_zero(Accum& a) {
  a.h00 = a.h01 = a.h02 = a.h11 = a.h12 = a.h22 = a.b0 = a.b1 = a.b2 = a.chi_in = a.chi_out = 0.0f;
  a.n_in = a.n_out = a.n_corr = 0;
}

// e = [ n_f.(q - p_f) ; n_q - n_f ],  J = [[ (R^T n_f)^T , n_f.(R J2 p_m) ], [ 0 , R J2 n_m ]]
// inl_only: an iteration of the aligner's inlier-only runs (lsm2d_aligner_params.enable_inlier_only_runs): a pair that is not an inlier under
// the slice's robustifier contributes nothing to H and b (weight +0: every fused add below then returns its addend), an inlier weight 1;
// the statistics are formed as in a regular iteration.  false: the regular loop, bit for bit what it was.
template <bool kInlineLog = false>
LSM2D_DEV void accumulate_pair(const Iso& T, float2 pf, float2 nf, float2 pm, float2 nm, bool cauchy,
                               float tau, Accum& A, bool inl_only = false) {
  float qx, qy, nqx, nqy;
  xf_point(T, pm.x, pm.y, qx, qy);
  xf_normal(T, nm.x, nm.y, nqx, nqy);
  const float dx = qx - pf.x, dy = qy - pf.y;
  const float e0 = __builtin_fmaf(nf.x, dx, nf.y * dy);
  const float e1 = nqx - nf.x, e2 = nqy - nf.y;
  const float a0 = __builtin_fmaf(T.c, nf.x, T.s * nf.y);
  const float a1 = __builtin_fmaf(-T.s, nf.x, T.c * nf.y);
  const float a2 = __builtin_fmaf(a1, pm.x, -(a0 * pm.y));
  const float d0 = -nqy, d1 = nqx;
  const float chi = __builtin_fmaf(e0, e0, __builtin_fmaf(e1, e1, e2 * e2));
  // branch-free bookkeeping (adding +0.0f is exact), so the sums stay in registers
  float w = 1.0f, kern = 0.0f;
  bool inlier = true;
  if (cauchy) {
    const float q = chi / tau;
    w = 1.0f / (1.0f + q);
    inlier = chi < tau;
    if (!inlier) kern = tau * (kInlineLog ? log_fixed_inline(1.0f + q) : log_fixed(1.0f + q));      // only outliers' statistic uses it: waves of inliers skip the logarithm
    if (inl_only) w = inlier ? 1.0f : 0.0f;
  }
  A.n_in += inlier ? 1 : 0;
  A.n_out += inlier ? 0 : 1;
  A.chi_in += inlier ? chi : 0.0f;
  A.chi_out += inlier ? 0.0f : kern;
  ++A.n_corr;
  const float dd = __builtin_fmaf(d0, d0, d1 * d1);
  const float de = __builtin_fmaf(d0, e1, d1 * e2);
  const float wa0 = w * a0, wa1 = w * a1, wa2 = w * a2;
  A.h00 = __builtin_fmaf(wa0, a0, A.h00); A.h01 = __builtin_fmaf(wa0, a1, A.h01); A.h02 = __builtin_fmaf(wa0, a2, A.h02);
  A.h11 = __builtin_fmaf(wa1, a1, A.h11); A.h12 = __builtin_fmaf(wa1, a2, A.h12);
  A.h22 = __builtin_fmaf(wa2, a2, A.h22); A.h22 = __builtin_fmaf(w, dd, A.h22);
  A.b0 = __builtin_fmaf(wa0, e0, A.b0); A.b1 = __builtin_fmaf(wa1, e0, A.b1);
  A.b2 = __builtin_fmaf(wa2, e0, A.b2); A.b2 = __builtin_fmaf(w, de, A.b2);
}

// chi^2 of ONE pair: accumulate_pair's operations up to `chi`, in its order -- what decides "inlier" (chi < tau) when the aligner's
// keep_only_inlier_correspondences filters the correspondences it hands back (lsm2d_align_batch_pairs)
LSM2D_DEV float pair_chi(const Iso& T, float2 pf, float2 nf, float2 pm, float2 nm) {
  float qx, qy, nqx, nqy;
  xf_point(T, pm.x, pm.y, qx, qy);
  xf_normal(T, nm.x, nm.y, nqx, nqy);
  const float dx = qx - pf.x, dy = qy - pf.y;
  const float e0 = __builtin_fmaf(nf.x, dx, nf.y * dy);
  const float e1 = nqx - nf.x, e2 = nqy - nf.y;
  return __builtin_fmaf(e0, e0, __builtin_fmaf(e1, e1, e2 * e2));
}

// ---- "sum_order" 1: H, b and the chi^2 statistics added PAIR AFTER PAIR, the reference's order --------------------------------------
// The reference adds one factor after the other into H and b (octave/solver/nicp_post.m:69-90: H += J'*J; b += J'*e inside the loop over the
// correspondences; the correspondence vector is in ascending column for the projective finder, correspondence_finder_projective_2d.cpp:55-74, and
// in ascending moving index for the point-query finders, correspondence_finder_kd_tree_2d.cpp:12-27).  The default kernels add in trees (a thread's
// pairs, then 64 lanes, then 8 waves): the same terms, another association.  With the option on, every thread that holds a pair writes the pair's
// TERMS -- accumulate_pair's operations up to the sums, in its order -- as a record of kSeqFields floats into LDS, slot = the pair's position in the
// reference's order within the current trip of the workgroup (a trip: 512 consecutive columns / moving indices, handed over in two halves); after the barrier
// eleven lanes of wave 0 walk the records in ascending slot, lane q adding quantity q with exactly the fused operations the sequential CPU restatement of the factor
// uses (the tests hold the two against each other bit for bit): h = fma(w a_i, a_j, h), two of them chained for h22 and b2, chi sums by plain adds.  A slot without a pair holds zeros:
// fma(+0, +0, h) == h and h + 0 == h bit for bit (no sum here is ever -0: they start at +0).  The counts are integers and keep their ballots.
// Record = 14 floats, 56 bytes: a0 a1 a2 e0 | wa0 wa1 wa2 w | dd de chi_in chi_out | 1 0 (wa_i = w * a_i, rounded by the thread that holds the pair: the
// reference's `wa` temporaries; the two constants travel with every record).  The records of HALF a trip (kSeqHalf = 256) sit in LDS; lane q of the walking wave
// reads, per record, the four operands of ITS two fused adds straight from LDS -- acc = fma(x1, y1, acc); acc = fma(x2, y2, acc) -- at four per-lane FIELD
// offsets and one stride for all: an operand that is a constant for this lane (1 for the chi sums' y1; 0, 0 for the second add of the nine quantities that have
// none) reads the record's constant fields.  So a record costs the walk four LDS reads with immediate offsets (the compiler pairs them: ds_read2_b32) and two
// dependent FMAs, nothing else.  (Round 6, measured in isolation -- tools/seq_walk_probe.py: a first form that selected and multiplied in the loop and waited for
// each record's loads 135 cycles per record; per-lane strides for the constants -- sixteen address adds per four records -- 65; this one: see DESIGN section 5.)
static constexpr int kSeqFields = 14, kSeqHalf = 256;
static constexpr int kSeqLdsBytes = kSeqHalf * kSeqFields * 4;
template <bool kInlineLog = false>
LSM2D_DEV void pair_terms(const Iso& T, float2 pf, float2 nf, float2 pm, float2 nm, bool cauchy, float tau, bool inl_only,
                          float (&t)[kSeqFields], bool& inlier) {
  float qx, qy, nqx, nqy;
  xf_point(T, pm.x, pm.y, qx, qy);
  xf_normal(T, nm.x, nm.y, nqx, nqy);
  const float dx = qx - pf.x, dy = qy - pf.y;
  const float e0 = __builtin_fmaf(nf.x, dx, nf.y * dy);
  const float e1 = nqx - nf.x, e2 = nqy - nf.y;
  const float a0 = __builtin_fmaf(T.c, nf.x, T.s * nf.y);
  const float a1 = __builtin_fmaf(-T.s, nf.x, T.c * nf.y);
  const float a2 = __builtin_fmaf(a1, pm.x, -(a0 * pm.y));
  const float d0 = -nqy, d1 = nqx;
  const float chi = __builtin_fmaf(e0, e0, __builtin_fmaf(e1, e1, e2 * e2));
  float w = 1.0f, kern = 0.0f;
  inlier = true;
  if (cauchy) {
    const float q = chi / tau;
    w = 1.0f / (1.0f + q);
    inlier = chi < tau;
    if (!inlier) kern = tau * (kInlineLog ? log_fixed_inline(1.0f + q) : log_fixed(1.0f + q));
    if (inl_only) w = inlier ? 1.0f : 0.0f;
  }
  t[0] = a0; t[1] = a1; t[2] = a2; t[3] = e0;
  t[4] = w * a0; t[5] = w * a1; t[6] = w * a2; t[7] = w;
  t[8] = __builtin_fmaf(d0, d0, d1 * d1);
  t[9] = __builtin_fmaf(d0, e1, d1 * e2);
  t[10] = inlier ? chi : 0.0f;
  t[11] = inlier ? 0.0f : kern;
}
// "no pair in this slot": zeros and the constants
LSM2D_DEV void seq_zero(float (&t)[kSeqFields]) {
#pragma unroll
  for (int f = 0; f < 12; ++f) t[f] = 0.0f;
  t[12] = 1.0f; t[13] = 0.0f;
}
// `rec`: the region's start; record `slot` (0 .. kSeqHalf - 1) as seven 8-byte stores
LSM2D_DEV void seq_store(float* rec, int slot, const float (&t)[kSeqFields]) {
  float2* r = reinterpret_cast<float2*>(rec + slot * kSeqFields);
#pragma unroll
  for (int f = 0; f < kSeqFields / 2; ++f) r[f] = make_float2(t[2 * f], t[2 * f + 1]);
}
// The walk, one wave.  The chain of a quantity is serial -- acc = fma(x1, y1, acc); acc = fma(x2, y2, acc), record after record -- and what a single lane walking
// it pays per record is the LDS round trip of the record's operands as far as nothing covers it: the first form of this round (lane q < 11 walks quantity q, two
// groups of two records in flight: all the registers the kernel has for it) took 44 cycles per record.  Now a QUAD of lanes walks a quantity: lane j of quad q
// (lane 4 q + j) holds the operands of records 8 b + 2 j and 8 b + 2 j + 1 of batch b, so ONE pair of LDS reads per operand brings eight records, and the running
// sum travels round the quad -- quad_perm:[3,0,1,2], one DPP move -- picking up each lane's two records in ascending record order: lane 0 takes it from lane 3
// (the previous batch's end), adds records 8 b and 8 b + 1, lane 1 takes it from lane 0, ...  Every lane executes every step; the value that counts is in lane s
// after step s, what the other lanes compute meanwhile is overwritten before it is ever read on the chain.  Per record: two fused adds + half a move = 2.5
// vector instructions on the chain and an eighth of the LDS reads, the next batch's loads in flight under this batch's twenty instructions.  The same fused operations on
// the same values in the same order as before: the same bits (the tests hold the kernel against the sequential CPU restatement bit for bit).
// `acc`: the caller keeps it per lane between calls (zero at first, in all lanes); the running sum of quantity q is lane 4 q + 3's after every call -- seq_total()
// brings it to lane q.  Records n .. 8 ceil(n / 8) - 1 of the buffer must hold "no pair" records (k_align_seq, which walks its pairs only, pads them; the split path's kernels write every slot of a half-trip).
#ifndef LSM2D_SEQ_WALK_QUADS
#define LSM2D_SEQ_WALK_QUADS 1      // 0: the single-lane walk (A/B: profiles/r06/sum_order_walker_quads_ab_r06.txt)
#endif
// Measured (tools/seq_walk_probe.py: cycles per record in isolation; configs[1] with "sum_order" 1: k_align_seq ms; profiles/r06/sum_order_walker_quads_ab_r06.txt):
//   single lane 44 / 1.123;  quads, R records per lane and batch, B batches in registers: R 1 B 4: 44 / 1.114;  R 2 B 2 (shipped): 30 / 0.955-0.961;  R 2 B 3: 28 / 0.971;
//   R 4 B 2: 23 / 0.944 with 48 bytes of scratch in the headline instantiation.  The cost is ~14 cycles for a record's two dependent fused adds + ~30 per move of the
//   sum to the next lane (a DPP result feeding the chain), i.e. per R records.  Bringing the OPERANDS to the sum instead (quad broadcasts, the x operand as
//   v_fmac_f32_dpp's own): 33-37 / 1.04-1.06 -- four DPP instructions per record cost more than half a move.  Loads pinned ahead of the chain (sched_barrier): slower.
#ifndef LSM2D_SEQ_QUAD_R
#define LSM2D_SEQ_QUAD_R 2
#endif
#ifndef LSM2D_SEQ_QUAD_BUFS
#define LSM2D_SEQ_QUAD_BUFS 2
#endif
#ifndef LSM2D_SEQ_UNROLL
#define LSM2D_SEQ_UNROLL 2      // single-lane walk: records per group; two groups in flight (2 / 3 / 4 / 6: 1.119 / 1.143 / 1.115 / 1.199 ms on configs[1]; 4 and up spill at 64 registers)
#endif
LSM2D_DEV float seq_quad_prev(float v) {      // lane 4 q + j <- lane 4 q + (j + 3) % 4
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x93 /* quad_perm:[3,0,1,2] */, 0xF, 0xF, true));
}
template <bool kTopPriority = false>      // kTopPriority: s_setprio 3 once the walk's addresses are made (k_align_seq: the caller says when it ends)
LSM2D_DEV float seq_walk(const float* rec, int n, int lane, float acc) {
  asm volatile("" : "+v"(lane));      // (the lane's field offsets are made HERE, every time: as loop invariants of the iteration loop they are registers held -- and spilled -- across the whole kernel)
#if LSM2D_SEQ_WALK_QUADS
  const int q = (lane >> 2) < 11 ? (lane >> 2) : 0, j = lane & 3;      // (quads 11 .. 15 walk along like quad 0 and hold nothing of value)
#else
  const int q = lane < 11 ? lane : 0;
#endif
  const bool chi = q >= 9, has2 = q == 5 || q == 8;
  const int fx1 = (int) ((0xBA654655444ull >> (4 * q)) & 15), fy1 = chi ? 12 : (int) ((0x00333221210ull >> (4 * q)) & 15);      // field of x1 (wa_i, or the chi term) and of y1 (a_j, e0, or the constant 1)
  const int fx2 = has2 ? 7 : 13, fy2 = has2 ? (q == 5 ? 8 : 9) : 13;                                                             // w and dd / de; elsewhere the constant 0 twice
#if LSM2D_SEQ_WALK_QUADS
  constexpr int kR = LSM2D_SEQ_QUAD_R, kBufs = LSM2D_SEQ_QUAD_BUFS;      // records per lane and batch; batches in registers
  const float* r0 = rec + kR * j * kSeqFields;      // this lane's first record of batch 0
  const float* px1 = r0 + fx1; const float* py1 = r0 + fy1; const float* px2 = r0 + fx2; const float* py2 = r0 + fy2;
  constexpr int kBatch = 4 * kR * kSeqFields;      // floats from one batch to the next
  struct Ops { float x1[kR], y1[kR], x2[kR], y2[kR]; };
  auto load = [&](int b) {
    Ops o; const int off = b * kBatch;
#pragma unroll
    for (int r = 0; r < kR; ++r) { o.x1[r] = px1[off + r * kSeqFields]; o.y1[r] = py1[off + r * kSeqFields]; o.x2[r] = px2[off + r * kSeqFields]; o.y2[r] = py2[off + r * kSeqFields]; }
    return o;
  };
  auto chain = [&](const Ops& o) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float t = seq_quad_prev(acc);
#pragma unroll
      for (int r = 0; r < kR; ++r) { t = __builtin_fmaf(o.x1[r], o.y1[r], t); t = __builtin_fmaf(o.x2[r], o.y2[r], t); }
      acc = t;
    }
  };
  const int nb = (n + 4 * kR - 1) / (4 * kR);
  if (nb <= 0) return acc;
  Ops buf[kBufs];
  if constexpr (kTopPriority) __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int i = 0; i + 1 < kBufs; ++i) buf[i] = load(i < nb ? i : nb - 1);
  for (int b = 0; b < nb; b += kBufs) {      // buf[i] holds batch b + i; the free buffer takes batch b + i + kBufs - 1 (past the end: the last batch again, never used)
#pragma unroll
    for (int i = 0; i < kBufs; ++i) {
      if (b + i < nb) {
        const int nxt = b + i + kBufs - 1;
        buf[(i + kBufs - 1) % kBufs] = load(nxt < nb ? nxt : nb - 1);
        chain(buf[i]);
      }
    }
  }
  return acc;
#else
  const float* px1 = rec + fx1; const float* py1 = rec + fy1; const float* px2 = rec + fx2; const float* py2 = rec + fy2;
  // two groups of kU records in flight: the NEXT group's loads are issued before this group's dependent adds run
  constexpr int kU = LSM2D_SEQ_UNROLL;
  float ax1[kU], ay1[kU], ax2[kU], ay2[kU], bx1[kU], by1[kU], bx2[kU], by2[kU];
  auto load = [&](int k0, float (&x1)[kU], float (&y1)[kU], float (&x2)[kU], float (&y2)[kU]) {
#pragma unroll
    for (int j = 0; j < kU; ++j) { x1[j] = px1[(k0 + j) * kSeqFields]; y1[j] = py1[(k0 + j) * kSeqFields]; x2[j] = px2[(k0 + j) * kSeqFields]; y2[j] = py2[(k0 + j) * kSeqFields]; }
  };
  auto add = [&](const float (&x1)[kU], const float (&y1)[kU], const float (&x2)[kU], const float (&y2)[kU]) {
#pragma unroll
    for (int j = 0; j < kU; ++j) { acc = __builtin_fmaf(x1[j], y1[j], acc); acc = __builtin_fmaf(x2[j], y2[j], acc); }
  };
  int k = 0;
  if (n >= kU) {
    load(0, ax1, ay1, ax2, ay2);
    for (; k + 3 * kU <= n; k += 2 * kU) {      // A holds group k; B <- k + kU; add A; A <- k + 2 kU; add B
      load(k + kU, bx1, by1, bx2, by2);
      add(ax1, ay1, ax2, ay2);
      load(k + 2 * kU, ax1, ay1, ax2, ay2);
      add(bx1, by1, bx2, by2);
    }
    if (k + 2 * kU <= n) { load(k + kU, bx1, by1, bx2, by2); add(ax1, ay1, ax2, ay2); add(bx1, by1, bx2, by2); k += 2 * kU; }
    else { add(ax1, ay1, ax2, ay2); k += kU; }
  }
  for (; k < n; ++k) { acc = __builtin_fmaf(px1[k * kSeqFields], py1[k * kSeqFields], acc); acc = __builtin_fmaf(px2[k * kSeqFields], py2[k * kSeqFields], acc); }
  return acc;
#endif
}
// the walk's result for lane q < 11 of the walking wave: the running sum of quantity q (every lane of the wave calls)
LSM2D_DEV float seq_total(float acc, int lane) {
#if LSM2D_SEQ_WALK_QUADS
  const int src = lane < 11 ? 4 * lane + 3 : lane;
  return __int_as_float(__builtin_amdgcn_ds_bpermute(4 * src, __float_as_int(acc)));
#else
  return acc;
#endif
}

// ---- wave64 / workgroup reduction: shuffle butterfly, one LDS hop, fixed order => deterministic ----
static constexpr int kAccumWords = 14;

// Wave64 sum with DPP row operations (no LDS traffic): inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), then
// lane 15 -> lanes 16..31 / lane 47 -> lanes 48..63 (row_bcast:15, row_mask 0xA) and lane 31 -> lanes 32..63 (row_bcast:31,
// row_mask 0xC).  The total ends up in LANE 63.  Lanes without a source keep the identity 0.  Six v_add with DPP operands
// per value; the ds_bpermute-based __shfl_xor butterfly this replaces cost 2.4 us for the 14 sums of one slice-iteration.
LSM2D_DEV float wave_sum63(float v) {
#define LSM2D_DPP_ADD_F(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, false))
  LSM2D_DPP_ADD_F(0x111, 0xF); LSM2D_DPP_ADD_F(0x112, 0xF); LSM2D_DPP_ADD_F(0x114, 0xF); LSM2D_DPP_ADD_F(0x118, 0xF);
  LSM2D_DPP_ADD_F(0x142, 0xA); LSM2D_DPP_ADD_F(0x143, 0xC);
#undef LSM2D_DPP_ADD_F
  return v;
}
LSM2D_DEV int wave_sum63_i(int v) {
#define LSM2D_DPP_ADD_I(ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xF, false)
  LSM2D_DPP_ADD_I(0x111, 0xF); LSM2D_DPP_ADD_I(0x112, 0xF); LSM2D_DPP_ADD_I(0x114, 0xF); LSM2D_DPP_ADD_I(0x118, 0xF);
  LSM2D_DPP_ADD_I(0x142, 0xA); LSM2D_DPP_ADD_I(0x143, 0xC);
#undef LSM2D_DPP_ADD_I
  return v;
}

// wave_sum63's tree for N values at once, level by level: independent instructions back to back (no wait states between a VALU write and
// the DPP read of it), and the two levels across rows as ONE instruction per value -- v_add with a DPP operand and a row mask leaves the
// masked-out rows' registers as they are (the builtin's form -- zero into those rows, then add -- costs a v_mov_dpp and a v_add each).
// The same sums: rows 1 and 3 add lane 15 of the row below, then rows 2 and 3 add lane 31.  (s_nop: a DPP operand written by the VALU
// instruction right before needs two wait states; inside the block the other values' instructions stand between a value's two levels.)
template <int N> LSM2D_DEV void wave_tree63(float (&f)[11]) {
  static_assert(N == 10 || N == 11, "one block of ten or eleven values");
#define LSM2D_LVL(ctrl, rmask) _Pragma("unroll") for (int k = 0; k < N; ++k) f[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f[k]), ctrl, rmask, 0xF, false));
  LSM2D_LVL(0x111, 0xF) LSM2D_LVL(0x112, 0xF) LSM2D_LVL(0x114, 0xF) LSM2D_LVL(0x118, 0xF)
#undef LSM2D_LVL
#define LSM2D_ROW(k, what) "v_add_f32_dpp %" #k ", %" #k ", %" #k " " what "\n\t"
#define LSM2D_ROWS10(what) LSM2D_ROW(0, what) LSM2D_ROW(1, what) LSM2D_ROW(2, what) LSM2D_ROW(3, what) LSM2D_ROW(4, what) LSM2D_ROW(5, what) \
                           LSM2D_ROW(6, what) LSM2D_ROW(7, what) LSM2D_ROW(8, what) LSM2D_ROW(9, what)
  if constexpr (N == 11)
    asm volatile("s_nop 1\n\t" LSM2D_ROWS10("row_bcast:15 row_mask:0xa") LSM2D_ROW(10, "row_bcast:15 row_mask:0xa")
                 LSM2D_ROWS10("row_bcast:31 row_mask:0xc") LSM2D_ROW(10, "row_bcast:31 row_mask:0xc")
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]));
  else
    asm volatile("s_nop 1\n\t" LSM2D_ROWS10("row_bcast:15 row_mask:0xa") LSM2D_ROWS10("row_bcast:31 row_mask:0xc")
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]));
#undef LSM2D_ROWS10
#undef LSM2D_ROW
}
// a wave's total of a small per-lane count (at most 2^bits - 1): a ballot and a scalar popcount per bit
LSM2D_DEV int wave_count(int v, int bits) {
  int n = 0;
  for (int b = 0; b < bits; ++b) n += __builtin_popcountll(__ballot((v >> b) & 1)) << b;
  return n;
}

// all threads call; afterwards `red` (LDS, nwaves*kAccumWords words) holds per-wave totals.
// count_bits: bits a thread's pair count can occupy (it accumulates at most ceil(cols / block) pairs); 0: unknown, counts go through the adder tree
LSM2D_DEV void block_reduce_store(const Accum& A, float* red, int tid, int count_bits = 0) {
  const int lane = tid & 63, wave = tid >> 6;
  float f[11] = {A.h00, A.h01, A.h02, A.h11, A.h12, A.h22, A.b0, A.b1, A.b2, A.chi_in, A.chi_out};
  wave_tree63<11>(f);
  int i0, i1, i2;
  if (count_bits > 0) { i0 = wave_count(A.n_in, count_bits); i2 = wave_count(A.n_corr, count_bits); i1 = i2 - i0; }      // n_out == n_corr - n_in in every lane
  else { i0 = wave_sum63_i(A.n_in); i1 = wave_sum63_i(A.n_out); i2 = wave_sum63_i(A.n_corr); }
  if (lane == 63) {
    float* r = red + wave * kAccumWords;
#pragma unroll
    for (int k = 0; k < 11; ++k) r[k] = f[k];
    r[11] = __int_as_float(i0); r[12] = __int_as_float(i1); r[13] = __int_as_float(i2);
  }
}
// one thread: sum the per-wave totals in wave order
LSM2D_DEV void block_reduce_gather(const float* red, int nwaves, Accum& A) {
  accum_zero(A);
  for (int w = 0; w < nwaves; ++w) {
    const float* r = red + w * kAccumWords;
    A.h00 += r[0]; A.h01 += r[1]; A.h02 += r[2]; A.h11 += r[3]; A.h12 += r[4]; A.h22 += r[5];
    A.b0 += r[6]; A.b1 += r[7]; A.b2 += r[8]; A.chi_in += r[9]; A.chi_out += r[10];
    A.n_in += __float_as_int(r[11]); A.n_out += __float_as_int(r[12]); A.n_corr += __float_as_int(r[13]);
  }
}

// a lane's own share of the totals: lane k < 11 adds the k-th float quantity over the waves (wave order, from 0 -- the same bits as
// block_reduce_gather), lanes 11..13 the integer counts.  Call from every lane of wave 0.
LSM2D_DEV void block_reduce_gather_lane(const float* red, int nwaves, int lane, float& v, int& vi) {
  v = 0.0f; vi = 0;
  if (lane < 11) { for (int w = 0; w < nwaves; ++w) v += red[w * kAccumWords + lane]; }
  else if (lane < kAccumWords) { for (int w = 0; w < nwaves; ++w) vi += __float_as_int(red[w * kAccumWords + lane]); }
}

// the same totals for lane 0 of wave 0, gathered in parallel: lanes 0..13 each add one quantity over the waves (same wave
// order, hence the same bits as block_reduce_gather), lane 0 collects them with v_readlane.  Call from every lane of wave 0.
LSM2D_DEV void block_reduce_gather_wave0(const float* red, int nwaves, int lane, Accum& A) {
  float v = 0.0f; int vi = 0;
  if (lane < 11) { for (int w = 0; w < nwaves; ++w) v += red[w * kAccumWords + lane]; }
  else if (lane < kAccumWords) { for (int w = 0; w < nwaves; ++w) vi += __float_as_int(red[w * kAccumWords + lane]); }
  const int b = __float_as_int(v);
#define LSM2D_RL_F(k) __int_as_float(__builtin_amdgcn_readlane(b, k))
  A.h00 = LSM2D_RL_F(0); A.h01 = LSM2D_RL_F(1); A.h02 = LSM2D_RL_F(2); A.h11 = LSM2D_RL_F(3); A.h12 = LSM2D_RL_F(4); A.h22 = LSM2D_RL_F(5);
  A.b0 = LSM2D_RL_F(6); A.b1 = LSM2D_RL_F(7); A.b2 = LSM2D_RL_F(8); A.chi_in = LSM2D_RL_F(9); A.chi_out = LSM2D_RL_F(10);
#undef LSM2D_RL_F
  A.n_in = __builtin_amdgcn_readlane(vi, 11); A.n_out = __builtin_amdgcn_readlane(vi, 12); A.n_corr = __builtin_amdgcn_readlane(vi, 13);
}

// ---- step: (H + damping I) dx = -b in fp64 (LDL^T), X <- X * v2t(dx) ------------------------------
// returns false on a non-positive / non-finite pivot (SingularH)
LSM2D_DEV bool solve_update(const float H[9], const float b[3], float damping, float pose[3]) {
  const double a00 = (double) H[0] + (double) damping, a01 = H[1], a02 = H[2];
  const double a11 = (double) H[4] + (double) damping, a12 = H[5], a22 = (double) H[8] + (double) damping;
  const double r0 = -(double) b[0], r1 = -(double) b[1], r2 = -(double) b[2];
  const double d0 = a00;
  if (!(d0 > 0) || !__builtin_isfinite(d0)) return false;
  const double l10 = a01 / d0, l20 = a02 / d0;
  const double d1 = a11 - l10 * a01;
  if (!(d1 > 0) || !__builtin_isfinite(d1)) return false;
  const double l21 = (a12 - l20 * a01) / d1;
  const double d2 = a22 - l20 * a02 - l21 * l21 * d1;
  if (!(d2 > 0) || !__builtin_isfinite(d2)) return false;
  const double y0 = r0, y1 = r1 - l10 * y0, y2 = r2 - l20 * y0 - l21 * y1;
  const double z2 = y2 / d2;
  const double z1 = y1 / d1 - l21 * z2;
  const double z0 = y0 / d0 - l10 * z1 - l20 * z2;
  if (!__builtin_isfinite(z0) || !__builtin_isfinite(z1) || !__builtin_isfinite(z2)) return false;
  const float dx = (float) z0, dy = (float) z1, dth = (float) z2;
  float s, c; sincos_fixed(pose[2], s, c);
  const float nx = __builtin_fmaf(c, dx, __builtin_fmaf(-s, dy, pose[0]));
  const float ny = __builtin_fmaf(s, dx, __builtin_fmaf(c, dy, pose[1]));
  pose[0] = nx; pose[1] = ny; pose[2] = wrap_angle(pose[2] + dth);
  return true;
}

// v2t(a) * v2t(b) -> t2v, with cos/sin of a given (host- or device-computed)
LSM2D_DEV void compose(float ca, float sa, const float a[3], const float b[3], float out[3]) {
  out[0] = __builtin_fmaf(ca, b[0], __builtin_fmaf(-sa, b[1], a[0]));
  out[1] = __builtin_fmaf(sa, b[0], __builtin_fmaf(ca, b[1], a[1]));
  out[2] = wrap_angle(a[2] + b[2]);
}

}  // namespace lsm2d
