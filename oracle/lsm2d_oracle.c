/*
 * lsm2d_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see lsm2d_oracle.h).
 * Instantiates lsm2d_oracle_impl.inc for fp32 (mirror) and fp64 (truth), defines the fixed
 * polynomial atan2 and the pthread batch driver used as bench.py's cpu_baseline ("port").
 * Build: make -C oracle   (gcc -O3 -ffp-contract=off, no fast-math)
 */
#define _POSIX_C_SOURCE 200809L      /* clock_gettime under -std=c11 */
#include "lsm2d_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* The bearing of a point as a FIXED operation sequence (IEEE sqrt, IEEE divide, fmaf Horner, exact octant fix-ups) so the column
 * index floor(K00*theta+K01) is bit-identical on CPU and GPU.  With r = sqrtf(x*x + y*y) -- the depth the projector needs anyway --
 * the octant angle is phi = asin(t), t = min(|x|,|y|) / r in [0, sqrt(1/2)]: asin(t) = t + t*s*P(s), s = t*t; coefficients from
 * tools/fit_asin.py (degree 6, max abs error 4.6e-8 rad, i.e. ~1e-7 rad after the fix-ups: 2e-5 of a 1081-column bin).  The HIP
 * kernels reach the same r and t from ONE v_rsq_f32 by sequences proved correctly rounded on the card (tools/fp_exact_check.hip).
 * Stands in for the libm atan2f the upstream projector calls; the two differ by a few ULP (PARITY.md section 5).
 * (Round 1 used atan(min/max) with a degree-7 polynomial: one more transcendental per point on the device.) */
static const float LSMO_ASIN_C[7] = {
  1.666723490e-01f, 7.478348911e-02f, 4.762428626e-02f, 1.043075230e-02f, 9.340071678e-02f, -1.153038889e-01f, 1.237212196e-01f};
/* the per-pair term of lsmo_iter_stats.pair_digest: plain 32-bit integer arithmetic (restated here on its own; the tests hold it against the
 * device's digest) */
unsigned long long lsmo_pair_hash(unsigned int slice, unsigned int f, unsigned int m) {
  const unsigned int a = f * 0x9E3779B1u, b = (m ^ (slice * 0x632BE5ABu)) * 0x85EBCA77u;
  unsigned int lo = a ^ ((b << 13) | (b >> 19)), hi = b ^ ((a << 19) | (a >> 13));
  lo += ((lo << 17) | (lo >> 15)) ^ b;
  hi += ((hi << 11) | (hi >> 21)) ^ a;
  return ((unsigned long long) hi << 32) | (unsigned long long) lo;
}

#define LSMO_PI_F      3.14159274101257324f
#define LSMO_HALF_PI_F 1.57079637050628662f

float lsmo_atan2f(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mn = ax > ay ? ay : ax;
  const float r2 = fmaf(x, x, y * y);       /* the projector's own r2 (lsm2d_oracle_impl.inc: project_cells) */
  float r = 0.0f;
  if (r2 >= 1e-30f && r2 <= 3e38f) {        /* every point the range gate lets through (ranges are clamped to [1e-15, 1e18] m) */
    const float rr = sqrtf(r2);
    float t = mn / rr;
    /* The device's short quotient sequence (csrc/lsm2d_device.h: div_by_depth) is the correctly rounded mn / r for every input of the
     * range gate EXCEPT the exact ties its last fused step cannot see: r with an all-ones mantissa (2 - ulp, scaled) and mn a power of
     * two, where it returns the float below (tools/fp_exact_check.hip enumerates all 2^47 mantissa pairs on the card and checks this
     * very rule).  The restatement follows it: one ulp of the sine (3e-8 rad) on inputs of probability ~1e-14 per point. */
    { unsigned int rb, nb; memcpy(&rb, &rr, 4); memcpy(&nb, &mn, 4);
      if ((rb & 0x7FFFFFu) == 0x7FFFFFu && (nb & 0x7FFFFFu) == 0u && mn >= 1e-12f) { unsigned int tb; memcpy(&tb, &t, 4); tb -= 1u; memcpy(&t, &tb, 4); } }
    const float s = t * t;
    float p = LSMO_ASIN_C[6];
    for (int i = 5; i >= 0; --i) p = fmaf(p, s, LSMO_ASIN_C[i]);
    r = fmaf(t * s, p, t);
  } else if (ax > 0.0f || ay > 0.0f) {      /* outside the gate nothing reads the bearing: any sane value (scale-free, through double) */
    r = (float) asin((double) mn / hypot((double) x, (double) y));
  }
  if (ay > ax) r = LSMO_HALF_PI_F - r;
  if (x < 0.0f) r = LSMO_PI_F - r;
  return copysignf(r, y);      /* atan2(-0, x<0) = -pi, as libm */
}

/* cos / sin of a pose angle as a FIXED operation sequence, for the same reason as atan2: the rotation of a pose must have the
 * same bits on the CPU and on the GPU, and the two libms (glibc, ROCm's ocml) differ in the last place now and then -- one
 * flipped bit in a rotation moves a point across a column edge of the z-buffer about once in 300 alignments.
 * k = rint(x * 2/pi); r = x - k*pi/2 in two fma steps; degree-2 polynomials in r^2 on [-pi/4, pi/4] (tools/fit_sincos.py:
 * max abs error 9.3e-8 for |x| <= 3000 rad, i.e. about 1 ulp of a value near 1); quadrant by k mod 4.  Stands in for the
 * cos/sin Eigen's Rotation2D calls upstream. */
void lsmo_sincosf(float x, float* sn, float* cs) {
  const float kf = rintf(x * 6.3661974669e-01f);
  float r = fmaf(-kf, 1.5707963705e+00f, x);
  r = fmaf(-kf, -4.3711388287e-08f, r);
  const float z = r * r;
  float ps = -1.9495635934e-04f; ps = fmaf(ps, z, 8.3319786936e-03f); ps = fmaf(ps, z, -1.6666650772e-01f);
  const float s = fmaf(r * z, ps, r);
  float pc = 2.4438450055e-05f; pc = fmaf(pc, z, -1.3887367677e-03f); pc = fmaf(pc, z, 4.1666645557e-02f);
  const float c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  const int q = (int) kf & 3;
  *sn = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
  *cs = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}
/* log of a positive normal number as a FIXED operation sequence (the Cauchy kernel's statistic chi_out = tau * log(1 + chi/tau) was
 * the last libm call on the path): x = m * 2^e with m in [sqrt(1/2), sqrt(2)), f = m - 1, log x = e*ln2 + f - f^2/2 + f^3 P(f);
 * degree-7 P from tools/fit_log.py, max relative error 1.3e-7. */
float lsmo_logf_fixed(float x) {
  unsigned int bits; memcpy(&bits, &x, 4);
  int e = (int) (bits >> 23) - 127;
  unsigned int mb = (bits & 0x7FFFFFu) | 0x3F800000u;
  float m; memcpy(&m, &mb, 4);
  if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
  const float f = m - 1.0f, z = f * f;
  float p = -7.6311752200e-02f;
  p = fmaf(p, f, 1.2854909897e-01f); p = fmaf(p, f, -1.3207894564e-01f); p = fmaf(p, f, 1.4190009236e-01f);
  p = fmaf(p, f, -1.6616521776e-01f); p = fmaf(p, f, 2.0001615584e-01f); p = fmaf(p, f, -2.5001060963e-01f);
  p = fmaf(p, f, 3.3333328366e-01f);
  const float r = fmaf(z * f, p, fmaf(-0.5f, z, f));
  return fmaf((float) e, 6.9314718246e-01f, r);
}
static float lsmo_cosf_fixed(float a) { float s, c; lsmo_sincosf(a, &s, &c); return c; }
static float lsmo_sinf_fixed(float a) { float s, c; lsmo_sincosf(a, &s, &c); return s; }

/* ---- fp32 mirror --------------------------------------------------------------------------- */
#define REAL float
#define LSMO_FLOAT_PASS 1
#define SFX(n) n##_f
#define R_FMA(a, b, c) fmaf((a), (b), (c))
#define R_SQRT(a) sqrtf(a)
#define R_COS(a) lsmo_cosf_fixed(a)
#define R_SIN(a) lsmo_sinf_fixed(a)
#define R_ATAN2(y, x) lsmo_atan2f((y), (x))
#define R_LOG(a) lsmo_logf_fixed(a)
#define R_FABS(a) fabsf(a)
#define R_FLOOR(a) floorf(a)
#define R_MAX FLT_MAX
#define R_TINY FLT_MIN
#define R_PI LSMO_PI_F
#define R_TWO_PI 6.28318548202514648f
#include "lsm2d_oracle_impl.inc"
#undef REAL
#undef LSMO_FLOAT_PASS
#undef SFX
#undef R_FMA
#undef R_SQRT
#undef R_COS
#undef R_SIN
#undef R_ATAN2
#undef R_LOG
#undef R_FABS
#undef R_FLOOR
#undef R_MAX
#undef R_TINY
#undef R_PI
#undef R_TWO_PI

/* ---- fp32 in the reference's own arithmetic ------------------------------------------------------
 * libm calls where the reference (Eigen's Rotation2D, the upstream polar projector, the Cauchy robustifier) calls libm, every
 * product and sum rounded on its own (no FMA contraction: the reference is plain x86-64 C++), Eigen's association for an
 * isometry applied to a point.  Not bit-reproducible across libms -- it is a measuring stick, not a mirror. */
#define REAL float
#define LSMO_REFERENCE_PASS 1
#define SFX(n) n##_r
#define R_FMA(a, b, c) ((a) * (b) + (c))
#define R_SQRT(a) sqrtf(a)
#define R_COS(a) cosf(a)
#define R_SIN(a) sinf(a)
#define R_ATAN2(y, x) atan2f((y), (x))
#define R_LOG(a) logf(a)
#define R_FABS(a) fabsf(a)
#define R_FLOOR(a) floorf(a)
#define R_MAX FLT_MAX
#define R_TINY FLT_MIN
#define R_PI LSMO_PI_F
#define R_TWO_PI 6.28318548202514648f
#include "lsm2d_oracle_impl.inc"
#undef REAL
#undef LSMO_REFERENCE_PASS
#undef SFX
#undef R_FMA
#undef R_SQRT
#undef R_COS
#undef R_SIN
#undef R_ATAN2
#undef R_LOG
#undef R_FABS
#undef R_FLOOR
#undef R_MAX
#undef R_TINY
#undef R_PI
#undef R_TWO_PI

/* ---- fp64 truth ------------------------------------------------------------------------------ */
#define REAL double
#define SFX(n) n##_d
#define R_FMA(a, b, c) fma((a), (b), (c))
#define R_SQRT(a) sqrt(a)
#define R_COS(a) cos(a)
#define R_SIN(a) sin(a)
#define R_ATAN2(y, x) atan2((y), (x))
#define R_LOG(a) log(a)
#define R_FABS(a) fabs(a)
#define R_FLOOR(a) floor(a)
#define R_MAX DBL_MAX
#define R_TINY ((double) FLT_MIN)
#define R_PI 3.14159265358979323846
#define R_TWO_PI 6.28318530717958647692
#include "lsm2d_oracle_impl.inc"

void lsmo_error_jacobian_d(const lsmo_point* f, const lsmo_point* m, const double pose[3],
                           double e[3], double J[9]) {
  iso_d T = v2t_d(pose);
  double a[3], d[2];
  err_jac_d(&T, f, m, e, a, d);
  J[0] = a[0]; J[1] = a[1]; J[2] = a[2];
  J[3] = 0; J[4] = 0; J[5] = d[0];
  J[6] = 0; J[7] = 0; J[8] = d[1];
}

/* ---- RawDataPreprocessorProjective2D (fp32 only: it is the producer of fp32 clouds) --------------------------
 * Assumptions about the un-vendored upstream pieces (each is what the in-tree call site constrains, nothing more):
 * F2.1 unprojector: beam c is valid iff range_min <= range <= range_max; its bearing is (c - n/2) * sensor_res with
 *      sensor_res = (angle_max - angle_min) / n -- the inverse of the sensor matrix [[1/sensor_res, n/2]] the
 *      reference installs (.cpp:87-90: the centre column, not angle_min, anchors the bearings); point =
 *      range * (cos, sin); valid points are appended in beam order (back_insert_iterator, .cpp:29-30).
 * F2.2 sliding-window normals: the window of point i grows from i to both sides over consecutive points while
 *      |p_j - p_i|^2 <= normal_point_distance^2; fewer than normal_min_points points -> the point is dropped;
 *      otherwise normal = eigenvector of the smallest eigenvalue of the window scatter matrix (closed form
 *      below), flipped to face the sensor (n . p <= 0).
 * F2.3 voxelize(res, res, 1, 1) (.cpp:38-41): key = floor of each of (x/res, y/res, nx, ny); points with the same key
 *      are averaged (normal re-normalised); voxels come out in ascending lexicographic key order. */
typedef struct { long long key; int idx; } lsmo_vox;
static int lsmo_vox_cmp(const void* a, const void* b) {
  const lsmo_vox* x = (const lsmo_vox*) a; const lsmo_vox* y = (const lsmo_vox*) b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx - y->idx;
}
/* voxel key packed in 64 bits: (kx+2^15) | (ky+2^15) | (knx+16) | (kny+16); returns -1 when out of the packable range.
 * inv_rn = 1 / (coefficient of the normal components): 1 for the preprocessor (.cpp:40), 10 for the clipper (coefficients
 * res, res, 0.1, 0.1: mapping/scene_clipper_projective_2d.cpp:46) */
static long long lsmo_vox_key(float x, float y, float nx, float ny, float inv_res, float inv_rn) {
  const float kx = floorf(x * inv_res), ky = floorf(y * inv_res), knx = floorf(nx * inv_rn), kny = floorf(ny * inv_rn);
  if (!(kx >= -32768.0f && kx < 32768.0f && ky >= -32768.0f && ky < 32768.0f && knx >= -16.0f && knx <= 15.0f && kny >= -16.0f && kny <= 15.0f)) return -1;
  return ((long long) ((int) kx + 32768) << 26) | ((long long) ((int) ky + 32768) << 10) | ((long long) ((int) knx + 16) << 5) | (long long) ((int) kny + 16);
}
/* F2.3: PointCloud::voxelize over k points given as four arrays; returns the number of voxels written to out */
static int lsmo_voxelize(const float* ox, const float* oy, const float* onx, const float* ony, int k, float inv_res, float inv_rn, lsmo_point* out) {
  lsmo_vox* vx = (lsmo_vox*) malloc(sizeof(lsmo_vox) * (size_t) (k > 0 ? k : 1));
  int nv = 0, n_out = 0;
  for (int i = 0; i < k; ++i) { const long long key = lsmo_vox_key(ox[i], oy[i], onx[i], ony[i], inv_res, inv_rn); if (key >= 0) { vx[nv].key = key; vx[nv].idx = i; ++nv; } }
  qsort(vx, (size_t) nv, sizeof(lsmo_vox), lsmo_vox_cmp);
  for (int b = 0; b < nv;) {
    int e = b; float ax = 0.0f, ay = 0.0f, anx = 0.0f, any_ = 0.0f;
    while (e < nv && vx[e].key == vx[b].key) { const int i = vx[e].idx; ax += ox[i]; ay += oy[i]; anx += onx[i]; any_ += ony[i]; ++e; }
    const float inv = 1.0f / (float) (e - b);
    ax *= inv; ay *= inv; anx *= inv; any_ *= inv;
    const float nn = sqrtf(fmaf(anx, anx, any_ * any_));
    if (nn > 0.0f) { anx = anx / nn; any_ = any_ / nn; }
    out[n_out].x = ax; out[n_out].y = ay; out[n_out].nx = anx; out[n_out].ny = any_; ++n_out;
    b = e;
  }
  free(vx);
  return n_out;
}

int lsmo_preprocess_scan_f(const lsmo_preprocessor* pp, const float* ranges, lsmo_point* out) {
  const int n = pp->n_beams;
  if (n <= 0 || !(pp->angle_max > pp->angle_min)) return LSMO_BAD_ARGUMENT;
  const float sensor_res = (pp->angle_max - pp->angle_min) / (float) n, k01 = (float) n * 0.5f;
  float* px = (float*) malloc(sizeof(float) * 2 * (size_t) n); float* py = px + n;
  int m = 0;
  for (int c = 0; c < n; ++c) {                                   /* F2.1 */
    const float r = ranges[c];
    if (!(r >= pp->range_min && r <= pp->range_max)) continue;
    const float a = ((float) c - k01) * sensor_res;
    px[m] = r * cosf(a); py[m] = r * sinf(a); ++m;
  }
  const float d2max = pp->normal_point_distance * pp->normal_point_distance;
  int k = 0;                                                      /* F2.2, compacting in place (k <= i) */
  float* ox = (float*) malloc(sizeof(float) * 4 * (size_t) (n > 0 ? n : 1)); float* oy = ox + n; float* onx = oy + n; float* ony = onx + n;
  for (int i = 0; i < m; ++i) {
    int lo = i, hi = i;
    while (lo > 0) { const float dx = px[lo - 1] - px[i], dy = py[lo - 1] - py[i]; if (!(fmaf(dx, dx, dy * dy) <= d2max)) break; --lo; }
    while (hi < m - 1) { const float dx = px[hi + 1] - px[i], dy = py[hi + 1] - py[i]; if (!(fmaf(dx, dx, dy * dy) <= d2max)) break; ++hi; }
    const int cnt = hi - lo + 1;
    if (cnt < pp->normal_min_points) continue;
    float sx = 0.0f, sy = 0.0f;
    for (int j = lo; j <= hi; ++j) { sx += px[j]; sy += py[j]; }
    const float inv = 1.0f / (float) cnt, mx = sx * inv, my = sy * inv;
    float sxx = 0.0f, sxy = 0.0f, syy = 0.0f;
    for (int j = lo; j <= hi; ++j) { const float dx = px[j] - mx, dy = py[j] - my; sxx = fmaf(dx, dx, sxx); sxy = fmaf(dx, dy, sxy); syy = fmaf(dy, dy, syy); }
    const float tr = sxx + syy, df = sxx - syy;
    const float disc = sqrtf(fmaf(df, df, 4.0f * (sxy * sxy)));
    const float lmin = 0.5f * (tr - disc);
    float v1x = sxy, v1y = lmin - sxx, v2x = lmin - syy, v2y = sxy;
    const float n1 = fmaf(v1x, v1x, v1y * v1y), n2 = fmaf(v2x, v2x, v2y * v2y);
    float vx = v1x, vy = v1y, nn = n1;
    if (n2 > n1) { vx = v2x; vy = v2y; nn = n2; }
    if (!(nn > 0.0f)) continue;                                   /* isotropic window: no direction */
    const float s = sqrtf(nn);
    vx = vx / s; vy = vy / s;
    if (fmaf(vx, px[i], vy * py[i]) > 0.0f) { vx = -vx; vy = -vy; }
    ox[k] = px[i]; oy[k] = py[i]; onx[k] = vx; ony[k] = vy; ++k;
  }
  int n_out = 0;
  if (!(pp->voxelize_resolution > 0.0f)) {
    for (int i = 0; i < k; ++i) { out[i].x = ox[i]; out[i].y = oy[i]; out[i].nx = onx[i]; out[i].ny = ony[i]; }
    n_out = k;
  } else {                                                        /* F2.3 */
    n_out = lsmo_voxelize(ox, oy, onx, ony, k, 1.0f / pp->voxelize_resolution, 1.0f, out);
  }
  free(px); free(ox);
  return n_out;
}

/* SceneClipperProjective2D::compute with voxelize_resolution > 0 (mapping/scene_clipper_projective_2d.cpp:36-48): the filled cells'
 * transformed points, in ascending column and still in the SENSOR frame, are voxelised with coefficients (res, res, 0.1, 0.1) --
 * assumption F2.3 with the normal components scaled by 1 / 0.1 -- and only then moved to the robot frame (:60-62). */
int lsmo_clip_scene_voxelized_f(const lsmo_projector* pr, const lsmo_point* scene, int n_scene, const float robot_in_local_map[3],
                                const float sensor_in_robot[3], float voxelize_resolution, lsmo_point* out) {
  if (!(voxelize_resolution > 0.0f)) return lsmo_clip_scene_f(pr, scene, n_scene, robot_in_local_map, sensor_in_robot, out, NULL);
  if (!pr || pr->canvas_cols <= 0) return LSMO_BAD_ARGUMENT;
  const int cols = pr->canvas_cols;
  float cam[3]; lsmo_compose_f(robot_in_local_map, sensor_in_robot, cam);
  const float ident[3] = {0.0f, 0.0f, 0.0f};
  lsmo_point* tmp = (lsmo_point*) malloc(sizeof(lsmo_point) * (size_t) cols);
  const int k = lsmo_clip_scene_f(pr, scene, n_scene, cam, ident, tmp, NULL);      /* camera at robot * sensor, points left in the sensor frame */
  if (k < 0) { free(tmp); return k; }
  float* ox = (float*) malloc(sizeof(float) * 4 * (size_t) (k > 0 ? k : 1)); float* oy = ox + k; float* onx = oy + k; float* ony = onx + k;
  for (int i = 0; i < k; ++i) { ox[i] = tmp[i].x; oy[i] = tmp[i].y; onx[i] = tmp[i].nx; ony[i] = tmp[i].ny; }
  const int n = lsmo_voxelize(ox, oy, onx, ony, k, 1.0f / voxelize_resolution, 1.0f / 0.1f, out);
  if (!(sensor_in_robot[0] == 0.0f && sensor_in_robot[1] == 0.0f && sensor_in_robot[2] == 0.0f)) {
    iso_f S = v2t_f(sensor_in_robot);
    for (int i = 0; i < n; ++i) {
      float x, y, nx, ny;
      xf_point_f(&S, out[i].x, out[i].y, &x, &y); xf_normal_f(&S, out[i].nx, out[i].ny, &nx, &ny);
      out[i].x = x; out[i].y = y; out[i].nx = nx; out[i].ny = ny;
    }
  }
  free(ox); free(tmp);
  return n;
}

/* ---- batch driver (cpu_baseline): pthreads over ONE atomic work counter ------------------------------------------------------------
 * Every thread takes the next alignment that nobody has taken yet (round 6; a static block partition before: with a CPU quota below the thread count -- the GPU
 * boxes of this pool give a 256-thread affinity mask and a 16-CPU cgroup quota -- the block of a thread that is descheduled waits for it while others idle).
 * thread_seconds / thread_jobs (may be NULL, [n_threads]): wall time each worker spent and the alignments it did -- what shows a throttled host. */
#include <stdatomic.h>
#include <time.h>
typedef struct {
  const lsmo_aligner_params* ap; const lsmo_slice_params* sp;
  const lsmo_point* fixed_packed; const int* offs; const lsmo_point* moving; int n_moving;
  const float* x0; float* x_out; float* H_out; int* status; lsmo_iter_stats* last;
  atomic_int* next; int n; double seconds; int jobs;
} batch_job;

static double now_seconds(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec; }

static void* batch_worker(void* arg) {
  batch_job* j = (batch_job*) arg;
  const double t0 = now_seconds();
  lsmo_iter_stats* st = (lsmo_iter_stats*) malloc(sizeof(lsmo_iter_stats) * (size_t) (j->ap->max_iterations > 0 ? j->ap->max_iterations : 1) * 2);
  for (;;) {
    const int i = atomic_fetch_add_explicit(j->next, 1, memory_order_relaxed);
    if (i >= j->n) break;
    const lsmo_point* f = j->fixed_packed + j->offs[i];
    const int nf = j->offs[i + 1] - j->offs[i];
    int its = 0;
    j->status[i] = lsmo_align_f(j->ap, 1, j->sp, &f, &nf, &j->moving, &j->n_moving, j->x0 + 3 * i,
                                j->x_out + 3 * i, j->H_out + 9 * i, st, &its);
    if (j->last) { if (its > 0) j->last[i] = st[its - 1]; else memset(&j->last[i], 0, sizeof(lsmo_iter_stats)); }
    ++j->jobs;
  }
  free(st);
  j->seconds = now_seconds() - t0;
  return NULL;
}

int lsmo_align_batch_timed_f(const lsmo_aligner_params* ap, const lsmo_slice_params* sp,
                             const lsmo_point* fixed_packed, const int* fixed_offsets, int n_alignments,
                             const lsmo_point* moving, int n_moving, const float* x0, float* x_out,
                             float* H_out, int* status_out, lsmo_iter_stats* last_stats, int n_threads, double* thread_seconds, int* thread_jobs) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_alignments) n_threads = n_alignments > 0 ? n_alignments : 1;
  pthread_t* th = (pthread_t*) malloc(sizeof(pthread_t) * (size_t) n_threads);
  batch_job* jobs = (batch_job*) malloc(sizeof(batch_job) * (size_t) n_threads);
  atomic_int next; atomic_init(&next, 0);
  for (int t = 0; t < n_threads; ++t) {
    batch_job j = {ap, sp, fixed_packed, fixed_offsets, moving, n_moving, x0, x_out, H_out, status_out, last_stats, &next, n_alignments, 0.0, 0};
    jobs[t] = j;
    if (n_threads == 1) batch_worker(&jobs[t]);
    else pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
  }
  if (n_threads > 1) for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  for (int t = 0; t < n_threads; ++t) { if (thread_seconds) thread_seconds[t] = jobs[t].seconds; if (thread_jobs) thread_jobs[t] = jobs[t].jobs; }
  free(th); free(jobs);
  return LSMO_SUCCESS;
}

int lsmo_align_batch_f(const lsmo_aligner_params* ap, const lsmo_slice_params* sp,
                       const lsmo_point* fixed_packed, const int* fixed_offsets, int n_alignments,
                       const lsmo_point* moving, int n_moving, const float* x0, float* x_out,
                       float* H_out, int* status_out, lsmo_iter_stats* last_stats, int n_threads) {
  return lsmo_align_batch_timed_f(ap, sp, fixed_packed, fixed_offsets, n_alignments, moving, n_moving, x0, x_out, H_out, status_out, last_stats, n_threads, NULL, NULL);
}
