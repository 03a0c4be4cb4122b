/*
 * lsm2d_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see lsm2d_oracle.h).
 * Instantiates lsm2d_oracle_impl.inc for fp32 (mirror) and fp64 (truth), defines the fixed
 * polynomial atan2 and the pthread batch driver used as bench.py's cpu_baseline ("port").
 * Build: make -C oracle   (gcc -O3 -ffp-contract=off, no fast-math)
 */
#include "lsm2d_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* atan2 as a FIXED operation sequence (IEEE div, fmaf Horner, exact octant fix-ups) so the column
 * index floor(K00*atan2+K01) is bit-identical on CPU and GPU.  atan(a) = a + a*s*P(s), s = a*a,
 * a in [0,1]; coefficients from tools/fit_atan.py (degree 7, max abs error 7.3e-8 rad on [0,1],
 * i.e. < 3e-7 rad after the fix-ups: 5e-5 of a 1081-column bin). Stands in for the libm atan2f the
 * upstream projector calls; the two differ by a few ULP. */
static const float LSMO_ATAN_C[8] = {
  -3.333298564e-01f, 1.999039650e-01f, -1.418597102e-01f, 1.057391763e-01f,
  -7.366676629e-02f, 4.112152755e-02f, -1.513234153e-02f, 2.622197615e-03f};
#define LSMO_PI_F      3.14159274101257324f
#define LSMO_HALF_PI_F 1.57079637050628662f

float lsmo_atan2f(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
  float r = 0.0f;
  if (mx > 0.0f) {
    const float a = mn / mx;
    const float s = a * a;
    float p = LSMO_ATAN_C[7];
    for (int i = 6; i >= 0; --i) p = fmaf(p, s, LSMO_ATAN_C[i]);
    r = fmaf(a * s, p, a);
  }
  if (ay > ax) r = LSMO_HALF_PI_F - r;
  if (x < 0.0f) r = LSMO_PI_F - r;
  return copysignf(r, y);      /* atan2(-0, x<0) = -pi, as libm */
}

/* ---- fp32 mirror --------------------------------------------------------------------------- */
#define REAL float
#define SFX(n) n##_f
#define R_FMA(a, b, c) fmaf((a), (b), (c))
#define R_SQRT(a) sqrtf(a)
#define R_COS(a) cosf(a)
#define R_SIN(a) sinf(a)
#define R_ATAN2(y, x) lsmo_atan2f((y), (x))
#define R_LOG(a) logf(a)
#define R_FABS(a) fabsf(a)
#define R_FLOOR(a) floorf(a)
#define R_MAX FLT_MAX
#define R_TINY FLT_MIN
#define R_PI LSMO_PI_F
#define R_TWO_PI 6.28318548202514648f
#include "lsm2d_oracle_impl.inc"
#undef REAL
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

/* ---- batch driver (cpu_baseline): static block partition over pthreads ---------------------- */
typedef struct {
  const lsmo_aligner_params* ap; const lsmo_slice_params* sp;
  const lsmo_point* fixed_packed; const int* offs; const lsmo_point* moving; int n_moving;
  const float* x0; float* x_out; float* H_out; int* status; lsmo_iter_stats* last;
  int begin, end;
} batch_job;

static void* batch_worker(void* arg) {
  batch_job* j = (batch_job*) arg;
  lsmo_iter_stats* st = (lsmo_iter_stats*) malloc(sizeof(lsmo_iter_stats) * (size_t) (j->ap->max_iterations > 0 ? j->ap->max_iterations : 1));
  for (int i = j->begin; i < j->end; ++i) {
    const lsmo_point* f = j->fixed_packed + j->offs[i];
    const int nf = j->offs[i + 1] - j->offs[i];
    int its = 0;
    j->status[i] = lsmo_align_f(j->ap, 1, j->sp, &f, &nf, &j->moving, &j->n_moving, j->x0 + 3 * i,
                                j->x_out + 3 * i, j->H_out + 9 * i, st, &its);
    if (j->last) { if (its > 0) j->last[i] = st[its - 1]; else memset(&j->last[i], 0, sizeof(lsmo_iter_stats)); }
  }
  free(st);
  return NULL;
}

int lsmo_align_batch_f(const lsmo_aligner_params* ap, const lsmo_slice_params* sp,
                       const lsmo_point* fixed_packed, const int* fixed_offsets, int n_alignments,
                       const lsmo_point* moving, int n_moving, const float* x0, float* x_out,
                       float* H_out, int* status_out, lsmo_iter_stats* last_stats, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_alignments) n_threads = n_alignments > 0 ? n_alignments : 1;
  pthread_t* th = (pthread_t*) malloc(sizeof(pthread_t) * (size_t) n_threads);
  batch_job* jobs = (batch_job*) malloc(sizeof(batch_job) * (size_t) n_threads);
  for (int t = 0; t < n_threads; ++t) {
    batch_job j = {ap, sp, fixed_packed, fixed_offsets, moving, n_moving, x0, x_out, H_out, status_out, last_stats,
                   (int) ((long long) n_alignments * t / n_threads), (int) ((long long) n_alignments * (t + 1) / n_threads)};
    jobs[t] = j;
    if (n_threads == 1) batch_worker(&jobs[t]);
    else pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
  }
  if (n_threads > 1) for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  free(th); free(jobs);
  return LSMO_SUCCESS;
}
