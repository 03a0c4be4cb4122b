/*
 * lsm2d_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the 2D scan-matching hot path of
 * rvp-group/srrg2_laser_slam_2d.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in un-vendored, un-pinned upstream modules
 * (srrg2_core, srrg2_solver, srrg2_slam_interfaces; reference package.xml:14-21) that are absent from
 * the build container, and the reference's own tests hold no golden vector for the finders, the
 * factor or the aligner.  This file is therefore a RESTATEMENT of the
 * reference algorithm from the in-tree sources, cited per function:
 *
 *   finder (projective)  srrg2_laser_slam_2d/src/srrg2_laser_slam_2d/registration/correspondence_finder_projective_2d.cpp:18-77
 *   finder (kd-tree/NN)  .../registration/correspondence_finder_kd_tree_2d.cpp:5-38
 *   finder (dist. map)   .../registration/correspondence_finder_nn_2d.cpp:10-97
 *   projector convention srrg2_laser_slam_2d/apps/synthetic_scene_generator.cpp:56-88
 *   factor / solver math srrg2_laser_slam_2d/octave/solver/nicp_post.m:4-26,69-97 (2-D restriction)
 *   aligner loop         srrg2_laser_slam_2d/apps/visual_test_aligner_2d.cpp:102-156 (driver),
 *                        configurations/stage_segway_double_config_MULTI.json:602-630,700-732
 *
 * Who may use this: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- as the
 * checker / the timed CPU baseline, never as (part of) the shipped product path.
 *
 * Three instantiations of every routine: `_f` = fp32 "mirror" (same IEEE operation sequence the
 * HIP kernels use, so index outputs agree bit-for-bit), `_d` = fp64 "truth", and `_r` = fp32 in the
 * REFERENCE'S OWN ARITHMETIC as far as it can be known without the upstream sources: libm atan2f / sinf /
 * cosf / logf (what Eigen and the projector call), no fused multiply-add (x86-64 builds of the reference
 * have none), Eigen's association in R p + t, sums pair after pair.  `_r` is what the HIP path is measured
 * against to show how much the fixed-polynomial arithmetic of `_f` moves z-buffer winners, pairs and poses
 * (tests/test_reference_arithmetic.py, PARITY.md section 5).
 */
#ifndef LSM2D_ORACLE_H
#define LSM2D_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* PointNormal2f payload: (x, y, nx, ny) fp32 -- 4-vector confirmed at
 * sensor_processing/raw_data_preprocessor_projective_2d.cpp:39-40 */
typedef struct { float x, y, nx, ny; } lsmo_point;

/* Correspondence(fixed_idx, moving_idx) -- ctor use at correspondence_finder_kd_tree_2d.cpp:25 */
typedef struct { int fixed_idx, moving_idx; } lsmo_corr;

/* PointNormal2fProjectorPolar parameters (synthetic_scene_generator.cpp:69-75) */
typedef struct {
  int   canvas_cols;
  float angle_min, angle_max;   /* angle_col_min / angle_col_max [rad] */
  float range_min, range_max;   /* [m] */
  float col_offset;             /* 0 = floor(K00*theta+K01) (SURVEY App. A.3 assumption); 0.5 = round-to-nearest */
} lsmo_projector;

enum { LSMO_FINDER_PROJECTIVE = 0, LSMO_FINDER_NN = 1, LSMO_FINDER_DISTMAP = 2,
       /* oracle only (the HIP library has no such finder): the KD-tree as upstream is BELIEVED to search it --
        * descent to the single leaf on the query's side, no backtracking, hence approximate; quantifies what the exact search of
        * LSMO_FINDER_NN changes (PARITY.md section 5) */
       LSMO_FINDER_KDTREE_APPROX = 3 };
enum { LSMO_ROBUST_NONE = 0, LSMO_ROBUST_CAUCHY = 1 };
enum {
  LSMO_SUCCESS = 0,
  LSMO_NOT_ENOUGH_CORRESPONDENCES = 1,
  LSMO_NOT_ENOUGH_INLIERS = 2,
  LSMO_SINGULAR_H = 3,
  LSMO_BAD_ARGUMENT = -1
};

typedef struct {
  int   finder;                       /* LSMO_FINDER_* */
  lsmo_projector projector;           /* projective finder */
  float point_distance;               /* projective: max |depth_f - depth_m|   (.h:16-20, default 0.5) */
  float normal_cos;                   /* all finders: min n_f . n_m             (default 0.8) */
  float max_distance;                 /* NN / distmap: max point distance [m]  (kd .h:23, nn .h:20-24) */
  float resolution;                   /* distmap: m / pixel                     (nn .h:25-29) */
  int   robustifier;                  /* LSMO_ROBUST_* */
  float chi_threshold;                /* Cauchy tau (MULTI.json:153-158) */
  int   min_num_correspondences;      /* slice skipped if #pairs <= this (MULTI.json:179) */
  float sensor_in_robot[3];           /* WithSensor variant (aligner_slice_processor_laser_2d_impl.cpp:7-10); (0,0,0) = plain */
  float kd_max_leaf_range;            /* LSMO_FINDER_KDTREE_APPROX: param_max_leaf_range  (correspondence_finder_kd_tree_2d.h:26-28, default 1e-2) */
  int   kd_min_leaf_points;           /*                            param_min_leaf_points (.h:29-33, default 20) */
} lsmo_slice_params;

typedef struct {
  int   max_iterations;               /* MULTI.json:711 (10), :613 (30); BASELINE 20 */
  int   min_num_inliers;              /* MULTI.json:714 */
  float damping;                      /* GN damping, MULTI.json:254-259 (0) */
  int   has_prior;                    /* odometry-prior slice (MULTI.json:402-422): e = t2v(Z^-1 X), Omega */
  float prior_z[3];
  float prior_omega[9];
  int   device_order;                 /* fp32 mirror only.  0: H, b and the statistics are summed pair after pair, as the reference's
                                         solver does.  1: they are summed in the ORDER THE HIP KERNELS USE (the pair of column c /
                                         moving point j goes to thread c mod 512 / j mod 512, threads sum their pairs in turn, a wave's
                                         64 partial sums are combined by the DPP scan tree of wave_sum63, the 8 wave totals in wave
                                         order) -- every other operation already is the same sequence on both sides, so with this
                                         switch the mirror reproduces the device's poses, H and statistics BIT FOR BIT through all
                                         iterations.  The two orders are
                                         equally valid fp32 evaluations of the same sums. */
  float termination_chi_epsilon;      /* the aligner's "termination_criteria" (MULTI.json:627-630,729-731: unset in both shipped aligners = always
                                         max_iterations).  ASSUMED (the upstream class is not in the tree; the solver's SimpleTerminationCriteria,
                                         MULTI.json:218-223, documents its epsilon as the "ratio of decay of chi2 between iteration"): 0 = off;
                                         > 0: stop after an iteration whose total chi^2 (inliers + kernelised outliers) differs from the previous
                                         iteration's by less than epsilon times itself; that iteration is still solved and applied. */
  /* MultiAligner2D's two remaining options (MULTI.json:606-610,704-708; both 0 in the shipped configurations).  The upstream class is not in the
   * tree: semantics RESTATED from the parameters' doc strings (PARITY.md section 2, [UPSTREAM-MEMORY]).
   * enable_inlier_only_runs ("toggles additional inlier only runs if sufficient inliers are available"): when the regular loop ended without a
   *   failure and its last iteration counted n_in >= min_num_inliers, a second loop of up to max_iterations iterations follows in which a pair whose
   *   factor is not an inlier under its slice's robustifier (chi^2 >= tau) contributes nothing to H and b and an inlier contributes with weight 1;
   *   finders, gates, statistics, the termination criterion (afresh) and the status rules are the regular loop's.  stats then needs room for
   *   2 * max_iterations entries.
   * keep_only_inlier_correspondences ("toggles removal of correspondences which factors are not inliers in the last iteration"): the pairs handed
   *   back by lsmo_align_pairs_* hold only the last iteration's inliers; nothing else depends on it. */
  int   enable_inlier_only_runs;
  int   keep_only_inlier_correspondences;
} lsmo_aligner_params;

typedef struct {
  int   n_corr, n_in, n_out;
  float chi_in, chi_out;
  /* order-independent digest of the iteration's correspondence set (test instrument, not part of the reference): the wrapping 64-bit sum of
   * lsmo_pair_hash(slice, fixed_idx, moving_idx) over every pair counted in n_corr, low and high word */
  unsigned int pair_digest_lo, pair_digest_hi;
} lsmo_iter_stats;
unsigned long long lsmo_pair_hash(unsigned int slice, unsigned int fixed_idx, unsigned int moving_idx);

/* ---- scalar helpers --------------------------------------------------------------------- */
float lsmo_atan2f(float y, float x);     /* the fixed-polynomial atan2 both CPU and GPU evaluate */
float lsmo_logf_fixed(float x);             /* the fixed-sequence log of the Cauchy kernel statistic (x > 0, normal) */
void  lsmo_sincosf(float x, float* sn, float* cs);     /* the fixed-sequence sin / cos of a pose angle both CPU and GPU evaluate */
void  lsmo_compose_f(const float a[3], const float b[3], float out[3]);   /* v2t(a)*v2t(b) -> t2v */
void  lsmo_inverse_f(const float a[3], float out[3]);
void  lsmo_compose_d(const double a[3], const double b[3], double out[3]);
void  lsmo_inverse_d(const double a[3], double out[3]);

/* ---- projector: one polar z-buffer pass -------------------------------- */
/* pose maps cloud points into the camera frame (= camera_pose^-1 of the reference projector).
 * out_src[cols] = winning source index or -1; out_depth[cols]; out_xyn[4*cols] = transformed point. */
int lsmo_project_f(const lsmo_projector* pr, const lsmo_point* cloud, int n, const float pose[3],
                   int* out_src, float* out_depth, float* out_xyn);
int lsmo_project_d(const lsmo_projector* pr, const lsmo_point* cloud, int n, const double pose[3],
                   int* out_src, double* out_depth, double* out_xyn);

/* ---- finders: return number of pairs written to out (capacity: cols resp. n_moving) ------- */
int lsmo_find_projective_f(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                           const lsmo_point* moving, int n_moving, const float pose[3], lsmo_corr* out);
int lsmo_find_projective_d(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                           const lsmo_point* moving, int n_moving, const double pose[3], lsmo_corr* out);
int lsmo_find_nn_f(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                   const lsmo_point* moving, int n_moving, const float pose[3], lsmo_corr* out);
int lsmo_find_nn_d(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                   const lsmo_point* moving, int n_moving, const double pose[3], lsmo_corr* out);
int lsmo_find_distmap_f(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                        const lsmo_point* moving, int n_moving, const float pose[3], lsmo_corr* out);
int lsmo_find_distmap_d(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                        const lsmo_point* moving, int n_moving, const double pose[3], lsmo_corr* out);
/* O(N_f*N_m) brute force, used only to validate the grid search above */
int lsmo_find_nn_brute_f(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                         const lsmo_point* moving, int n_moving, const float pose[3], lsmo_corr* out);

/* ---- factor + robustifier: H (row-major 3x3), b, statistics ------------- */
int lsmo_linearize_f(const lsmo_slice_params* sp, const lsmo_point* fixed, const lsmo_point* moving,
                     const lsmo_corr* corr, int n_corr, const float pose[3],
                     float H[9], float b[3], lsmo_iter_stats* st);
int lsmo_linearize_d(const lsmo_slice_params* sp, const lsmo_point* fixed, const lsmo_point* moving,
                     const lsmo_corr* corr, int n_corr, const double pose[3],
                     double H[9], double b[3], lsmo_iter_stats* st);
/* the same sums in the order a HIP launch of `threads` threads in workgroups of `block` forms them (see lsmo_aligner_params.device_order
 * and lsm2d_oracle_impl.inc): pair k -> thread (slot ? slot[k] : k) mod threads.  lsm2d_linearize: block 256, threads = 256 *
 * min(1024, ceil(n / 256)); the aligner kernels: block = threads = 512, slot = column / moving index. */
int lsmo_linearize_device_order_f(const lsmo_slice_params* sp, const lsmo_point* fixed, const lsmo_point* moving,
                                  const lsmo_corr* corr, const int* slot, int n_corr, const float pose[3], int threads, int block,
                                  float H[9], float b[3], lsmo_iter_stats* st);
/* error and Jacobian of ONE pair (for the finite-difference test) */
void lsmo_error_jacobian_d(const lsmo_point* f, const lsmo_point* m, const double pose[3],
                           double e[3], double J[9]);

/* ---- step: (H + damping I) dx = -b ; X <- X * v2t(dx) ------------------- */
int lsmo_solve_update_f(const float H[9], const float b[3], float damping, float pose[3], float dx[3]);
int lsmo_solve_update_d(const double H[9], const double b[3], double damping, double pose[3], double dx[3]);

/* ---- aligner, multi-slice + optional prior ------------------------------ */
int lsmo_align_f(const lsmo_aligner_params* ap, int n_slices, const lsmo_slice_params* sp,
                 const lsmo_point* const* fixed, const int* n_fixed,
                 const lsmo_point* const* moving, const int* n_moving,
                 const float x0[3], float x_out[3], float H_out[9],
                 lsmo_iter_stats* stats /* [max_iterations] or NULL */, int* iterations_done);
int lsmo_align_d(const lsmo_aligner_params* ap, int n_slices, const lsmo_slice_params* sp,
                 const lsmo_point* const* fixed, const int* n_fixed,
                 const lsmo_point* const* moving, const int* n_moving,
                 const double x0[3], double x_out[3], double H_out[9],
                 lsmo_iter_stats* stats, int* iterations_done);
/* the same, additionally handing back what the aligner leaves in every slice's correspondence vector (slice->correspondences(),
 * apps/visual_test_aligner_2d.cpp:129-143): the pairs of the last iteration started, finder order; with keep_only_inlier_correspondences only
 * that iteration's inliers.  out_pairs[s]: room for canvas_cols resp. n_moving[s] pairs; out_n_pairs[s]: how many.  stats: room for
 * max_iterations * (1 + enable_inlier_only_runs) entries. */
int lsmo_align_pairs_f(const lsmo_aligner_params* ap, int n_slices, const lsmo_slice_params* sp,
                       const lsmo_point* const* fixed, const int* n_fixed, const lsmo_point* const* moving, const int* n_moving,
                       const float x0[3], float x_out[3], float H_out[9], lsmo_iter_stats* stats, int* iterations_done,
                       lsmo_corr* const* out_pairs, int* out_n_pairs);
int lsmo_align_pairs_d(const lsmo_aligner_params* ap, int n_slices, const lsmo_slice_params* sp,
                       const lsmo_point* const* fixed, const int* n_fixed, const lsmo_point* const* moving, const int* n_moving,
                       const double x0[3], double x_out[3], double H_out[9], lsmo_iter_stats* stats, int* iterations_done,
                       lsmo_corr* const* out_pairs, int* out_n_pairs);
int lsmo_align_pairs_r(const lsmo_aligner_params* ap, int n_slices, const lsmo_slice_params* sp,
                       const lsmo_point* const* fixed, const int* n_fixed, const lsmo_point* const* moving, const int* n_moving,
                       const float x0[3], float x_out[3], float H_out[9], lsmo_iter_stats* stats, int* iterations_done,
                       lsmo_corr* const* out_pairs, int* out_n_pairs);

/* ---- sensor processing: RawDataPreprocessorProjective2D ----------------------------
 * sensor_processing/raw_data_preprocessor_projective_2d.cpp:13-51 (compute) and :77-104 (_processLaserMessage).
 * The three upstream pieces it chains -- PointNormal2fUnprojectorPolar::compute<WithNormals>,
 * NormalComputator1DSlidingWindow::computeNormals, PointCloud::voxelize -- are NOT in the tree; their restatement
 * here rests on the assumptions F2.1-F2.3 listed in lsm2d_oracle.c.  Pinned by the one value the reference's own
 * test holds: the `Synthetic` fixture (tests/fixtures.hpp:8-53) must yield exactly 100 points
 * (tests/test_measurement_adaptor.cpp:36). */
typedef struct {
  int   n_beams;
  float angle_min, angle_max;          /* LaserMessage angle_min / angle_max */
  float range_min, range_max;          /* max(msg, param) / min(msg, param), .cpp:83-84 */
  float normal_point_distance;         /* NormalComputator1DSlidingWindow (MULTI.json:845-853: 0.3) */
  int   normal_min_points;             /* (5) */
  float voxelize_resolution;           /* .h:41-45 default 0.02; <= 0: keep every valid point */
} lsmo_preprocessor;
/* out: capacity n_beams points; returns the number of points (>= 0) */
int lsmo_preprocess_scan_f(const lsmo_preprocessor* pp, const float* ranges, lsmo_point* out);

/* ---- mapping: scene clipper and merger around the aligner ------------------------ */
int lsmo_clip_scene_f(const lsmo_projector* pr, const lsmo_point* scene, int n_scene, const float robot_in_local_map[3],
                      const float sensor_in_robot[3], lsmo_point* out, int* out_src);
int lsmo_clip_scene_d(const lsmo_projector* pr, const lsmo_point* scene, int n_scene, const double robot_in_local_map[3],
                      const double sensor_in_robot[3], lsmo_point* out, int* out_src);
/* voxelize_resolution > 0 branch of the clipper (scene_clipper_projective_2d.cpp:36-48); <= 0: lsmo_clip_scene_f */
int lsmo_clip_scene_voxelized_f(const lsmo_projector* pr, const lsmo_point* scene, int n_scene, const float robot_in_local_map[3],
                                const float sensor_in_robot[3], float voxelize_resolution, lsmo_point* out);
int lsmo_merge_scene_f(const lsmo_projector* pr, lsmo_point* scene, int n_scene, const lsmo_point* meas, int n_meas,
                       const float measurement_in_scene[3], float merge_threshold, int counts[3]);
int lsmo_merge_scene_d(const lsmo_projector* pr, lsmo_point* scene, int n_scene, const lsmo_point* meas, int n_meas,
                       const double measurement_in_scene[3], double merge_threshold, int counts[3]);

/* batch convenience for the CPU baseline: one shared moving cloud (the map), ragged fixed clouds
 * (scans) packed back to back with offsets[n+1]; single slice; n_threads >= 1 (pthreads). */
int lsmo_align_batch_f(const lsmo_aligner_params* ap, const lsmo_slice_params* sp,
                       const lsmo_point* fixed_packed, const int* fixed_offsets, int n_alignments,
                       const lsmo_point* moving, int n_moving,
                       const float* x0 /* [n][3] */, float* x_out /* [n][3] */, float* H_out /* [n][9] */,
                       int* status_out, lsmo_iter_stats* last_stats /* [n] or NULL */, int n_threads);
/* ... the same with every worker's wall time and number of alignments handed back ([n_threads] each, may be NULL): the workers share one atomic work counter */
int lsmo_align_batch_timed_f(const lsmo_aligner_params* ap, const lsmo_slice_params* sp,
                             const lsmo_point* fixed_packed, const int* fixed_offsets, int n_alignments,
                             const lsmo_point* moving, int n_moving, const float* x0, float* x_out, float* H_out,
                             int* status_out, lsmo_iter_stats* last_stats, int n_threads, double* thread_seconds, int* thread_jobs);

/* ---- `_r`: the same routines in the reference's own arithmetic (see the header comment) ------------------------------ */
int lsmo_project_r(const lsmo_projector* pr, const lsmo_point* cloud, int n, const float pose[3],
                   int* out_src, float* out_depth, float* out_xyn);
int lsmo_find_projective_r(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                           const lsmo_point* moving, int n_moving, const float pose[3], lsmo_corr* out);
int lsmo_find_nn_r(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                   const lsmo_point* moving, int n_moving, const float pose[3], lsmo_corr* out);
int lsmo_find_distmap_r(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                        const lsmo_point* moving, int n_moving, const float pose[3], lsmo_corr* out);
int lsmo_linearize_r(const lsmo_slice_params* sp, const lsmo_point* fixed, const lsmo_point* moving,
                     const lsmo_corr* corr, int n_corr, const float pose[3], float H[9], float b[3], lsmo_iter_stats* st);
int lsmo_solve_update_r(const float H[9], const float b[3], float damping, float pose[3], float dx[3]);
int lsmo_align_r(const lsmo_aligner_params* ap, int n_slices, const lsmo_slice_params* sp,
                 const lsmo_point* const* fixed, const int* n_fixed, const lsmo_point* const* moving, const int* n_moving,
                 const float x0[3], float x_out[3], float H_out[9], lsmo_iter_stats* stats, int* iterations_done);
int lsmo_clip_scene_r(const lsmo_projector* pr, const lsmo_point* scene, int n_scene, const float robot_in_local_map[3],
                      const float sensor_in_robot[3], lsmo_point* out, int* out_src);
int lsmo_merge_scene_r(const lsmo_projector* pr, lsmo_point* scene, int n_scene, const lsmo_point* meas, int n_meas,
                       const float measurement_in_scene[3], float merge_threshold, int counts[3]);
void lsmo_compose_r(const float a[3], const float b[3], float out[3]);
void lsmo_inverse_r(const float a[3], float out[3]);
/* the believed upstream KD-tree search (LSMO_FINDER_KDTREE_APPROX), all three arithmetics */
int lsmo_find_kdtree_f(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                       const lsmo_point* moving, int n_moving, const float pose[3], lsmo_corr* out);
int lsmo_find_kdtree_d(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                       const lsmo_point* moving, int n_moving, const double pose[3], lsmo_corr* out);
int lsmo_find_kdtree_r(const lsmo_slice_params* sp, const lsmo_point* fixed, int n_fixed,
                       const lsmo_point* moving, int n_moving, const float pose[3], lsmo_corr* out);

#ifdef __cplusplus
}
#endif
#endif
