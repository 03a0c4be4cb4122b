/*
 * lsm2d.h -- C ABI of the MI355X-native 2D scan-matching core (liblsm2d_hip.so).
 *
 * Drop-in boundary for ONE hot path of rvp-group/srrg2_laser_slam_2d: the per-scan inner loop
 * "correspondence search + plane-to-plane ICP Gauss-Newton step" that the reference runs through
 *   - CorrespondenceFinder_<Isometry2f, PointNormal2fVectorCloud, PointNormal2fVectorCloud>
 *     (srrg2_laser_slam_2d/src/srrg2_laser_slam_2d/registration/correspondence_finder_normal_2f.h:9-13),
 *   - AlignerSliceProcessorLaser2D[WithSensor]  (registration/aligner_slice_processor_laser_2d.h:7-42),
 *   - MultiAligner2D (upstream; driven as in apps/visual_test_aligner_2d.cpp:123-156).
 * Paths below are relative to srrg2_laser_slam_2d/src/srrg2_laser_slam_2d/ unless they start
 * with apps/ or configurations/.
 *
 * Conventions (apps/visual_test_aligner_2d.cpp:126,145):
 *   - a point is the PointNormal2f payload: 4 x fp32 (x, y, nx, ny), array-of-structs on the host side;
 *   - a pose is float[3] = (x, y, theta) = geometry2d::t2v(Isometry2f), metres / radians;
 *   - the estimate maps MOVING into FIXED (setMovingInFixed / setLocalMapInSensor);
 *   - H is row-major 3x3 in the right-perturbation basis X <- X * v2t(dx).
 * Plain pointers and sizes only; no exceptions cross this boundary: every entry point returns an
 * lsm2d_status (0 = ok, < 0 = call-level error); per-alignment outcomes go to out_status[].
 * One lsm2d_context = one device + one HIP stream; a context is not thread-safe, different contexts
 * may be used concurrently.  Host buffers are borrowed for the duration of a call only.
 */
#ifndef LSM2D_H
#define LSM2D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSM2D_VERSION 160 /* 0.1.1: + lsm2d_preprocess_scan_into, asynchronous clip / merge (NULL size outputs), non-blocking upload;
                             0.1.11: + lsm2d_get_option, align_path 3; 0.1.12: + lsm2d_merge_scenes;
                             0.2.0: + lsm2d_clip_scene_voxelized, lsm2d_sweep_* (multi-device loop-closure sweep), in-kernel clock options;
                             0.2.1: + lsm2d_cloudset_cloud_sizes, pinned / device-resident ranges in lsm2d_preprocess_scans, options
                                    "distmap_build", "grid_big_threshold", "find_path", "zero_copy_max";
                             0.3.0: + LSM2D_FINDER_KDTREE (the reference KD-tree's own build and single-leaf descent, honouring max_leaf_range /
                                    min_leaf_points: lsm2d_slice_params grew two fields), lsm2d_aligner_params.termination_chi_epsilon,
                                    sweep option "peer_copy";
                             0.4.0: lsm2d_iteration_stats grew the order-independent digest of the iteration's correspondence set (pair_digest_lo / _hi);
                                    lsm2d_aligner_params grew enable_inlier_only_runs / keep_only_inlier_correspondences (no longer refused);
                                    + lsm2d_align_batch_pairs (the correspondences the aligner leaves in its slices), lsm2d_stats_capacity,
                                    lsm2d_pair_hash, lsm2d_estimate_work (work-aware sharding of a candidate sweep);
                             0.5.0: + lsm2d_align_batch_begin / _wait (a batch in flight while the host prepares the next one), lsm2d_preprocess_scans_refill
                                    (fresh scans into an existing set: no allocation, nothing waits); the option keys below are the WHOLE public set (the A/B
                                    knobs of rounds 1-4 exist only in a -DLSM2D_EXPERIMENTS build); read-only keys "uploads", "last_cull_estimate", "experiments";
                             0.6.0: + option "sum_order" (1: H, b and the chi^2 statistics are added pair after pair in the reference's order -- the ONE option
                                    results depend on: with it the aligner is bitwise the sequential fp32 restatement of nicp_post.m:69-90) */

/* ---- status codes -------------------------------------------------------------------------
 * Replace: std::runtime_error throws of the finders (registration/correspondence_finder_projective_2d.cpp:21-31,
 * registration/correspondence_finder_nn_2d.cpp:11-18) and the upstream aligner status enum. */
typedef enum {
  LSM2D_SUCCESS                    = 0,
  LSM2D_NOT_ENOUGH_CORRESPONDENCES = 1,  /* per-alignment */
  LSM2D_NOT_ENOUGH_INLIERS         = 2,  /* per-alignment */
  LSM2D_SINGULAR_H                 = 3,  /* per-alignment */
  LSM2D_BAD_ARGUMENT               = -1,
  LSM2D_DEVICE_ERROR               = -2, /* a HIP call failed; lsm2d_last_error() has the text */
  LSM2D_OUT_OF_MEMORY              = -3,
  LSM2D_CAPACITY_EXCEEDED          = -4, /* output buffer too small / canvas too large for LDS */
  LSM2D_NO_DEVICE                  = -5
} lsm2d_status;

/* ---- parameter blocks ----------------------------------------------------------------------
 * PointNormal2fProjectorPolar parameters as the reference sets them
 * (apps/synthetic_scene_generator.cpp:69-75; configurations/stage_segway_double_config_MULTI.json:71-97).
 * Camera matrix convention [[cols/(angle_max-angle_min), cols/2],[0,1]]
 * (apps/synthetic_scene_generator.cpp:59-66, sensor_processing/raw_data_preprocessor_projective_2d.cpp:83-90). */
typedef struct {
  int32_t canvas_cols;
  float   angle_min, angle_max;  /* angle_col_min / angle_col_max [rad] */
  float   range_min, range_max;  /* [m] */
  float   col_offset;            /* column = floor(K00*atan2(y,x) + K01 + col_offset); 0 = truncate, 0.5 = nearest */
} lsm2d_projector;

typedef enum {
  LSM2D_FINDER_PROJECTIVE = 0, /* CorrespondenceFinderProjective2f  (registration/correspondence_finder_projective_2d.cpp:18-77) */
  LSM2D_FINDER_NN         = 1, /* CorrespondenceFinderKDTree2D      (registration/correspondence_finder_kd_tree_2d.cpp:5-38) with an EXACT
                                  nearest-neighbour search (uniform grid): a superset-quality substitute, max_leaf_range / min_leaf_points unused */
  LSM2D_FINDER_DISTMAP    = 2, /* CorrespondenceFinderNN2D          (registration/correspondence_finder_nn_2d.cpp:54-97) */
  LSM2D_FINDER_KDTREE     = 3  /* CorrespondenceFinderKDTree2D as the reference runs it: reset() builds KDTree2D(coordinates, max_leaf_range,
                                  min_leaf_points) (.cpp:31-38, .h:36-39), compute() calls findNeighbor per moved point (.cpp:12-27).  Upstream's
                                  tree (srrg2_core, not in the reference tree) is restated as SURVEY.md App. A.4 believes it: split through the
                                  mean along the principal eigenvector until a node holds fewer than min_leaf_points points or is smaller than
                                  max_leaf_range; descent to the single leaf on the query's side, no backtracking -- hence APPROXIMATE: it may
                                  return a farther neighbour than LSM2D_FINDER_NN, or none */
} lsm2d_finder;

typedef enum { LSM2D_ROBUST_NONE = 0, LSM2D_ROBUST_CAUCHY = 1 } lsm2d_robustifier;

/* One aligner slice = finder + factor + robustifier, i.e. one AlignerSliceProcessorLaser2D[WithSensor]
 * (registration/aligner_slice_processor_laser_2d.h:7-42) with the config fields of MULTI.json:160-188. */
typedef struct {
  int32_t         finder;                  /* lsm2d_finder */
  lsm2d_projector projector;               /* projective finder: param_projector (.h:22-26) */
  float           point_distance;          /* projective: param_point_distance, default 0.5 (.h:16-20) */
  float           normal_cos;              /* all finders: param_normal_cos, default 0.8 */
  float           max_distance;            /* NN / DISTMAP: param_max_distance_m (kd .h:23, nn .h:20-24) */
  float           resolution;              /* DISTMAP: param_resolution [m/pixel] (nn .h:25-29) */
  int32_t         robustifier;             /* lsm2d_robustifier (RobustifierCauchy, MULTI.json:153-158) */
  float           chi_threshold;           /* Cauchy tau */
  int32_t         min_num_correspondences; /* slice skipped when #pairs <= this (MULTI.json:179,391,591) */
  float           sensor_in_robot[3];      /* WithSensor variant (registration/aligner_slice_processor_laser_2d_impl.cpp:7-10); zeros = plain */
  float           kd_max_leaf_range;       /* KDTREE: param_max_leaf_range  (kd .h:26-28); <= 0: the class default 1e-2 */
  int32_t         kd_min_leaf_points;      /* KDTREE: param_min_leaf_points (kd .h:29-33); <= 0: the class default 20 */
} lsm2d_slice_params;

/* MultiAligner2D parameters (MULTI.json:700-732, :602-630) + GN damping (MULTI.json:254-259). */
typedef struct {
  int32_t max_iterations;
  int32_t min_num_inliers;
  float   damping;
  /* device-side counterpart of the aligner's "termination_criteria" (MULTI.json:627-630,729-731: unset in both shipped aligners = always
   * max_iterations; the solver's SimpleTerminationCriteria, MULTI.json:218-223, documents epsilon as the "ratio of decay of chi2 between
   * iteration"): 0 = off; > 0: the loop stops after an iteration whose total chi^2 (inliers + kernelised outliers) differs from the
   * previous iteration's by less than epsilon times itself.  The iteration that triggers the stop is still solved and applied. */
  float   termination_chi_epsilon;
  /* MultiAligner2D's two remaining options (MULTI.json:606-610,704-708; both 0 in the shipped configurations).  The upstream class is not in
   * the reference tree, so their semantics are RESTATED from the parameters' own doc strings (PARITY.md section 2, [UPSTREAM-MEMORY]):
   * enable_inlier_only_runs ("toggles additional inlier only runs if sufficient inliers are available"): when the regular loop ended without
   *   a failure and its last iteration counted n_inliers >= min_num_inliers, a second loop of up to max_iterations iterations follows in which
   *   a pair whose factor is not an inlier under its slice's robustifier (chi^2 >= chi_threshold) contributes nothing and an inlier contributes
   *   with weight 1 (no kernel); finders, gates, statistics, the termination criterion (afresh) and the status rules are those of the regular
   *   loop.  An alignment then runs up to 2 * max_iterations iterations: see lsm2d_stats_capacity.
   * keep_only_inlier_correspondences ("toggles removal of correspondences which factors are not inliers in the last iteration"): the
   *   correspondences the aligner leaves in its slices (lsm2d_align_batch_pairs) hold only the pairs whose factor was an inlier in the last
   *   iteration; poses, information matrices and statistics do not depend on it. */
  int32_t enable_inlier_only_runs;
  int32_t keep_only_inlier_correspondences;
} lsm2d_aligner_params;
/* iterations one alignment may run under `aligner`, i.e. the row length of out_stats: max_iterations, doubled with enable_inlier_only_runs (>= 1) */
int32_t lsm2d_stats_capacity(const lsm2d_aligner_params* aligner);

/* Optional odometry-prior cue (AlignerSliceOdom2DPrior, MULTI.json:402-422): e = t2v(Z^-1 X), information omega. */
typedef struct {
  float z[3];
  float omega[9];
} lsm2d_prior;

/* Correspondence(fixed_idx, moving_idx) (registration/correspondence_finder_kd_tree_2d.cpp:25) */
typedef struct { int32_t fixed_idx, moving_idx; } lsm2d_correspondence;

/* per-iteration statistics = what aligner->iterationStats() prints (apps/visual_test_aligner_2d.cpp:156) */
typedef struct {
  int32_t n_correspondences, n_inliers, n_outliers;
  float   chi_inliers, chi_outliers;
  /* order-independent digest of the iteration's correspondence SET -- the pairs the reference aligner exposes per slice
   * (apps/visual_test_aligner_2d.cpp:129-143): the wrapping 64-bit sum of lsm2d_pair_hash(slice, fixed_idx, moving_idx) over every pair counted
   * in n_correspondences (all slices, skipped ones included).  Equal digests <=> the same pairs went into the iteration (up to a 2^-64 hash
   * collision); tests use it to tell "same pairs, different summation order" from "different pairs".  Computed only when statistics are asked for. */
  uint32_t pair_digest_lo, pair_digest_hi;
} lsm2d_iteration_stats;
/* the per-pair hash behind pair_digest -- plain 32-bit integer arithmetic, the same on the host, in the kernels and in the CPU oracle:
 *   a = fixed_idx * 0x9E3779B1;  b = (moving_idx ^ (slice * 0x632BE5AB)) * 0x85EBCA77;
 *   lo = a ^ rotl(b, 13);  hi = b ^ rotl(a, 19);  lo += rotl(lo, 17) ^ b;  hi += rotl(hi, 11) ^ a;  hash = hi << 32 | lo
 * (so a caller holding a correspondence vector can form the digest it should see) */
uint64_t lsm2d_pair_hash(uint32_t slice, uint32_t fixed_idx, uint32_t moving_idx);

typedef struct lsm2d_context  lsm2d_context;
typedef struct lsm2d_cloudset lsm2d_cloudset;

/* ---- library / context ------------------------------------------------------------------------ */
int         lsm2d_version(void);
const char* lsm2d_status_string(int status);
/* text of the last HIP failure seen by this context (or by the library when ctx == NULL) */
const char* lsm2d_last_error(const lsm2d_context* ctx);
/* hip_stream: an existing hipStream_t to launch on (e.g. the caller's torch stream), or NULL to own one */
int  lsm2d_create(int device_id, void* hip_stream, lsm2d_context** out_ctx);
/* waits for the stream, frees the context.  Cloud sets still alive on it stay the caller's to destroy (any time), but no call takes them any more. */
/* (batches begun with lsm2d_align_batch_begin must have been waited for: the call brings every stream of the context to rest before it frees anything, but a
 * lsm2d_pending that is still outstanding refers to the context and must not be used afterwards) */
void lsm2d_destroy(lsm2d_context* ctx);
/* blocks until everything queued on the context's stream has finished */
int  lsm2d_synchronize(lsm2d_context* ctx);
/* Options: the WHOLE public set (15 keys; anything else is LSM2D_BAD_ARGUMENT "unknown option").  Results never depend on any of them but "sum_order".
 * "sum_order": 0 (default) = H, b and the chi^2 statistics of an iteration are added in TREES (a thread's pairs, then the 64 lanes of a wave, then the eight
 *   waves): the fast order.  1 = added PAIR AFTER PAIR in the order of the reference's correspondence vector -- ascending canvas column for the projective
 *   finder (registration/correspondence_finder_projective_2d.cpp:55-74), ascending moving index for the point-query finders
 *   (registration/correspondence_finder_kd_tree_2d.cpp:12-27) -- one factor after the other into H and b as octave/solver/nicp_post.m:69-90 does, slice
 *   totals added in slice order.  Both orders use the same per-pair terms and the same fused operations; they differ in the association of fp32 sums,
 *   i.e. in the last bits of H and b, which a pair sitting on a gate can turn into another correspondence set a few iterations later (PARITY.md section 0).
 *   With 1 the aligner (lsm2d_align_batch and its begin / wait / pairs forms, every finder kind, priors, sensor offsets, the split path) and
 *   lsm2d_linearize equal the sequential fp32 oracle (oracle/: lsmo_align_f, lsmo_linearize_f) BIT FOR BIT.  Cost: the pairs' terms go through LDS (14 KB
 *   more per workgroup) and one wave adds them one after the other (a quad of lanes per quantity): configs[1] (1000 scans vs a 100k-point map) takes
 *   about 1.2 x the default order's step (1.03 M against 1.24 M alignments/s: DESIGN.md section 5; the point-query finders in the tracker's wiring, whose 100 000
 *   queries per iteration all pass a barrier per 512, 3 - 5 x).  Calls the latency kernel would take (align_path 3) run on k_align instead: "last_align_path" reads 1.
 * "align_width": the launch form of a culled projective batch (k_align): 0 = automatic (default: workgroups of 512 threads; of 256 -- six alignments per CU round instead
 *   of four -- for batches just above a multiple of 1024 alignments, where the last few would otherwise run a round of their own on an empty chip; PACKED -- one round
 *   of 1024 workgroups, the lightest alignments two to a workgroup, one after the other -- for 1025 .. 1048 and 1537 .. 2047 alignments, and for up to 32 more than 2048 or 3072; with "sum_order" 1, which has no narrow form: 1025 .. 1600), 512 / 256 = always that
 *   width, 1024 = packed whenever the batch has more than 1024 and fewer than 4096 alignments and is not a multiple of 1024.  The narrow workgroups keep the wide kernel's 512 virtual threads in the bin walk and the
 *   sums, a packed workgroup runs the same kernel body twice: bit-identical results (get: "last_align_width": 512, 256, or 1024 for a packed launch).
 * "align_path": 0 = automatic (default), 1 = always one workgroup per alignment (k_align), 2 = always the split path (k_split_project +
 *   k_split_finish per iteration; projective slices only), 3 = the latency kernel whenever the batch has one or two projective slices
 *   (k_align_pair: 512 threads per slice, two slices' passes side by side in one workgroup; automatic for <= 256 alignments).  All paths return
 *   bit-identical results; the split path is for a handful of alignments against a large cloud, the latency kernel for calls that cannot fill
 *   the chip (the live tracker).
 * "kernel_timing": 1 records HIP events around the hot-path launches so that lsm2d_last_kernel_ms can report them; 0 (default) does not -- the
 *   two timed events per operation cost a latency-critical caller such as the live tracker ~20 % of its step -- and lsm2d_last_kernel_ms
 *   returns LSM2D_BAD_ARGUMENT.  "clock_stride" (default 0 = ~32 per launch): with kernel timing on, every clock_stride-th workgroup of a k_align
 *   launch stamps its shader-cycle and 100 MHz counters (get: "last_kernel_clock_khz", "last_workgroup_lifetime_ns").
 * "find_path": 0 = automatic (default: a point-query lsm2d_find_correspondences call with more queries than one workgroup takes in a trip runs on
 *   many workgroups, two launches; the projective finder z-buffers a cloud of more than 32768 points over many workgroups first), 1 = always one
 *   workgroup; same pairs, same order.
 * "zero_copy_max" (default 256): batches of at most this many alignments read their arguments from, and write their results to, pinned host memory
 *   directly (no transfers, status words polled) -- above 256 alignments only when the batch carries no index arrays; measured slower than the
 *   transfers at 1000 alignments.  0 switches the zero-copy form off.
 * "cull" (default 1): the aligner's projective slices drop, exactly, the parts of a map-sized moving cloud that cannot yield a pair (and its
 *   point-query slices the tiles of queries nobody is near); 0 streams everything.  "cull_margin_um" (10000) / "cull_margin_urad" (2000, at most
 *   50000): how far a pose may move before the kept survivor lists of the projective culling are rebuilt.
 * "balance" (default 1): batches of 257 .. 4096 culled alignments are placed on the chip by estimated work (one small launch ahead of k_align; a
 *   batch run again with unchanged sets and start poses keeps its placement: get "last_cull_estimate"); 0: workgroup i runs alignment i.
 * "grid_big_threshold" (default 16384): fixed clouds of at least this many points get the NN finder's search grid built by chip-wide kernels
 *   (histogram / scan / scatter over many workgroups) instead of one workgroup per cloud.
 * "distmap_build": 0 = automatic (default: the distance maps of CorrespondenceFinderNN2D are built from the points' side, one disc of atomic minima
 *   per point, whenever squared pixel distance and point index pack into 31 bits), 1 = always the per-pixel gather build; identical maps.
 * "kd_lds_nodes" (default 1536): nodes of the fixed cloud's KD-tree a k_align workgroup keeps in LDS (its top levels).
 * Read-only (lsm2d_get_option): "last_align_path" (1 k_align, 2 split, 3 slice pair), "last_query_cull", "last_cull_estimate", "last_kd_levels",
 *   "last_kd_nodes", "last_kernel_clock_khz", "last_workgroup_lifetime_ns", "max_dyn_lds", "uploads" (host-to-device cloud uploads queued so far:
 *   lsm2d_cloudset_create / _upload / lsm2d_preprocess_scans_refill), "last_h2d_bytes", "experiments" (1: this library was built with
 *   -DLSM2D_EXPERIMENTS and also knows the A/B knobs of DESIGN.md App. A; the shipped library: 0). */
int  lsm2d_set_option(lsm2d_context* ctx, const char* key, int64_t value);
/* reads a knob back; also "last_align_path": what the most recent lsm2d_align_batch ran (1 k_align, 2 split, 3 slice pair) */
int  lsm2d_get_option(lsm2d_context* ctx, const char* key, int64_t* out_value);
/* device time [ms] of the hot-path kernel launches of the most recent call (HIP events on the context's stream);
 * needs lsm2d_set_option(ctx, "kernel_timing", 1) */
int  lsm2d_last_kernel_ms(lsm2d_context* ctx, float* out_ms);

/* ---- clouds: device-resident ragged sets of PointNormal2fVectorCloud ---------------------------
 * Replace: the raw non-owning PointNormal2fVectorCloud* the reference hands to setFixed / setMoving
 * (apps/visual_test_correspondence_finder_projective_2d.cpp:74-75).  A set holds n_clouds clouds
 * packed back to back; offsets[n_clouds+1] delimit them (NULL when n_clouds == 1).  The local map
 * shared by a batch is a set with ONE cloud. */
int  lsm2d_cloudset_create(lsm2d_context* ctx, const float* points_xynn, const int32_t* offsets,
                           int32_t n_clouds, int64_t total_points, lsm2d_cloudset** out_set);
/* same, from points already in device memory on ctx's device (float4 per point); offsets stay host.
 * ORDERING: the points are read on the CONTEXT's stream.  Whatever produced them (another stream, e.g. torch's current stream) must have
 * finished -- synchronise that stream or make it wait-for -- or the context must have been created on that very stream
 * (lsm2d_create's hip_stream); the library cannot know the producer.  The same holds for device-resident `ranges` below. */
int  lsm2d_cloudset_create_from_device(lsm2d_context* ctx, const void* d_points_xynn, const int32_t* offsets,
                                       int32_t n_clouds, int64_t total_points, lsm2d_cloudset** out_set);
/* a single growable cloud (count 0) with room for capacity_points: the device-resident local map / clipped scene */
int  lsm2d_cloudset_create_reserved(lsm2d_context* ctx, int64_t capacity_points, lsm2d_cloudset** out_set);
/* refill an existing SINGLE-cloud set in place (no allocation); LSM2D_CAPACITY_EXCEEDED when it does not fit.  The points are
 * copied into the set's pinned staging buffer before the call returns and travel asynchronously on the context's stream:
 * for scan-sized sets (<= 16 384 points) whoever reads the set first queues their unpacking, and a single-alignment projective
 * lsm2d_align_batch unpacks its fixed sets inside the aligner kernel -- the live tracker's scans cost no launch of their own. */
int  lsm2d_cloudset_upload(lsm2d_cloudset* set, const float* points_xynn, int64_t n_points);
/* copy cloud `cloud_index` back to the host as (x, y, nx, ny) rows; *out_n = its size */
int  lsm2d_cloudset_download(const lsm2d_cloudset* set, int32_t cloud_index, float* out_points_xynn, int64_t capacity, int64_t* out_n);
void lsm2d_cloudset_destroy(lsm2d_cloudset* set);
int32_t lsm2d_cloudset_num_clouds(const lsm2d_cloudset* set);
int64_t lsm2d_cloudset_num_points(const lsm2d_cloudset* set);
/* points in cloud `cloud_index` (-1 when out of range).  After an asynchronous lsm2d_clip_scene / lsm2d_merge_scene the size
 * of the set they wrote is known to the device only; the size queries (and lsm2d_cloudset_download) then wait for the stream. */
int64_t lsm2d_cloudset_cloud_size(const lsm2d_cloudset* set, int32_t cloud_index);
/* all sizes at once (a preprocessed batch holds thousands of clouds): fills out_sizes[0 .. min(capacity, num_clouds)), returns the number written or a negative status */
int32_t lsm2d_cloudset_cloud_sizes(const lsm2d_cloudset* set, int32_t* out_sizes, int32_t capacity);

/* ---- a3: PointNormal2fProjectorPolar::compute -------------------------------------------------
 * One polar z-buffer pass of cloud `cloud_index` seen through `pose` (points are mapped by pose,
 * i.e. pose = camera_pose^-1).  Outputs (host, each canvas_cols long; any may be NULL):
 * source index or -1, depth, transformed point (x, y, nx, ny). */
int lsm2d_project(lsm2d_context* ctx, const lsm2d_projector* projector, const lsm2d_cloudset* cloud,
                  int32_t cloud_index, const float pose[3], int32_t* out_source_idx, float* out_depth,
                  float* out_transformed_xynn);

/* ---- RawDataPreprocessorProjective2D::compute, batched (sensor_processing/raw_data_preprocessor_projective_2d.cpp:13-51,
 * :77-104): ranges of n_scans LaserMessages -> one cloud set of n_scans PointNormal2fVectorClouds that never leaves the
 * device (ready to be the aligner's `fixed`).  Per scan: polar unprojection with the reference's sensor matrix
 * [[n/(angle_max-angle_min), n/2]], sliding-window normals (NormalComputator1DSlidingWindow: normal_point_distance,
 * normal_min_points; MULTI.json:845-853), voxelisation at voxelize_resolution (.h:41-45; <= 0 keeps every valid point).
 * n_beams <= 2048.  The upstream pieces are not in the reference tree: semantics as restated in oracle/lsm2d_oracle.c F2.1-F2.3. */
typedef struct {
  int32_t n_beams;
  float   angle_min, angle_max;    /* LaserMessage angle_min / angle_max [rad] */
  float   range_min, range_max;    /* max(message, param) / min(message, param), .cpp:83-84 */
  float   normal_point_distance;
  int32_t normal_min_points;
  float   voxelize_resolution;
} lsm2d_preprocessor;
/* `ranges` [n_scans][n_beams] may live in pageable host memory (staged through the context's pinned buffer), in pinned / registered host
 * memory (copied from directly) or in device memory of the context's device (read in place: nothing crosses the host link; see the
 * ORDERING note at lsm2d_cloudset_create_from_device: the producer's stream must have been synchronised with). */
int lsm2d_preprocess_scans(lsm2d_context* ctx, const lsm2d_preprocessor* params, const float* ranges,
                           int32_t n_scans, lsm2d_cloudset** out_set);
/* The same operation INTO a set that lsm2d_preprocess_scans made for the same number of scans and beams: no allocation, nothing waits (RawDataPreprocessorProjective2D::compute
 * per incoming message, sensor_processing/raw_data_preprocessor_projective_2d.cpp:13-51, for a BATCH of fresh messages per step).  Pinned `ranges` are fetched by an
 * asynchronous copy, pageable ones staged first, device-resident ones read in place; the clouds' sizes stay on the device (lsm2d_cloudset_cloud_size asks for them: a wait).
 * Same kernel, same bits as lsm2d_preprocess_scans.  The caller keeps `ranges` untouched until the batch that reads the set has been waited for.
 * While a batch is in flight the refill runs on a stream of its own, ordered before the next aligner call on this context (and before lsm2d_cloudset_cloud_size /
 * lsm2d_synchronize); other entry points that read a set refilled while a batch was in flight: lsm2d_synchronize first. */
int lsm2d_preprocess_scans_refill(lsm2d_context* ctx, const lsm2d_preprocessor* params, const float* ranges, int32_t n_scans, lsm2d_cloudset* set);
/* The live tracker's form of the same operation: ONE scan into an existing reserved single-cloud set (capacity >= n_beams) --
 * no allocation, nothing waits: the ranges are staged in the set's pinned buffer, the cloud's size stays on the device until
 * somebody asks (see lsm2d_clip_scene).  Same kernel, same bits as lsm2d_preprocess_scans with n_scans = 1.  Unless kernel timing is
 * on, the launch itself is queued by the set's first reader, and an lsm2d_align_batch that reads several such sets (front and rear
 * scanner) queues them as one launch, one workgroup per scan. */
int lsm2d_preprocess_scan_into(lsm2d_context* ctx, const lsm2d_preprocessor* params, const float* ranges /* host [n_beams] */,
                               lsm2d_cloudset* out_reserved_set);

/* ---- SceneClipperProjective2D::compute (mapping/scene_clipper_projective_2d.cpp:11-65, voxelize_resolution = 0
 * as in both shipped configs, MULTI.json:673-683): what the sensor at robot_in_local_map * sensor_in_robot sees of
 * `full_scene` -- at most one point per projector column, ascending column, expressed in the ROBOT frame.
 * `clipped` is a reserved single-cloud set (capacity >= canvas_cols) that is overwritten and stays on the device,
 * ready to be the aligner's moving cloud.  out_source_idx (host, canvas_cols entries) may be NULL.
 * out_n_points == NULL (then out_source_idx must be NULL too): ASYNCHRONOUS -- the call only queues the work on the context's
 * stream and nothing is copied back; lsm2d_align_batch (projective slices), lsm2d_merge_scene and another lsm2d_clip_scene accept
 * the sets as they are, so a tracker step (clip -> upload -> align -> merge) synchronises once, for the aligner's pose. */
int lsm2d_clip_scene(lsm2d_context* ctx, const lsm2d_projector* projector, const lsm2d_cloudset* full_scene,
                     int32_t scene_index, const float robot_in_local_map[3], const float sensor_in_robot[3],
                     lsm2d_cloudset* clipped, int32_t* out_n_points, int32_t* out_source_idx);

/* The same with the clipper's voxelize_resolution > 0 branch (mapping/scene_clipper_projective_2d.cpp:36-48): the clipped points, still
 * in the sensor frame, are voxelised with coefficients (res, res, 0.1, 0.1) -- equal keys averaged, ascending key order (assumption
 * F2.3, oracle/lsm2d_oracle.c) -- and then moved to the robot frame.  out_source_idx must be NULL (a voxel has no source point);
 * canvas_cols <= 2048; voxelize_resolution <= 0 is lsm2d_clip_scene. */
int lsm2d_clip_scene_voxelized(lsm2d_context* ctx, const lsm2d_projector* projector, const lsm2d_cloudset* full_scene,
                               int32_t scene_index, const float robot_in_local_map[3], const float sensor_in_robot[3],
                               float voxelize_resolution, lsm2d_cloudset* clipped, int32_t* out_n_points, int32_t* out_source_idx);

/* ---- MergerProjective2D::compute (mapping/merger_projective_2d.cpp:9-100): folds `measurement` (cloud
 * measurement_index, in its own sensor/robot frame) into the single-cloud reserved set `scene`, in place: per projector
 * column seen from measurement_in_scene a measurement point is merged into (|depth difference| < merge_threshold),
 * replaces (it lies behind) or is appended next to the scene's nearest point; measurement depths beyond
 * 0.9*range_max are ignored.  out_counts[3] = {new, merged, replaced} (may be NULL).
 * out_scene_size == NULL (then out_counts must be NULL too): ASYNCHRONOUS, see lsm2d_clip_scene. */
int lsm2d_merge_scene(lsm2d_context* ctx, const lsm2d_projector* projector, lsm2d_cloudset* scene,
                      const lsm2d_cloudset* measurement, int32_t measurement_index, const float measurement_in_scene[3],
                      float merge_threshold, int32_t* out_scene_size, int32_t* out_counts);
/* Several measurements merged into the scene one after the other, exactly as n calls of lsm2d_merge_scene in this order would (the live
 * tracker merges the front and the rear scan at the corrected pose) -- but as ONE launch when the scene and the measurements are small
 * (<= 4 measurements, scene + (n-1) canvas_cols <= 32 768 points, kernel timing off).  meas_index NULL: cloud 0 of every set;
 * measurement_in_scene: [n][3]; out_size NULL: asynchronous (see lsm2d_clip_scene); out_counts: [n][3] (new, merged, replaced) or NULL. */
int lsm2d_merge_scenes(lsm2d_context* ctx, const lsm2d_projector* projector, lsm2d_cloudset* scene, int32_t n_measurements,
                       const lsm2d_cloudset* const* measurements, const int32_t* meas_index, const float* measurement_in_scene,
                       float merge_threshold, int32_t* out_size, int32_t* out_counts);

/* ---- plugin interface #1: CorrespondenceFinder_::compute ---------------------------------------
 * Replaces compute() of the three finders (registration/correspondence_finder_projective_2d.cpp:18-77,
 * registration/correspondence_finder_kd_tree_2d.cpp:5-38, registration/correspondence_finder_nn_2d.cpp:54-97)
 * after setFixed / setMoving / setLocalMapInSensor / setCorrespondences.  Pairs come out in the
 * reference's order (ascending column for PROJECTIVE, ascending moving index otherwise). */
int lsm2d_find_correspondences(lsm2d_context* ctx, const lsm2d_slice_params* slice,
                               const lsm2d_cloudset* fixed, int32_t fixed_index,
                               const lsm2d_cloudset* moving, int32_t moving_index,
                               const float local_map_in_sensor[3],
                               lsm2d_correspondence* out_pairs, int32_t capacity, int32_t* out_n_pairs);

/* ---- factor: SE2Plane2PlaneErrorFactor over a correspondence vector ----------------------------
 * Replaces the per-correspondence errorAndJacobian + robustifier + H/b accumulation of the
 * correspondence-driven factor bound at registration/aligner_slice_processor_laser_2d.h:4,8
 * (math: octave/solver/nicp_post.m:4-26,69-90). */
int lsm2d_linearize(lsm2d_context* ctx, const lsm2d_slice_params* slice,
                    const lsm2d_cloudset* fixed, int32_t fixed_index,
                    const lsm2d_cloudset* moving, int32_t moving_index,
                    const lsm2d_correspondence* pairs, int32_t n_pairs, const float pose[3],
                    float out_H[9], float out_b[3], lsm2d_iteration_stats* out_stats);

/* ---- plugin interface #2: MultiAligner2D::compute, batched --------------------------------------
 * Replaces aligner->setFixed / setMoving / setMovingInFixed / compute / movingInFixed /
 * iterationStats (apps/visual_test_aligner_2d.cpp:123-156) for n_alignments independent alignments
 * (the tracker uses 1; MultiLoopDetectorBruteForce2D's candidate loop, MULTI.json:964-986, uses many).
 * Each alignment runs max_iterations x { every slice: finder + factor ; one 3x3 solve ; right update }
 * entirely on the device.  Cloud selection for alignment i, slice s: fixed[s] cloud
 * fixed_index[s*n+i] (or i when fixed_index == NULL, or 0 when the set holds one cloud); same for moving. */
typedef struct {
  int32_t                       n_alignments;
  int32_t                       n_slices;
  const lsm2d_slice_params*     slices;        /* [n_slices] */
  const lsm2d_cloudset* const*  fixed;         /* [n_slices] */
  const lsm2d_cloudset* const*  moving;        /* [n_slices] */
  const int32_t*                fixed_index;   /* [n_slices][n_alignments] or NULL */
  const int32_t*                moving_index;  /* [n_slices][n_alignments] or NULL */
  const float*                  init_pose;     /* [n_alignments][3]  setMovingInFixed */
  const lsm2d_prior*            prior;         /* [n_alignments] or NULL */
} lsm2d_batch;

int lsm2d_align_batch(lsm2d_context* ctx, const lsm2d_aligner_params* aligner, const lsm2d_batch* batch,
                      float* out_pose,                  /* [n][3]  movingInFixed() */
                      float* out_H,                     /* [n][9]  information matrix = H of the last iteration; may be NULL */
                      int32_t* out_status,              /* [n]     lsm2d_status >= 0 */
                      int32_t* out_iterations,          /* [n]     iterations started; may be NULL */
                      lsm2d_iteration_stats* out_stats  /* [n][lsm2d_stats_capacity(aligner)]; may be NULL */);

/* What an alignment of `batch` will cost relative to the others, WITHOUT running it: for projective slices against a map-sized moving cloud the
 * number of chunks of that cloud (of 512) that survive the exact culling against the alignment's fixed canvas at its start pose -- what its first
 * iteration streams (k_cull_estimate: the quantity lsm2d_align_batch itself places a batch on the chip by); 1 for every alignment when there is
 * no such slice.  A caller that shards a candidate sweep over several devices or processes (MultiLoopDetectorBruteForce2D's loop, MULTI.json:964-986)
 * balances the shards by the SUM of these numbers instead of by candidate count: with the culling an alignment's time follows it (33-59 % of the
 * chunks survive on configs[1]).  out_work: [n_alignments]. */
int lsm2d_estimate_work(lsm2d_context* ctx, const lsm2d_batch* batch, int32_t* out_work);

/* lsm2d_align_batch in two halves: what a host that feeds the aligner batch after batch does between launching one and needing its poses
 * (the reference's candidate loop, MULTI.json:964-986, and its per-scan tracker, apps/visual_test_aligner_2d.cpp:123-156, are synchronous: this is
 * what a pipelined host puts in their place).  begin() queues everything -- start poses, placement, kernels, the copies of the results -- and returns;
 * the batch descriptor and what it points to may be reused as soon as it has.  wait() blocks until THAT batch is done (a younger one may be queued
 * behind it) and fills the outputs exactly as lsm2d_align_batch would have: begin + wait == lsm2d_align_batch, bit for bit.  While a batch is in flight,
 * the NEXT batch's lsm2d_preprocess_scans_refill and the pre-kernels of its begin() run on side streams, in the slots the launch in flight leaves
 * free; and each asynchronously begun batch launches on a stream of its LANE's own, so the younger of two batches in flight starts on the slots the older one's
 * tail leaves free instead of waiting for its last workgroup (configs[1], the same resident batch begun again and again: 0.69 ms per batch against 0.76 one at a time).  At most two batches are in flight per context; they are waited for in the order they were begun; every begun batch must be waited for.
 * While TWO are in flight the context's staging buffers are theirs: lsm2d_preprocess_scans_refill, lsm2d_align_batch_wait, the option calls and
 * lsm2d_synchronize work, every other call that moves data returns LSM2D_BAD_ARGUMENT (with one in flight everything works).
 * The cloud sets a batch in flight reads must not be modified: a pipeline alternates between two scan sets -- or, better, between THREE, refilling a step
 * ahead: per step  begin(i) ; refill(set of step i + 1) ; wait(i - 1).  The launch in flight holds every wave slot of the chip, so a preprocessing launch
 * queued beside it finishes only after it; queued a step ahead it is done before begin(i + 1) queues the estimate that reads its clouds, and batch i + 1
 * starts where batch i ends (tests/cpp/stream_step_bench.cpp, bench.py --stream).
 * want_stats != 0: the batch keeps per-iteration statistics (wait's out_stats may then be non-NULL). */
typedef struct lsm2d_pending lsm2d_pending;
int lsm2d_align_batch_begin(lsm2d_context* ctx, const lsm2d_aligner_params* aligner, const lsm2d_batch* batch, int32_t want_stats, lsm2d_pending** out_pending);
int lsm2d_align_batch_wait(lsm2d_pending* pending, float* out_pose, float* out_H, int32_t* out_status, int32_t* out_iterations, lsm2d_iteration_stats* out_stats);

/* The same call, additionally handing back what aligner->compute() leaves in every slice's correspondence vector (the reference's
 * slice->correspondences(), apps/visual_test_aligner_2d.cpp:129-143): the pairs of the LAST iteration each alignment started, in the finder's
 * order (ascending column / ascending moving index) -- with keep_only_inlier_correspondences only those whose factor was an inlier in that
 * iteration.  out_pairs: [n][n_slices][pair_capacity]; out_n_pairs: [n][n_slices].  pair_capacity must hold a slice's largest possible vector
 * (canvas_cols of a projective slice, the largest moving cloud of a point-query slice) or the call returns LSM2D_CAPACITY_EXCEEDED.  The
 * pairs are re-derived after the aligner kernel from the pose its last iteration started at (one finder pass per alignment and slice, same
 * arithmetic, hence the same pairs: tests compare their digest with the in-kernel one); out_pairs == NULL is lsm2d_align_batch. */
int lsm2d_align_batch_pairs(lsm2d_context* ctx, const lsm2d_aligner_params* aligner, const lsm2d_batch* batch,
                            float* out_pose, float* out_H, int32_t* out_status, int32_t* out_iterations, lsm2d_iteration_stats* out_stats,
                            lsm2d_correspondence* out_pairs, int32_t pair_capacity, int32_t* out_n_pairs);

/* ---- the candidate loop of MultiLoopDetectorBruteForce2D / MultiRelocalizer2D (MULTI.json:964-986, :749-769) over the GPUs of one
 * node, in ONE process and without Python: a context per device, the submap uploaded to device_ids[0] once and replicated device to
 * device (xGMI), the distinct candidate scans replicated likewise, candidates block-sharded [r N / G, (r + 1) N / G) over the
 * devices, one host thread per device, results in candidate order.  Alignments are independent, so there is no collective on the
 * data path and the result of a candidate does not depend on G (tests: bit-identical to one lsm2d_align_batch).  Role assignment of
 * the reference tracker / loop detector: fixed = the candidate's scan, moving = the submap; one laser slice.
 * device_ids may name a device more than once (a rehearsal of G > 1 on a one-GPU box). */
typedef struct lsm2d_sweep lsm2d_sweep;
int     lsm2d_sweep_create(const int32_t* device_ids, int32_t n_devices, lsm2d_sweep** out_sweep);
void    lsm2d_sweep_destroy(lsm2d_sweep* sweep);
int32_t lsm2d_sweep_num_devices(const lsm2d_sweep* sweep);
const char* lsm2d_sweep_last_error(const lsm2d_sweep* sweep);
/* "peer_copy": 0 = automatic (default): a replica on another device is filled device to device when hipDeviceCanAccessPeer says the two
 * devices reach each other (peer access is enabled for the pair once), else -- and whenever a peer copy fails -- straight from the caller's
 * host buffer; 1 = always from the host buffer (what a node without xGMI / with peer access disabled gets; results do not depend on it).
 * Read-only: "replicas_by_peer_copy", "replicas_through_host", "replicas_same_device" -- how the replicas of the most recent
 * lsm2d_sweep_set_map / lsm2d_sweep_set_scans were filled. */
int     lsm2d_sweep_set_option(lsm2d_sweep* sweep, const char* key, int64_t value);
int     lsm2d_sweep_get_option(const lsm2d_sweep* sweep, const char* key, int64_t* out_value);
int     lsm2d_sweep_set_map(lsm2d_sweep* sweep, const float* map_xynn, int64_t n_points);
/* the distinct scans the candidates refer to: packed back to back, offsets[n_scans + 1] */
int     lsm2d_sweep_set_scans(lsm2d_sweep* sweep, const float* scans_xynn, const int32_t* offsets, int32_t n_scans);
/* candidate i = (scan scan_index[i] -- or scan i when scan_index is NULL -- , init_pose[i]); out_last_stats: the statistics of the last
 * iteration each candidate started (what the acceptance test of MULTI.json:979-985 reads); out_H / out_iterations / out_last_stats may be NULL */
int     lsm2d_sweep_align(lsm2d_sweep* sweep, const lsm2d_aligner_params* aligner, const lsm2d_slice_params* slice, int32_t n_candidates,
                          const int32_t* scan_index, const float* init_pose, float* out_pose, float* out_H, int32_t* out_status,
                          int32_t* out_iterations, lsm2d_iteration_stats* out_last_stats);

#ifdef __cplusplus
}
#endif
#endif /* LSM2D_H */
