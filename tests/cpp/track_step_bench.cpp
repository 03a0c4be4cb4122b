// Latency of one tracker step through the C ABI alone (no Python in the loop): what an SRRG-side C++ adapter pays.
//   track_step_bench map.bin scan0.bin scan1.bin gx gy gth steps mode [ranges0.bin ranges1.bin amin amax]
//   mode 0: every call synchronous; 1: asynchronous clip / upload / merge; 2: as 1, but raw ranges in (preprocessed on the device)
// Every step: clip the device-resident local map around the guess, upload the two scans, align (2 laser slices with their
// extrinsics + odometry prior, 10 iterations, 721 columns: the MULTI.json parameters), compose the corrected pose on the host,
// merge both scans.  The map is re-uploaded every `reset` steps so it stays the size a local map between key frames has.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "lsm2d.h"

static std::vector<float> read_bin(const char* path) {
  FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f) / 4; fseek(f, 0, SEEK_SET);
  std::vector<float> v((size_t) n);
  if (n && fread(v.data(), 4, (size_t) n, f) != (size_t) n) exit(2);
  fclose(f); return v;
}
static void compose(const double a[3], const double b[3], double o[3]) {
  const double c = cos(a[2]), s = sin(a[2]);
  o[0] = a[0] + c * b[0] - s * b[1]; o[1] = a[1] + s * b[0] + c * b[1]; o[2] = a[2] + b[2];
}
static void inverse(const double a[3], double o[3]) {
  const double c = cos(a[2]), s = sin(a[2]);
  o[0] = -(c * a[0] + s * a[1]); o[1] = -(-s * a[0] + c * a[1]); o[2] = -a[2];
}
#define CK(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, lsm2d_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 9) { fprintf(stderr, "usage: %s map.bin scan0.bin scan1.bin gx gy gth steps async\n", argv[0]); return 2; }
  const std::vector<float> map = read_bin(argv[1]), s0 = read_bin(argv[2]), s1 = read_bin(argv[3]);
  const double guess[3] = {atof(argv[4]), atof(argv[5]), atof(argv[6])};
  const int steps = atoi(argv[7]), mode = atoi(argv[8]); const bool async = mode != 0;
  std::vector<float> r0, r1; lsm2d_preprocessor pp; memset(&pp, 0, sizeof pp);
  if (mode == 2) {
    if (argc < 13) { fprintf(stderr, "mode 2 needs ranges0.bin ranges1.bin amin amax\n"); return 2; }
    r0 = read_bin(argv[9]); r1 = read_bin(argv[10]);
    pp.n_beams = (int32_t) r0.size(); pp.angle_min = (float) atof(argv[11]); pp.angle_max = (float) atof(argv[12]);
    pp.range_min = 0.3f; pp.range_max = 20.0f; pp.normal_point_distance = 0.3f; pp.normal_min_points = 5; pp.voxelize_resolution = 0.02f;
  }
  lsm2d_context* ctx = nullptr;
  CK(lsm2d_create(0, nullptr, &ctx));
  const bool single_merges = getenv("LSM2D_TSB_SINGLE_MERGES") != nullptr;      // one lsm2d_merge_scene call per scan instead of one lsm2d_merge_scenes
  const bool timing = getenv("LSM2D_TSB_TIMING") != nullptr;        // kernel events cost ~30 us per step: off unless asked for
  if (timing) CK(lsm2d_set_option(ctx, "kernel_timing", 1));
  if (const char* o = getenv("LSM2D_TSB_OPTIONS")) {                // "key=value,key=value": context options (A/B runs of tuning knobs)
    std::string all(o); size_t at = 0;
    while (at < all.size()) {
      size_t e = all.find(',', at); if (e == std::string::npos) e = all.size();
      const std::string kv = all.substr(at, e - at); const size_t q = kv.find('=');
      if (q != std::string::npos) CK(lsm2d_set_option(ctx, kv.substr(0, q).c_str(), atoll(kv.c_str() + q + 1)));
      at = e + 1;
    }
  }
  lsm2d_cloudset *local_map, *clipped, *m0, *m1;
  CK(lsm2d_cloudset_create_reserved(ctx, 60000, &local_map));
  CK(lsm2d_cloudset_create_reserved(ctx, 721, &clipped));
  CK(lsm2d_cloudset_create_reserved(ctx, 1024, &m0)); CK(lsm2d_cloudset_create_reserved(ctx, 1024, &m1));
  lsm2d_projector pr = {721, -(float) M_PI, (float) M_PI, 0.3f, 20.0f, 0.0f};
  const float S0[3] = {0.2f, 0.1f, 0.1f}, S1[3] = {-0.3f, 0.0f, (float) M_PI};
  lsm2d_slice_params sl[2]; memset(sl, 0, sizeof sl);
  for (int i = 0; i < 2; ++i) {
    sl[i].finder = LSM2D_FINDER_PROJECTIVE; sl[i].projector = pr; sl[i].point_distance = 0.5f; sl[i].normal_cos = i ? 0.8f : 0.9f;
    sl[i].robustifier = i ? LSM2D_ROBUST_NONE : LSM2D_ROBUST_CAUCHY; sl[i].chi_threshold = 0.01f; sl[i].min_num_correspondences = 5;
    memcpy(sl[i].sensor_in_robot, i ? S1 : S0, sizeof S0);
  }
  // LSM2D_TSB_FINDER=kdtree | nn | distmap: the same step with a point-query finder in both slices (CorrespondenceFinderKDTree2D as the reference
  // runs it -- its tree rebuilt for every new scan, correspondence_finder_kd_tree_2d.cpp:6-8,31-38 --, the exact grid search, the distance map): what
  // the live tracker pays for the finders' reset() at scan rate
  if (const char* fk = getenv("LSM2D_TSB_FINDER")) {
    const int kind = !strcmp(fk, "kdtree") ? LSM2D_FINDER_KDTREE : (!strcmp(fk, "nn") ? LSM2D_FINDER_NN : (!strcmp(fk, "distmap") ? LSM2D_FINDER_DISTMAP : LSM2D_FINDER_PROJECTIVE));
    for (int i = 0; i < 2; ++i) { sl[i].finder = kind; sl[i].max_distance = 0.3f; sl[i].resolution = 0.05f; sl[i].kd_max_leaf_range = 0.01f; sl[i].kd_min_leaf_points = 20; }
  }
  lsm2d_aligner_params ap = {10, 10, 0.0f, 0.0f, 0, 0};
  lsm2d_prior prior; memset(&prior, 0, sizeof prior); prior.omega[0] = prior.omega[4] = prior.omega[8] = 100.0f;
  const lsm2d_cloudset* fixed[2] = {m0, m1}; const lsm2d_cloudset* moving[2] = {clipped, clipped};
  const float x0[3] = {0, 0, 0};
  lsm2d_batch b; memset(&b, 0, sizeof b);
  b.n_alignments = 1; b.n_slices = 2; b.slices = sl; b.fixed = fixed; b.moving = moving; b.init_pose = x0; b.prior = &prior;
  const float g32[3] = {(float) guess[0], (float) guess[1], (float) guess[2]};
  double est[3] = {0, 0, 0}, est_fresh[3] = {0, 0, 0}; int status = -1; float ms_kernel = 0.0f, kernel_sum = 0.0f;
  const int reset = 50;
  std::chrono::duration<double> total(0), ph[4] = {};      // host time inside: clip call, scan calls, aligner call (includes the wait), pose + merge calls
  for (int k = -20; k < steps; ++k) {          // 20 warm-up steps
    if ((k + 20) % reset == 0) { CK(lsm2d_cloudset_upload(local_map, map.data(), (int64_t) (map.size() / 4))); CK(lsm2d_synchronize(ctx)); }
    const auto t0 = std::chrono::steady_clock::now();
    int32_t n_clip = 0, n_map = 0;
    CK(lsm2d_clip_scene(ctx, &pr, local_map, 0, g32, S0, clipped, async ? nullptr : &n_clip, nullptr));
    const auto tc = std::chrono::steady_clock::now();
    if (mode == 2) { CK(lsm2d_preprocess_scan_into(ctx, &pp, r0.data(), m0)); CK(lsm2d_preprocess_scan_into(ctx, &pp, r1.data(), m1)); }
    else { CK(lsm2d_cloudset_upload(m0, s0.data(), (int64_t) (s0.size() / 4))); CK(lsm2d_cloudset_upload(m1, s1.data(), (int64_t) (s1.size() / 4))); }
    const auto tu = std::chrono::steady_clock::now();
    float x[3];
    CK(lsm2d_align_batch(ctx, &ap, &b, x, nullptr, &status, nullptr, nullptr));
    const auto ta = std::chrono::steady_clock::now();
    if (k >= 0) { ph[0] += tc - t0; ph[1] += tu - tc; ph[2] += ta - tu; }
    if (k >= 0 && timing) { lsm2d_last_kernel_ms(ctx, &ms_kernel); kernel_sum += ms_kernel; }
    const double xd[3] = {x[0], x[1], x[2]}; double xi[3]; inverse(xd, xi); compose(guess, xi, est);
    if ((k + 20) % reset == 0) memcpy(est_fresh, est, sizeof est);      // the step right after a map reset: comparable with the oracle
    float mf[6];                                 // both scans at the corrected pose: one call (one launch: lsm2d_merge_scenes)
    for (int i = 0; i < 2; ++i) {
      const double Sd[3] = {(i ? S1 : S0)[0], (i ? S1 : S0)[1], (i ? S1 : S0)[2]}; double mis[3]; compose(est, Sd, mis);
      mf[3 * i] = (float) mis[0]; mf[3 * i + 1] = (float) mis[1]; mf[3 * i + 2] = (float) mis[2];
    }
    if (single_merges) { for (int i = 0; i < 2; ++i) CK(lsm2d_merge_scene(ctx, &pr, local_map, i ? m1 : m0, 0, mf + 3 * i, 0.2f, async ? nullptr : &n_map, nullptr)); }
    else { const lsm2d_cloudset* ms[2] = {m0, m1}; CK(lsm2d_merge_scenes(ctx, &pr, local_map, 2, ms, nullptr, mf, 0.2f, async ? nullptr : &n_map, nullptr)); }
    const auto t1 = std::chrono::steady_clock::now();
    if (k >= 0) { total += t1 - t0; ph[3] += t1 - ta; }
  }
  CK(lsm2d_synchronize(ctx));
  printf("{\"steps\": %d, \"mode\": %d, \"asynchronous\": %s, \"ms_per_step_wall\": %.5f, \"align_kernel_ms_per_step\": %.5f, \"host_us_in_calls\": {\"clip\": %.2f, \"scans\": %.2f, \"align_incl_wait\": %.2f, \"pose_and_merges\": %.2f}, \"status\": %d, \"map_points\": %lld, \"est_on_fresh_map\": [%.9f, %.9f, %.9f]}\n",
         steps, mode, async ? "true" : "false", 1e3 * total.count() / steps, kernel_sum / steps,
         1e6 * ph[0].count() / steps, 1e6 * ph[1].count() / steps, 1e6 * ph[2].count() / steps, 1e6 * ph[3].count() / steps, status, (long long) lsm2d_cloudset_num_points(local_map), est_fresh[0], est_fresh[1], est_fresh[2]);
  lsm2d_cloudset_destroy(m0); lsm2d_cloudset_destroy(m1); lsm2d_cloudset_destroy(clipped); lsm2d_cloudset_destroy(local_map);
  lsm2d_destroy(ctx);
  return 0;
}
