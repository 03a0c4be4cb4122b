// One step of BASELINE configs[1] through the C ABI alone (no Python in the loop): what an SRRG-side C++ caller of lsm2d_align_batch pays per
// batch of candidates -- MultiAligner2D::compute per candidate in the reference's loop-closure / relocalisation loops
// (configurations/stage_segway_double_config_MULTI.json:964-986).
//   batch_step_bench map.bin scans.bin offsets.bin x0.bin steps warmup iterations beams
// map / scans: float32 [N,4]; offsets: int32 [n+1]; x0: float32 [n,3] (map in scan: role A, fixed = scan, moving = map).  Prints one JSON line.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "lsm2d.h"

template <typename T> static std::vector<T> read_bin(const char* path) {
  FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f) / (long) sizeof(T); fseek(f, 0, SEEK_SET);
  std::vector<T> v((size_t) n);
  if (n && fread(v.data(), sizeof(T), (size_t) n, f) != (size_t) n) exit(2);
  fclose(f); return v;
}
#define CK(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, lsm2d_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 9) { fprintf(stderr, "usage: %s map.bin scans.bin offsets.bin x0.bin steps warmup iterations beams\n", argv[0]); return 2; }
  const std::vector<float> map = read_bin<float>(argv[1]), scans = read_bin<float>(argv[2]), x0 = read_bin<float>(argv[4]);
  const std::vector<int32_t> offs = read_bin<int32_t>(argv[3]);
  const int steps = atoi(argv[5]), warmup = atoi(argv[6]), its = atoi(argv[7]), beams = atoi(argv[8]);
  const int n = (int) offs.size() - 1;
  if (n < 1 || (int) x0.size() != 3 * n) { fprintf(stderr, "inconsistent inputs\n"); return 2; }
  lsm2d_context* ctx = nullptr;
  CK(lsm2d_create(0, nullptr, &ctx));
  lsm2d_cloudset *map_set, *scan_set;
  CK(lsm2d_cloudset_create(ctx, map.data(), nullptr, 1, (int64_t) (map.size() / 4), &map_set));
  CK(lsm2d_cloudset_create(ctx, scans.data(), offs.data(), n, (int64_t) (scans.size() / 4), &scan_set));
  lsm2d_slice_params sl; memset(&sl, 0, sizeof sl);
  sl.finder = LSM2D_FINDER_PROJECTIVE;
  sl.projector.canvas_cols = beams; sl.projector.angle_min = -(float) M_PI; sl.projector.angle_max = (float) M_PI; sl.projector.range_min = 0.3f; sl.projector.range_max = 30.0f;
  sl.point_distance = 0.5f; sl.normal_cos = 0.8f; sl.robustifier = LSM2D_ROBUST_NONE; sl.min_num_correspondences = 10;
  lsm2d_aligner_params ap; memset(&ap, 0, sizeof ap); ap.max_iterations = its; ap.min_num_inliers = 10;
  const lsm2d_cloudset* fixed[1] = {scan_set}; const lsm2d_cloudset* moving[1] = {map_set};
  lsm2d_batch b; memset(&b, 0, sizeof b);
  b.n_alignments = n; b.n_slices = 1; b.slices = &sl; b.fixed = fixed; b.moving = moving; b.init_pose = x0.data();
  std::vector<float> pose((size_t) 3 * n), H((size_t) 9 * n); std::vector<int32_t> status((size_t) n), iters((size_t) n);
  for (int k = 0; k < warmup; ++k) CK(lsm2d_align_batch(ctx, &ap, &b, pose.data(), H.data(), status.data(), iters.data(), nullptr));
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < steps; ++k) CK(lsm2d_align_batch(ctx, &ap, &b, pose.data(), H.data(), status.data(), iters.data(), nullptr));
  const std::chrono::duration<double> dt = std::chrono::steady_clock::now() - t0;
  // the same step with two batches in flight (lsm2d_align_batch_begin / _wait: begin(k) ; wait(k - 1)): each lane launches on a stream of its own, the younger
  // launch's workgroups fill the older one's tail; every run the same bits as the synchronous step
  const std::vector<float> want_pose = pose; const std::vector<int32_t> want_status = status;
  double pipelined_ms = 0.0; long differed = 0;
  {
    std::vector<float> p2((size_t) 3 * n), H2((size_t) 9 * n); std::vector<int32_t> st2((size_t) n);
    lsm2d_pending* pend[2] = {nullptr, nullptr};
    const int reps = steps + 30;
    std::chrono::steady_clock::time_point tp0;
    for (int k = 0; k < reps; ++k) {
      if (k == 30) tp0 = std::chrono::steady_clock::now();
      CK(lsm2d_align_batch_begin(ctx, &ap, &b, 0, &pend[k & 1]));
      if (k > 0) {
        CK(lsm2d_align_batch_wait(pend[(k - 1) & 1], p2.data(), H2.data(), st2.data(), nullptr, nullptr)); pend[(k - 1) & 1] = nullptr;
        if (memcmp(p2.data(), want_pose.data(), sizeof(float) * 3 * (size_t) n) || memcmp(st2.data(), want_status.data(), sizeof(int32_t) * (size_t) n)) ++differed;
      }
    }
    CK(lsm2d_align_batch_wait(pend[(reps - 1) & 1], p2.data(), H2.data(), st2.data(), nullptr, nullptr));
    pipelined_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count() / steps;
  }
  int ok = 0; for (int i = 0; i < n; ++i) ok += status[i] == 0;
  printf("{\"alignments\": %d, \"steps\": %d, \"ms_per_step_wall\": %.5f, \"alignments_per_s\": %.1f, \"ms_per_step_two_in_flight\": %.5f, \"alignments_per_s_two_in_flight\": %.1f, "
         "\"runs_in_flight_that_differed\": %ld, \"status_ok\": %d, \"pose0\": [%.9g, %.9g, %.9g]}\n",
         n, steps, 1e3 * dt.count() / steps, (double) n * steps / dt.count(), pipelined_ms, 1e3 * (double) n / pipelined_ms, differed, ok, pose[0], pose[1], pose[2]);
  lsm2d_cloudset_destroy(scan_set); lsm2d_cloudset_destroy(map_set); lsm2d_destroy(ctx);
  return 0;
}
