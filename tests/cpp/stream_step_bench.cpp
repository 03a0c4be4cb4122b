// The streamed pipeline through the C ABI alone (no Python in the loop): what an SRRG-side C++ host pays per batch of FRESH LaserMessages -- per incoming message
// RawDataPreprocessorProjective2D::compute (sensor_processing/raw_data_preprocessor_projective_2d.cpp:13-51) feeding MultiAligner2D::compute
// (apps/visual_test_aligner_2d.cpp:123-156) -- with one batch in flight: lsm2d_preprocess_scans_refill into one of two scan sets, lsm2d_align_batch_begin for
// step i, lsm2d_align_batch_wait for step i - 1.  Every distinct batch's streamed result is compared BIT FOR BIT with the synchronous calls on the same ranges.
//   stream_step_bench map.bin ranges.bin x0.bin n_scans n_beams n_batches steps warmup iterations angle_min angle_max [refill_ahead = 1]
// map: float32 [N,4]; ranges: float32 [n_batches][n_scans][n_beams]; x0: float32 [n_batches][n_scans][3].  Prints one JSON line.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "lsm2d.h"
#include "lsm2d.hpp"      // the C++ mirror's LaserMessageBatchStream: the same pipeline as a class, checked below against the same bits

template <typename T> static std::vector<T> read_bin(const char* path) {
  FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f) / (long) sizeof(T); fseek(f, 0, SEEK_SET);
  std::vector<T> v((size_t) n);
  if (n && fread(v.data(), sizeof(T), (size_t) n, f) != (size_t) n) exit(2);
  fclose(f); return v;
}
#define CK(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, lsm2d_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 12) { fprintf(stderr, "usage: %s map.bin ranges.bin x0.bin n_scans n_beams n_batches steps warmup iterations angle_min angle_max\n", argv[0]); return 2; }
  const std::vector<float> map = read_bin<float>(argv[1]), ranges = read_bin<float>(argv[2]), x0 = read_bin<float>(argv[3]);
  const int n = atoi(argv[4]), nb = atoi(argv[5]), nbatch = atoi(argv[6]), steps = atoi(argv[7]), warmup = atoi(argv[8]), its = atoi(argv[9]);
  const float a0 = (float) atof(argv[10]), a1 = (float) atof(argv[11]);
  if (n < 1 || nbatch < 2 || ranges.size() != (size_t) nbatch * n * nb || x0.size() != (size_t) nbatch * n * 3) { fprintf(stderr, "inconsistent inputs\n"); return 2; }
  lsm2d_context* ctx = nullptr;
  CK(lsm2d_create(0, nullptr, &ctx));
  if (const char* e = getenv("LSM2D_SSB_OPTIONS")) {      // "key=value,key=value": context options for A/B runs
    std::string o(e); size_t p0 = 0;
    while (p0 < o.size()) {
      size_t p1 = o.find(',', p0); if (p1 == std::string::npos) p1 = o.size();
      const std::string kv = o.substr(p0, p1 - p0); const size_t eq = kv.find('=');
      if (eq != std::string::npos) CK(lsm2d_set_option(ctx, kv.substr(0, eq).c_str(), atoll(kv.c_str() + eq + 1)));
      p0 = p1 + 1;
    }
  }
  // the messages wait in pinned host memory, as a driver's receive buffers would
  float* pinned = nullptr;
  if (hipHostMalloc((void**) &pinned, ranges.size() * sizeof(float), hipHostMallocDefault) != hipSuccess) { fprintf(stderr, "hipHostMalloc failed\n"); return 1; }
  memcpy(pinned, ranges.data(), ranges.size() * sizeof(float));
  lsm2d_cloudset* map_set;
  CK(lsm2d_cloudset_create(ctx, map.data(), nullptr, 1, (int64_t) (map.size() / 4), &map_set));
  lsm2d_preprocessor pp; memset(&pp, 0, sizeof pp);
  pp.n_beams = nb; pp.angle_min = a0; pp.angle_max = a1; pp.range_min = 0.3f; pp.range_max = 30.0f; pp.normal_point_distance = 0.3f; pp.normal_min_points = 5; pp.voxelize_resolution = 0.02f;
  lsm2d_slice_params sl; memset(&sl, 0, sizeof sl);
  sl.finder = LSM2D_FINDER_PROJECTIVE;
  sl.projector.canvas_cols = nb; sl.projector.angle_min = -(float) M_PI; sl.projector.angle_max = (float) M_PI; sl.projector.range_min = 0.3f; sl.projector.range_max = 30.0f;
  sl.point_distance = 0.5f; sl.normal_cos = 0.8f; sl.robustifier = LSM2D_ROBUST_NONE; sl.min_num_correspondences = 10;
  lsm2d_aligner_params ap; memset(&ap, 0, sizeof ap); ap.max_iterations = its; ap.min_num_inliers = 10;
  const size_t rstride = (size_t) n * nb, xstride = (size_t) n * 3;
  // the synchronous calls on every distinct batch: what each streamed step must reproduce bit for bit
  std::vector<std::vector<float>> want_pose((size_t) nbatch, std::vector<float>(xstride)), want_H((size_t) nbatch, std::vector<float>((size_t) 9 * n));
  std::vector<std::vector<int32_t>> want_st((size_t) nbatch, std::vector<int32_t>((size_t) n));
  for (int k = 0; k < nbatch; ++k) {
    lsm2d_cloudset* s = nullptr;
    CK(lsm2d_preprocess_scans(ctx, &pp, pinned + k * rstride, n, &s));
    const lsm2d_cloudset* fx[1] = {s}; const lsm2d_cloudset* mv[1] = {map_set};
    lsm2d_batch b; memset(&b, 0, sizeof b); b.n_alignments = n; b.n_slices = 1; b.slices = &sl; b.fixed = fx; b.moving = mv; b.init_pose = x0.data() + k * xstride;
    CK(lsm2d_align_batch(ctx, &ap, &b, want_pose[k].data(), want_H[k].data(), want_st[k].data(), nullptr, nullptr));
    lsm2d_cloudset_destroy(s);
  }
  // ahead = 1 (default): THREE scan sets, per step  begin(i) ; refill(set of step i + 1) ; wait(i - 1)  -- the preprocessing of step i + 1 is queued a whole launch
  // before the estimate that reads its clouds (include/lsm2d.h at lsm2d_align_batch_begin); ahead = 0: two sets, refill(i) ; begin(i) ; wait(i - 1)
  const int ahead = argc > 12 ? atoi(argv[12]) : 1, nsets = 2 + (ahead ? 1 : 0);
  lsm2d_cloudset* sets[3] = {nullptr, nullptr, nullptr};
  for (int k = 0; k < nsets; ++k) CK(lsm2d_preprocess_scans(ctx, &pp, pinned + (size_t) (k % nbatch) * rstride, n, &sets[k]));
  std::vector<float> pose[2] = {std::vector<float>(xstride), std::vector<float>(xstride)}, H[2] = {std::vector<float>((size_t) 9 * n), std::vector<float>((size_t) 9 * n)};
  std::vector<int32_t> status[2] = {std::vector<int32_t>((size_t) n), std::vector<int32_t>((size_t) n)};
  lsm2d_pending* pending[2] = {nullptr, nullptr};
  long checked = 0, differed = 0;
  long step_i = 0;
  double host_s[3] = {0.0, 0.0, 0.0};      // seconds inside refill / begin / wait
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto refill = [&](long i) -> int {
    const auto t = now();
    CK(lsm2d_preprocess_scans_refill(ctx, &pp, pinned + (size_t) (i % nbatch) * rstride, n, sets[i % nsets]));
    host_s[0] += std::chrono::duration<double>(now() - t).count();
    return 0;
  };
  auto one_step = [&]() -> int {
    const int k = (int) (step_i % nbatch), lane = (int) (step_i & 1);
    if (!ahead || step_i == 0) if (refill(step_i)) return 1;
    const lsm2d_cloudset* fx[1] = {sets[step_i % nsets]}; const lsm2d_cloudset* mv[1] = {map_set};
    lsm2d_batch b; memset(&b, 0, sizeof b); b.n_alignments = n; b.n_slices = 1; b.slices = &sl; b.fixed = fx; b.moving = mv; b.init_pose = x0.data() + k * xstride;
    auto t = now();
    CK(lsm2d_align_batch_begin(ctx, &ap, &b, 0, &pending[lane]));
    host_s[1] += std::chrono::duration<double>(now() - t).count();
    if (ahead) if (refill(step_i + 1)) return 1;      // (the set batch step_i - 2 read: waited for in the previous step)
    if (step_i > 0) {
      const int pl = lane ^ 1, pk = (int) ((step_i - 1) % nbatch);
      t = now();
      CK(lsm2d_align_batch_wait(pending[pl], pose[pl].data(), H[pl].data(), status[pl].data(), nullptr, nullptr)); pending[pl] = nullptr;
      host_s[2] += std::chrono::duration<double>(now() - t).count();
      ++checked;
      if (memcmp(pose[pl].data(), want_pose[pk].data(), xstride * sizeof(float)) || memcmp(H[pl].data(), want_H[pk].data(), (size_t) 9 * n * sizeof(float)) ||
          memcmp(status[pl].data(), want_st[pk].data(), (size_t) n * sizeof(int32_t))) ++differed;
    }
    ++step_i;
    return 0;
  };
  for (int k = 0; k < warmup; ++k) if (one_step()) return 1;
  host_s[0] = host_s[1] = host_s[2] = 0.0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < steps; ++k) if (one_step()) return 1;
  const std::chrono::duration<double> dt = std::chrono::steady_clock::now() - t0;
  { const int pl = (int) ((step_i - 1) & 1); CK(lsm2d_align_batch_wait(pending[pl], pose[pl].data(), H[pl].data(), status[pl].data(), nullptr, nullptr)); }
  // the resident-input step on the same scans (lsm2d_align_batch; the library keeps the placement of a batch that comes again)
  const lsm2d_cloudset* fx[1] = {sets[0]}; const lsm2d_cloudset* mv[1] = {map_set};
  CK(lsm2d_synchronize(ctx));
  const double host_us[3] = {1e6 * host_s[0] / steps, 1e6 * host_s[1] / steps, 1e6 * host_s[2] / steps};
  CK(lsm2d_preprocess_scans_refill(ctx, &pp, pinned, n, sets[0]));
  lsm2d_batch b; memset(&b, 0, sizeof b); b.n_alignments = n; b.n_slices = 1; b.slices = &sl; b.fixed = fx; b.moving = mv; b.init_pose = x0.data();
  for (int k = 0; k < 30; ++k) CK(lsm2d_align_batch(ctx, &ap, &b, pose[0].data(), H[0].data(), status[0].data(), nullptr, nullptr));
  const auto t1 = std::chrono::steady_clock::now();
  for (int k = 0; k < steps; ++k) CK(lsm2d_align_batch(ctx, &ap, &b, pose[0].data(), H[0].data(), status[0].data(), nullptr, nullptr));
  const std::chrono::duration<double> dr = std::chrono::steady_clock::now() - t1;
  // the same pipeline through the C++ mirror's class (srrg2_laser_slam_2d_amd/host/lsm2d.hpp), its own context: every batch that comes out has the same bits
  long mirror_checked = 0, mirror_differed = 0;
  try {
    lsm2d_host::Context hc(0);
    lsm2d_host::PointNormal2fVectorCloud mc(map.size() / 4); memcpy(mc.data(), map.data(), map.size() * sizeof(float));
    lsm2d_host::CloudSet hmap(hc, mc);
    lsm2d_host::LaserMessageBatchStream stream(hc, pp, hmap, sl, ap, n);
    auto same = [&](int k) {
      ++mirror_checked;
      if (memcmp(stream.pose[0].data(), want_pose[k].data(), xstride * sizeof(float)) || memcmp(stream.information[0].data(), want_H[k].data(), (size_t) 9 * n * sizeof(float)) ||
          memcmp(stream.status.data(), want_st[k].data(), (size_t) n * sizeof(int32_t))) ++mirror_differed;
    };
    const int pushes = 2 * nbatch + 1;
    for (int i = 0; i < pushes; ++i)
      if (stream.push(pinned + (size_t) (i % nbatch) * rstride, reinterpret_cast<const lsm2d_host::Vector3f*>(x0.data() + (size_t) (i % nbatch) * xstride))) same((int) (stream.retired() % nbatch));
    while (stream.flush()) same((int) (stream.retired() % nbatch));
    if (stream.retired() != pushes - 1) ++mirror_differed;
  } catch (const std::exception& e) { fprintf(stderr, "LaserMessageBatchStream: %s\n", e.what()); return 1; }
  // the resident-input step PIPELINED: the same batch begun again while its previous run is still on the chip (two lanes, two streams: the younger launch's
  // workgroups fill the older one's tail); every run's results the same bits
  double pipelined_ms = 0.0; long pipe_differed = 0;
  {
    lsm2d_pending* pp2[2] = {nullptr, nullptr};
    const int reps = steps + 30;
    std::chrono::steady_clock::time_point tp0;
    for (int k = 0; k < reps; ++k) {
      if (k == 30) tp0 = std::chrono::steady_clock::now();
      CK(lsm2d_align_batch_begin(ctx, &ap, &b, 0, &pp2[k & 1]));
      if (k > 0) {
        CK(lsm2d_align_batch_wait(pp2[(k - 1) & 1], pose[0].data(), H[0].data(), status[0].data(), nullptr, nullptr)); pp2[(k - 1) & 1] = nullptr;
        if (memcmp(pose[0].data(), want_pose[0].data(), xstride * sizeof(float)) || memcmp(status[0].data(), want_st[0].data(), (size_t) n * sizeof(int32_t))) ++pipe_differed;
      }
    }
    CK(lsm2d_align_batch_wait(pp2[(reps - 1) & 1], pose[0].data(), H[0].data(), status[0].data(), nullptr, nullptr));
    pipelined_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count() / steps;
  }
  int ok = 0; for (int i = 0; i < n; ++i) ok += want_st[0][i] == 0;
  printf("{\"alignments\": %d, \"steps\": %d, \"ms_per_step_streamed\": %.5f, \"alignments_per_s_streamed\": %.1f, \"h2d_GBs\": %.3f, \"refill_ahead\": %d, \"host_us_in_refill_begin_wait\": [%.1f, %.1f, %.1f], \"ms_per_step_resident\": %.5f, \"ms_per_step_resident_pipelined\": %.5f, \"pipelined_runs_that_differed\": %ld, "
         "\"streamed_over_resident\": %.4f, \"steps_checked_bitwise\": %ld, \"steps_that_differed\": %ld, \"mirror_batches_checked\": %ld, \"mirror_batches_that_differed\": %ld, \"status_ok_batch0\": %d}\n",
         n, steps, 1e3 * dt.count() / steps, (double) n * steps / dt.count(), 4.0 * n * nb * steps / dt.count() / 1e9, ahead, host_us[0], host_us[1], host_us[2], 1e3 * dr.count() / steps, pipelined_ms, pipe_differed,
         dr.count() / dt.count(), checked, differed, mirror_checked, mirror_differed, ok);
  for (int k = 0; k < nsets; ++k) lsm2d_cloudset_destroy(sets[k]);
  lsm2d_cloudset_destroy(map_set); lsm2d_destroy(ctx); (void) hipHostFree(pinned);
  return differed || mirror_differed || pipe_differed ? 3 : 0;
}
