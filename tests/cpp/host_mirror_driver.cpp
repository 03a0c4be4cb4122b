// Drives the C++ host mirror the way the reference's own drivers do
// (apps/visual_test_correspondence_finder_projective_2d.cpp:71-97, apps/visual_test_aligner_2d.cpp:102-156):
//   host_mirror_driver fixed.bin moving.bin x y theta cols iterations
// reads two float32 [N,4] clouds, runs the finder at the given pose and the aligner from it, prints JSON.
#include <cstdio>
#include <cstdlib>
#include <lsm2d.hpp>

using namespace lsm2d_host;

static Vector3f x_inv(const Vector3f& a) {     // (R,t)^-1 in (x, y, theta) form: the sensor pose in the map for pose = map-in-sensor
  const float c = cosf(a[2]), s = sinf(a[2]);
  return Vector3f{{-(c * a[0] + s * a[1]), -(-s * a[0] + c * a[1]), -a[2]}};
}

static PointNormal2fVectorCloud read_cloud(const char* path) {
  FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f) / (long) sizeof(PointNormal2f); fseek(f, 0, SEEK_SET);
  PointNormal2fVectorCloud c((size_t) n);
  if (n && fread(c.data(), sizeof(PointNormal2f), (size_t) n, f) != (size_t) n) exit(2);
  fclose(f); return c;
}

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: %s fixed.bin moving.bin x y theta cols iterations\n", argv[0]); return 2; }
  try {
    Context ctx(0);
    PointNormal2fVectorCloud fixed = read_cloud(argv[1]), moving = read_cloud(argv[2]);
    Vector3f pose{{(float) atof(argv[3]), (float) atof(argv[4]), (float) atof(argv[5])}};
    const int cols = atoi(argv[6]), its = atoi(argv[7]);

    // reference error behaviour: compute() without inputs throws
    int threw = 0;
    { CorrespondenceFinderProjective2f empty(ctx); CorrespondenceVector cv; empty.setCorrespondences(&cv);
      try { empty.compute(); } catch (const std::runtime_error&) { threw = 1; } }

    std::shared_ptr<CorrespondenceFinderProjective2f> cf(new CorrespondenceFinderProjective2f(ctx));
    cf->param_projector->param_canvas_cols = cols; cf->param_projector->param_range_max = 30.f;
    cf->param_projector->param_angle_col_min = -(float) M_PI; cf->param_projector->param_angle_col_max = (float) M_PI;
    CorrespondenceVector correspondences;
    cf->setFixed(&fixed); cf->setMoving(&moving); cf->setLocalMapInSensor(pose); cf->setCorrespondences(&correspondences);
    cf->compute();

    AlignerSliceProcessorLaser2DPtr slice(new AlignerSliceProcessorLaser2D);
    slice->param_finder = cf; slice->param_min_num_correspondences = 10;
    MultiAligner2D::PropertyContainer fixed_props{{"points", &fixed}}, moving_props{{"points", &moving}};
    MultiAligner2D aligner(ctx);
    aligner.param_max_iterations = its; aligner.param_slice_processors.push_back(slice);
    aligner.setFixed(&fixed_props); aligner.setMoving(&moving_props); aligner.setMovingInFixed(pose);
    aligner.compute();

    // round 4: what the reference leaves in slice->correspondences() (apps/visual_test_aligner_2d.cpp:129-143), the pair digest of the last
    // iteration, and MultiAligner2D's inlier options (MULTI.json:606-610) on a robustified slice
    size_t n_kept = 0, n_all = 0, its_runs = 0; unsigned long long dig_stats = 0, dig_pairs = 0; int last_inl = 0;
    { AlignerSliceProcessorLaser2DPtr rs(new AlignerSliceProcessorLaser2D);
      rs->param_finder = cf; rs->param_min_num_correspondences = 10;
      rs->param_robustifier.reset(new RobustifierCauchy); rs->param_robustifier->param_chi_threshold = 2e-5f;
      MultiAligner2D al2(ctx);
      al2.param_max_iterations = its; al2.param_slice_processors.push_back(rs); al2.store_correspondences = true;
      al2.setFixed(&fixed_props); al2.setMoving(&moving_props); al2.setMovingInFixed(pose);
      al2.compute();
      n_all = al2.correspondences(0).size();
      const lsm2d_iteration_stats& st = al2.iterationStats().back();
      dig_stats = ((unsigned long long) st.pair_digest_hi << 32) | st.pair_digest_lo;
      for (const auto& p : al2.correspondences(0)) dig_pairs += lsm2d_pair_hash(0u, (uint32_t) p.fixed_idx, (uint32_t) p.moving_idx);
      al2.param_keep_only_inlier_correspondences = true; al2.param_enable_inlier_only_runs = true;
      al2.setMovingInFixed(pose); al2.compute();
      n_kept = al2.correspondences(0).size(); its_runs = al2.iterationStats().size(); last_inl = al2.iterationStats().back().n_inliers;
    }

    // the other two finders and the mapping steps, through their reference-named classes
    CorrespondenceVector nn_pairs, dm_pairs, kd_pairs;
    { CorrespondenceFinderKDTree2D kd(ctx, "exact"); kd.param_max_distance_m = 0.3f;      // the exact grid search
      kd.setFixed(&fixed); kd.setMoving(&moving); kd.setLocalMapInSensor(pose); kd.setCorrespondences(&nn_pairs); kd.compute(); }
    { CorrespondenceFinderKDTree2D kd(ctx); kd.param_max_distance_m = 0.3f; kd.param_max_leaf_range = 0.02f; kd.param_min_leaf_points = 9;      // the reference's own tree (the default)
      kd.setFixed(&fixed); kd.setMoving(&moving); kd.setLocalMapInSensor(pose); kd.setCorrespondences(&kd_pairs); kd.compute(); }
    { CorrespondenceFinderNN2D dm(ctx); dm.param_max_distance_m = 0.5f; dm.param_resolution = 0.1f;
      dm.setFixed(&fixed); dm.setMoving(&moving); dm.setLocalMapInSensor(pose); dm.setCorrespondences(&dm_pairs); dm.compute(); }
    ReservedCloud scene(ctx, (int64_t) moving.size() + 4096), clipped(ctx, cols);
    scene.upload(moving);
    SceneClipperProjective2D clipper(ctx);
    clipper.param_voxelize_resolution = 0.f;      // as both shipped configurations set it (the class default is 0.1)
    clipper.param_projector->param_canvas_cols = cols; clipper.param_projector->param_range_max = 30.f;
    clipper.param_projector->param_angle_col_min = -(float) M_PI; clipper.param_projector->param_angle_col_max = (float) M_PI;
    clipper.setFullScene(&scene); clipper.setClippedSceneInRobot(&clipped);
    Vector3f robot{{-x_inv(pose)[0], -x_inv(pose)[1], -pose[2]}};
    clipper.setRobotInLocalMap(x_inv(pose));
    const int n_clipped = clipper.compute();
    MergerProjective2D merger(ctx);
    *merger.param_projector = *clipper.param_projector;
    merger.setScene(&scene); merger.setMeasurement(&fixed); merger.setMeasurementInScene(x_inv(pose));
    const int merged_size = merger.compute();
    (void) robot;

    printf("{\"threw_on_missing_inputs\": %d, \"n_nn\": %zu, \"n_kdtree\": %zu, \"n_distmap\": %zu, \"n_clipped\": %d, \"merged_size\": %d, \"merge_counts\": [%d,%d,%d], \"n_pairs\": %zu, \"pairs\": [",
           threw, nn_pairs.size(), kd_pairs.size(), dm_pairs.size(), n_clipped, merged_size, merger.counts[0], merger.counts[1], merger.counts[2], correspondences.size());
    for (size_t i = 0; i < correspondences.size(); ++i) printf("%s[%d,%d]", i ? "," : "", correspondences[i].fixed_idx, correspondences[i].moving_idx);
    const Vector3f& x = aligner.movingInFixed();
    printf("], \"status\": %d, \"pose\": [%.9g, %.9g, %.9g], \"iterations\": %zu, \"last_n_corr\": %d, \"n_all\": %zu, \"digest_matches\": %d, \"n_kept\": %zu, "
           "\"iterations_with_inlier_runs\": %zu, \"last_inliers_with_inlier_runs\": %d}\n", aligner.status(), x[0], x[1], x[2],
           aligner.iterationStats().size(), aligner.iterationStats().empty() ? 0 : aligner.iterationStats().back().n_correspondences,
           n_all, (int) (dig_stats == dig_pairs && dig_stats != 0), n_kept, its_runs, last_inl);
  } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 0;
}
