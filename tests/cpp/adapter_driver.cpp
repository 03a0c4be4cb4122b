// adapter_driver.cpp -- TEST INFRASTRUCTURE: drives the SRRG-side adapters (adapters/srrg/*) the way the reference drives its own
// classes (apps/visual_test_correspondence_finder_projective_2d.cpp:60-80, apps/visual_test_aligner_2d.cpp:102-156), compiled against
// the stand-in headers of tests/cpp/adapter_shim and linked with the real liblsm2d_hip.so.  Prints one JSON object; the Python test
// (tests/test_gpu_parity.py::test_srrg_adapters_compile_and_run) compares it with the C ABI driven directly.
//   adapter_driver fixed.bin moving.bin x y theta canvas_cols iterations [ranges.bin angle_min angle_max out_cloud.bin]
#include <correspondence_finder_hip_2d.h>
#include <mapping_hip_2d.h>
#include <multi_aligner_hip_2d.h>
#include <raw_data_preprocessor_hip_2d.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

using namespace srrg2_core;
using namespace srrg2_laser_slam_2d;
using namespace srrg2_slam_interfaces;

static PointNormal2fVectorCloud readCloud(const char* path) {
  std::ifstream f(path, std::ios::binary);
  f.seekg(0, std::ios::end); const size_t bytes = (size_t) f.tellg(); f.seekg(0);
  std::vector<float> raw(bytes / 4); f.read((char*) raw.data(), (std::streamsize) bytes);
  PointNormal2fVectorCloud c(raw.size() / 4);
  for (size_t i = 0; i < c.size(); ++i) {
    c[i].coordinates() = Vector2f(raw[4 * i], raw[4 * i + 1]); c[i].normal() = Vector2f(raw[4 * i + 2], raw[4 * i + 3]);
  }
  return c;
}
static std::string pairsJson(const CorrespondenceVector& v) {
  std::ostringstream o; o << "[";
  for (size_t i = 0; i < v.size(); ++i) o << (i ? "," : "") << v[i].fixed_idx << "," << v[i].moving_idx;
  o << "]"; return o.str();
}
static std::string f3(const Vector3f& v) { char b[128]; snprintf(b, sizeof b, "[%.9g,%.9g,%.9g]", v.x(), v.y(), v.z()); return b; }

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: adapter_driver fixed.bin moving.bin x y theta cols iterations\n"); return 2; }
  PointNormal2fVectorCloud fixed = readCloud(argv[1]), moving = readCloud(argv[2]);
  const Vector3f x0((float) atof(argv[3]), (float) atof(argv[4]), (float) atof(argv[5]));
  const int cols = atoi(argv[6]), iters = atoi(argv[7]);
  std::ostringstream out; out << "{";

  auto projector = PointNormal2fProjectorPolarPtr(new PointNormal2fProjectorPolar);
  projector->param_canvas_cols.setValue(cols); projector->param_range_min.setValue(0.3f); projector->param_range_max.setValue(30.f);
  projector->param_angle_col_min.setValue(-3.14159274f); projector->param_angle_col_max.setValue(3.14159274f);

  // ---- plugin interface #1: the three finder siblings (visual_test_correspondence_finder_projective_2d.cpp:74-79)
  {
    CorrespondenceVector corr;
    auto fp = std::make_shared<CorrespondenceFinderHIP2D>();
    fp->param_projector.setValue(projector);
    int threw = 0;
    try { fp->compute(); } catch (const std::runtime_error&) { threw = 1; }                 // missing fixed / moving: the reference throws too
    fp->setFixed(&fixed); fp->setMoving(&moving); fp->setLocalMapInSensor(geometry2d::v2t(x0)); fp->setCorrespondences(&corr);
    fp->compute();
    out << "\"threw_on_missing_inputs\":" << threw << ",\"pairs_projective\":" << pairsJson(corr);
    // the moving cloud changes IN PLACE (the tracker's clipped scene: same object, same size): the adapter must see the new contents
    PointNormal2fVectorCloud moved = moving;
    for (auto& p : moved) { p.coordinates().x() += 0.2f; }
    PointNormal2fVectorCloud work = moving;
    fp->setMoving(&work); fp->compute(); const size_t n_before = corr.size();
    work = moved;                                                                             // same object, same size, new contents
    fp->compute();
    auto fresh = std::make_shared<CorrespondenceFinderHIP2D>(); CorrespondenceVector corr2;
    fresh->param_projector.setValue(projector);
    fresh->setFixed(&fixed); fresh->setMoving(&moved); fresh->setLocalMapInSensor(geometry2d::v2t(x0)); fresh->setCorrespondences(&corr2);
    fresh->compute();
    bool same = corr.size() == corr2.size();
    for (size_t i = 0; same && i < corr.size(); ++i) same = corr[i].fixed_idx == corr2[i].fixed_idx && corr[i].moving_idx == corr2[i].moving_idx;
    out << ",\"in_place_change_seen\":" << (same ? 1 : 0) << ",\"pairs_before_change\":" << n_before << ",\"pairs_after_change\":" << corr.size();

    // the reference's own aligner loop around plugin interface #1 (apps/visual_test_aligner_2d.cpp:123-156 drives MultiAligner2D, which calls the slice's finder
    // once per iteration with a new local_map_in_sensor): twenty compute() calls, a moving cloud of map size that never changes -- ONE upload of it (round 5:
    // content check), the pairs those of a finder that uploads every time would give; and the siblings of one process share ONE device context
    {
      PointNormal2fVectorCloud big(100000);
      for (size_t i = 0; i < big.size(); ++i) {
        const auto& src = moving[i % moving.size()];
        const float shift = 1e-4f * (float) (i / moving.size());
        big[i].coordinates() = Vector2f(src.coordinates().x() + shift, src.coordinates().y()); big[i].normal() = src.normal();
      }
      auto loop = std::make_shared<CorrespondenceFinderHIP2D>(); CorrespondenceVector cl;
      loop->param_projector.setValue(projector);
      loop->setFixed(&fixed); loop->setMoving(&big); loop->setCorrespondences(&cl);
      loop->setLocalMapInSensor(geometry2d::v2t(x0)); loop->compute();
      const long long up0 = (long long) loop->contextUploads();
      size_t checksum = 0;
      for (int it = 0; it < 20; ++it) {
        const Vector3f xi(x0.x() + 0.002f * (float) it, x0.y() - 0.001f * (float) it, x0.z() + 0.0005f * (float) it);
        loop->setLocalMapInSensor(geometry2d::v2t(xi)); loop->compute();
        for (const auto& c : cl) checksum = checksum * 1000003u + (size_t) c.fixed_idx * 131u + (size_t) c.moving_idx;
      }
      const long long uploads_in_loop = (long long) loop->contextUploads() - up0;
      // the same twenty calls by a finder on a context of its own whose moving cloud is ANOTHER object every call (two copies taking turns: always uploaded)
      auto own = std::make_shared<lsm2d_srrg::HipContext>();
      auto every = std::make_shared<CorrespondenceFinderHIP2D>(); CorrespondenceVector ce;
      every->param_projector.setValue(projector); every->param_context.setValue(own);
      every->setFixed(&fixed); every->setCorrespondences(&ce);
      size_t checksum2 = 0; long long up1 = 0;
      PointNormal2fVectorCloud copies[2] = {big, big};
      for (int it = 0; it < 20; ++it) {
        const Vector3f xi(x0.x() + 0.002f * (float) it, x0.y() - 0.001f * (float) it, x0.z() + 0.0005f * (float) it);
        every->setMoving(&copies[it & 1]); every->setLocalMapInSensor(geometry2d::v2t(xi)); every->compute();
        if (it == 0) up1 = (long long) every->contextUploads();
        for (const auto& c : ce) checksum2 = checksum2 * 1000003u + (size_t) c.fixed_idx * 131u + (size_t) c.moving_idx;
      }
      out << ",\"aligner_loop_moving_uploads\":" << uploads_in_loop << ",\"aligner_loop_same_pairs\":" << (checksum == checksum2 ? 1 : 0)
          << ",\"aligner_loop_pairs_last\":" << cl.size() << ",\"every_call_uploads\":" << ((long long) every->contextUploads() - up1 + 1)
          << ",\"siblings_share_a_context\":" << (fp->contextUploads() == loop->contextUploads() ? 1 : 0)
          << ",\"own_context_is_separate\":" << (every->contextUploads() != loop->contextUploads() ? 1 : 0);
    }

    auto fk = std::make_shared<CorrespondenceFinderKDTreeHIP2D>(); CorrespondenceVector ck;
    fk->param_max_distance_m.setValue(0.3f);
    fk->param_search.setValue("exact");                                                       // the exact grid search
    fk->setFixed(&fixed); fk->setMoving(&moving); fk->setLocalMapInSensor(geometry2d::v2t(x0)); fk->setCorrespondences(&ck); fk->compute();
    // the sibling's default: the reference's own tree and descent, with the leaf parameters a KDTree2D configuration carries
    auto ft = std::make_shared<CorrespondenceFinderKDTreeHIP2D>(); CorrespondenceVector ct;
    ft->param_max_distance_m.setValue(0.3f); ft->param_max_leaf_range.setValue(0.05f); ft->param_min_leaf_points.setValue(12);
    ft->setFixed(&fixed); ft->setMoving(&moving); ft->setLocalMapInSensor(geometry2d::v2t(x0)); ft->setCorrespondences(&ct); ft->compute();
    int threw_search = 0;
    ft->param_search.setValue("octree");
    try { ft->compute(); } catch (const std::runtime_error&) { threw_search = 1; }
    out << ",\"n_kdtree_tree\":" << ct.size() << ",\"pairs_kdtree_tree\":" << pairsJson(ct) << ",\"threw_on_bad_search\":" << threw_search;
    auto fn = std::make_shared<CorrespondenceFinderNNHIP2D>(); CorrespondenceVector cn;
    fn->param_max_distance_m.setValue(0.5f); fn->param_resolution.setValue(0.1f);
    fn->setFixed(&fixed); fn->setMoving(&moving); fn->setLocalMapInSensor(geometry2d::v2t(x0)); fn->setCorrespondences(&cn); fn->compute();
    out << ",\"n_kdtree\":" << ck.size() << ",\"n_nn\":" << cn.size();
  }

  // ---- plugin interface #2: the aligner (visual_test_aligner_2d.cpp:102-156)
  auto makeSlice = [&](auto slice, float normal_cos, float tau) {
    slice->param_fixed_slice_name.setValue("points"); slice->param_moving_slice_name.setValue("points");
    auto finder = std::make_shared<CorrespondenceFinderProjective2f>();                        // the REFERENCE's finder object: parameters only
    finder->param_projector.setValue(projector); finder->param_normal_cos.setValue(normal_cos); finder->param_point_distance.setValue(0.5f);
    slice->param_finder.setValue(finder);
    slice->param_min_num_correspondences.setValue(10);
    if (tau > 0) { auto rb = std::make_shared<srrg2_solver::RobustifierCauchy>(); rb->param_chi_threshold.setValue(tau); slice->param_robustifier.setValue(rb); }
    return slice;
  };
  PropertyContainerDynamic fixed_scene, moving_scene;
  auto* prop_fixed = new Property_<PointNormal2fVectorCloud*>("points", "", &fixed_scene); prop_fixed->setValue(&fixed);
  auto* prop_moving = new Property_<PointNormal2fVectorCloud*>("points", "", &moving_scene); prop_moving->setValue(&moving);
  {
    auto aligner = std::make_shared<MultiAlignerHIP2D>();
    aligner->param_max_iterations.setValue(iters); aligner->param_min_num_inliers.setValue(10);
    auto slice = makeSlice(std::make_shared<AlignerSliceProcessorLaser2D>(), 0.8f, 0.f);
    aligner->param_slice_processors.pushBack(slice);
    aligner->param_publish_correspondences.setValue(1);
    aligner->setFixed(&fixed_scene); aligner->setMoving(&moving_scene); aligner->setMovingInFixed(geometry2d::v2t(x0));
    aligner->compute();
    const Vector3f est = geometry2d::t2v(aligner->movingInFixed());
    const Matrix3f& H = aligner->informationMatrix();
    out << ",\"pose\":" << f3(est) << ",\"status\":" << (int) aligner->status() << ",\"device_status\":" << aligner->lastDeviceStatus()
        << ",\"iterations\":" << aligner->iterationStats().size() << ",\"H00\":" << H(0, 0) << ",\"H22\":" << H(2, 2)
        << ",\"last_inliers\":" << aligner->iterationStats().back().num_inliers << ",\"slice_pairs\":" << slice->correspondences().size()
        << ",\"slice_fixed_bound\":" << (slice->fixed() == &fixed ? 1 : 0);
    // a second call on the same object: device clouds are reused, the result is the same
    aligner->setMovingInFixed(geometry2d::v2t(x0)); aligner->compute();
    out << ",\"pose_again\":" << f3(geometry2d::t2v(aligner->movingInFixed()));
    // NotEnoughInliers must come back as that status, not as a success with a moved pose
    aligner->param_min_num_inliers.setValue(1000000); aligner->setMovingInFixed(geometry2d::v2t(x0)); aligner->compute();
    out << ",\"status_not_enough_inliers\":" << (int) aligner->status();
    aligner->param_min_num_inliers.setValue(10);
    // the aligner's remaining options (MULTI.json:606-610,627-630): a termination_criteria object WITHOUT an epsilon cannot be translated and is
    // refused, one that carries the float property "epsilon" (as the solver's SimpleTerminationCriteria does, MULTI.json:218-223) is
    auto refuses = [&]() { try { aligner->setMovingInFixed(geometry2d::v2t(x0)); aligner->compute(); } catch (const std::runtime_error&) { return 1; } return 0; };
    aligner->param_termination_criteria.setValue(std::make_shared<AlignerTerminationCriteriaBase>());
    out << ",\"threw_on_opaque_termination_criteria\":" << refuses();
    struct ChiDecayCriteria : public AlignerTerminationCriteriaBase { PARAM(PropertyFloat, epsilon, "ratio of decay of chi2 between iteration", 1e-3f, 0); };
    aligner->param_termination_criteria.setValue(std::make_shared<ChiDecayCriteria>());
    out << ",\"refused_criteria_with_epsilon\":" << refuses();
    out << ",\"iterations_with_criteria_object\":" << aligner->iterationStats().size() << ",\"pose_with_criteria_object\":" << f3(geometry2d::t2v(aligner->movingInFixed()));
    aligner->param_termination_criteria.setValue(std::shared_ptr<AlignerTerminationCriteriaBase>());
    aligner->param_termination_chi_epsilon.setValue(-1.f);
    out << ",\"threw_on_negative_epsilon\":" << refuses();
    // ... the same criterion through the adapter's own PARAM: with an epsilon the loop stops early, at the same pose to the tolerance
    aligner->param_termination_chi_epsilon.setValue(1e-3f);
    out << ",\"refused_after_reset\":" << refuses();
    out << ",\"iterations_with_epsilon\":" << aligner->iterationStats().size() << ",\"pose_with_epsilon\":" << f3(geometry2d::t2v(aligner->movingInFixed()));
    aligner->param_termination_chi_epsilon.setValue(0.f);
  }
  {
    // enable_inlier_only_runs / keep_only_inlier_correspondences (MULTI.json:606-610) on a robustified slice: the second loop runs (up to 2 x
    // max_iterations statistics come back), and the slice is left with the last iteration's inliers only
    auto aligner = std::make_shared<MultiAlignerHIP2D>();
    aligner->param_max_iterations.setValue(iters); aligner->param_min_num_inliers.setValue(10);
    auto slice = makeSlice(std::make_shared<AlignerSliceProcessorLaser2D>(), 0.8f, 0.05f);
    aligner->param_slice_processors.pushBack(slice);
    aligner->param_publish_correspondences.setValue(1);
    aligner->setFixed(&fixed_scene); aligner->setMoving(&moving_scene);
    // a start off by 15 cm / 3 degrees: the first iterations see outliers
    const Vector3f x_off(x0.x() + 0.15f, x0.y() - 0.1f, x0.z() + 0.05f);
    aligner->setMovingInFixed(geometry2d::v2t(x_off)); aligner->compute();
    out << ",\"plain_iterations\":" << aligner->iterationStats().size() << ",\"plain_pairs\":" << slice->correspondences().size()
        << ",\"plain_last_inliers\":" << aligner->iterationStats().back().num_inliers << ",\"plain_last_outliers\":" << aligner->iterationStats().back().num_outliers
        << ",\"plain_pose\":" << f3(geometry2d::t2v(aligner->movingInFixed()));
    aligner->param_keep_only_inlier_correspondences.setValue(true);
    aligner->setMovingInFixed(geometry2d::v2t(x_off)); aligner->compute();
    out << ",\"keep_pairs\":" << slice->correspondences().size() << ",\"keep_last_inliers\":" << aligner->iterationStats().back().num_inliers
        << ",\"keep_pose\":" << f3(geometry2d::t2v(aligner->movingInFixed()));
    aligner->param_enable_inlier_only_runs.setValue(true);
    aligner->setMovingInFixed(geometry2d::v2t(x_off)); aligner->compute();
    out << ",\"inlier_runs_iterations\":" << aligner->iterationStats().size() << ",\"inlier_runs_status\":" << aligner->lastDeviceStatus()
        << ",\"inlier_runs_pairs\":" << slice->correspondences().size() << ",\"inlier_runs_last_inliers\":" << aligner->iterationStats().back().num_inliers
        << ",\"inlier_runs_pose\":" << f3(geometry2d::t2v(aligner->movingInFixed()));
  }
  {
    // the tracker's configuration (MULTI.json:715-721): laser slice with sensor extrinsics + Cauchy, the odometry prior, a second laser slice
    auto aligner = std::make_shared<MultiAlignerHIP2D>();
    aligner->param_max_iterations.setValue(iters);
    auto s0 = makeSlice(std::make_shared<AlignerSliceProcessorLaser2DWithSensor>(), 0.9f, 0.01f);
    s0->_sensor_in_robot = Isometry2f::Identity();
    auto odom = std::make_shared<AlignerSliceOdom2DPrior>();
    odom->param_fixed_slice_name.setValue("odom"); odom->param_moving_slice_name.setValue("odom");
    auto s1 = makeSlice(std::make_shared<AlignerSliceProcessorLaser2D>(), 0.8f, 0.f);
    aligner->param_slice_processors.pushBack(s0); aligner->param_slice_processors.pushBack(odom); aligner->param_slice_processors.pushBack(s1);
    // odometry says: the moving scene's origin is at x0 seen from the fixed one (slightly off the truth)
    auto* of = new Property_<Isometry2f>("odom", "", &fixed_scene); of->setValue(Isometry2f::Identity());
    auto* om = new Property_<Isometry2f>("odom", "", &moving_scene); om->setValue(geometry2d::v2t(x0));
    aligner->setFixed(&fixed_scene); aligner->setMoving(&moving_scene); aligner->setMovingInFixed(geometry2d::v2t(x0));
    aligner->compute();
    out << ",\"pose_multi\":" << f3(geometry2d::t2v(aligner->movingInFixed())) << ",\"status_multi\":" << (int) aligner->status();
    // an unknown slice processor is an error, never skipped
    struct Unknown : public AlignerSliceProcessorBase {};
    aligner->param_slice_processors.pushBack(std::make_shared<Unknown>());
    int threw = 0;
    try { aligner->compute(); } catch (const std::runtime_error&) { threw = 1; }
    out << ",\"threw_on_unknown_slice\":" << threw;
  }

  // ---- mapping siblings (row f1)
  {
    PointNormal2fVectorCloud clipped;
    SceneClipperHIP2D clipper; clipper.param_projector.setValue(projector);
    clipper.param_voxelize_resolution.setValue(0.f);      // as both shipped configurations set it (the class default is 0.1, as the reference's)
    clipper.setFullScene(&moving); clipper.setClippedSceneInRobot(&clipped);
    clipper.setRobotInLocalMap(geometry2d::v2t(x0).inverse()); clipper.setSensorInRobot(Isometry2f::Identity());
    clipper.compute();
    const size_t n_plain = clipped.size();
    clipper.param_voxelize_resolution.setValue(0.2f); clipper.compute();
    out << ",\"clipped\":" << n_plain << ",\"clipped_voxelized\":" << clipped.size() << ",\"clip_status\":" << (int) clipper.status();
    PointNormal2fVectorCloud scene = moving;
    MergerHIP2D merger; merger.param_projector.setValue(projector);
    merger.setScene(&scene); merger.setMeasurement(&fixed); merger.setMeasurementInScene(geometry2d::v2t(x0).inverse());
    merger.compute();
    out << ",\"merged_size\":" << scene.size() << ",\"merge_status\":" << (int) merger.status();
  }

  // ---- raw-data preprocessor sibling (row f2), driven as the pipeline drives the reference module: setRawData(message), compute()
  if (argc >= 12) {
    std::ifstream rf(argv[8], std::ios::binary);
    rf.seekg(0, std::ios::end); const size_t bytes = (size_t) rf.tellg(); rf.seekg(0);
    auto msg = std::make_shared<LaserMessage>();
    msg->ranges.value().resize(bytes / 4); rf.read((char*) msg->ranges.value().data(), (std::streamsize) bytes);
    msg->angle_min.setValue((float) atof(argv[9])); msg->angle_max.setValue((float) atof(argv[10]));
    msg->range_min.setValue(0.0f); msg->range_max.setValue(30.0f); msg->topic.setValue("/scan_front");
    RawDataPreprocessorHIP2D pre;
    pre.param_range_min.setValue(0.3f); pre.param_range_max.setValue(20.0f); pre.param_scan_topic.setValue("/scan_front");
    PointNormal2fVectorCloud meas;
    pre.compute();                                              // nothing set: status Error, no throw (.cpp:13-17)
    const int status_unset = (int) pre.status();
    pre.setMeas(&meas);
    auto other = std::make_shared<LaserMessage>(); other->topic.setValue("/scan_rear");
    const bool took_other = pre.setRawData(other);              // not this module's topic (.cpp:64-69)
    int threw_null = 0;
    try { pre.setRawData(nullptr); } catch (const std::runtime_error&) { threw_null = 1; }
    const bool took = pre.setRawData(msg);
    pre.compute();
    std::ofstream of(argv[11], std::ios::binary);
    for (const auto& p : meas) { const float v[4] = {p.coordinates().x(), p.coordinates().y(), p.normal().x(), p.normal().y()}; of.write((const char*) v, sizeof v); }
    out << ",\"prep_status_unset\":" << status_unset << ",\"prep_took_other_topic\":" << (int) took_other << ",\"prep_threw_on_null\":" << threw_null
        << ",\"prep_took\":" << (int) took << ",\"prep_status\":" << (int) pre.status() << ",\"prep_points\":" << meas.size()
        << ",\"unprojector_range_max\":" << pre.param_unprojector->param_range_max.value() << ",\"unprojector_range_min\":" << pre.param_unprojector->param_range_min.value();
  }
  out << "}";
  printf("%s\n", out.str().c_str());
  return 0;
}
