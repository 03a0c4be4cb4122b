// lsm2d.hpp -- header-only C++ host mirror of the reference's finder / aligner surface over the C ABI
// (include/lsm2d.h).  Same method names, argument meaning and error behaviour as the reference classes:
//   CorrespondenceFinderProjective2f   registration/correspondence_finder_projective_2d.{h,cpp}
//       setFixed / setMoving / setLocalMapInSensor / setCorrespondences / compute
//       (apps/visual_test_correspondence_finder_projective_2d.cpp:74-79); throws std::runtime_error on a
//       missing projector / fixed / moving exactly where the reference does (.cpp:21-31)
//   MultiAligner2D                     upstream, driven as in apps/visual_test_aligner_2d.cpp:123-156
//       param_slice_processors / setFixed / setMoving / setMovingInFixed / compute / movingInFixed /
//       iterationStats / status
//   LaserMessageBatchStream            batches of fresh LaserMessages through preprocessor + aligner, pipelined over three scan sets
//       (lsm2d_preprocess_scans_refill, lsm2d_align_batch_begin / _wait)
// This header depends on nothing but the C ABI and the standard library, so it compiles with plain g++;
// the SRRG-side adapter (adapters/srrg/) is the same code expressed with srrg2_core types.
#pragma once
#include <lsm2d.h>

#include <array>
#include <cmath>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace lsm2d_host {

struct PointNormal2f { float x, y, nx, ny; };                 // srrg2_core::PointNormal2f payload
using PointNormal2fVectorCloud = std::vector<PointNormal2f>;
using Correspondence = lsm2d_correspondence;                  // {fixed_idx, moving_idx}
using CorrespondenceVector = std::vector<Correspondence>;
using Vector3f = std::array<float, 3>;                        // geometry2d::t2v(Isometry2f) = (x, y, theta)

inline void check(int rc, const char* where, const lsm2d_context* ctx = nullptr) {
  if (rc < 0) throw std::runtime_error(std::string(where) + ": " + lsm2d_status_string(rc) + " " + lsm2d_last_error(ctx));
}

class Context {
 public:
  explicit Context(int device = 0, void* hip_stream = nullptr) { check(lsm2d_create(device, hip_stream, &_h), "lsm2d_create"); }
  ~Context() { lsm2d_destroy(_h); }
  Context(const Context&) = delete; Context& operator=(const Context&) = delete;
  lsm2d_context* get() const { return _h; }
  void setKernelTiming(bool on) { check(lsm2d_set_option(_h, "kernel_timing", on ? 1 : 0), "lsm2d_set_option", _h); }   // needed before lastKernelMs()
  // "align_path": 0 automatic, 1 k_align, 2 split over many workgroups, 3 two projective slices side by side (include/lsm2d.h)
  void setOption(const char* key, int64_t value) { check(lsm2d_set_option(_h, key, value), "lsm2d_set_option", _h); }
  int64_t getOption(const char* key) const { int64_t v = 0; check(lsm2d_get_option(_h, key, &v), "lsm2d_get_option", _h); return v; }
  float lastKernelMs() const { float ms = 0; check(lsm2d_last_kernel_ms(_h, &ms), "lsm2d_last_kernel_ms", _h); return ms; }
 private:
  lsm2d_context* _h = nullptr;
};

// device-resident copy of one cloud or of a ragged batch of clouds
class CloudSet {
 public:
  CloudSet(Context& ctx, const PointNormal2fVectorCloud& cloud) : _ctx(&ctx) {
    check(lsm2d_cloudset_create(ctx.get(), cloud.empty() ? nullptr : &cloud[0].x, nullptr, 1, (int64_t) cloud.size(), &_h), "lsm2d_cloudset_create", ctx.get());
  }
  CloudSet(Context& ctx, const std::vector<PointNormal2fVectorCloud>& clouds) : _ctx(&ctx) {
    std::vector<int32_t> offs(1, 0); PointNormal2fVectorCloud packed;
    for (const auto& c : clouds) { packed.insert(packed.end(), c.begin(), c.end()); offs.push_back((int32_t) packed.size()); }
    check(lsm2d_cloudset_create(ctx.get(), packed.empty() ? nullptr : &packed[0].x, offs.data(), (int32_t) clouds.size(), (int64_t) packed.size(), &_h),
          "lsm2d_cloudset_create", ctx.get());
  }
  ~CloudSet() { lsm2d_cloudset_destroy(_h); }
  CloudSet(const CloudSet&) = delete; CloudSet& operator=(const CloudSet&) = delete;
  lsm2d_cloudset* get() const { return _h; }
  int32_t numClouds() const { return lsm2d_cloudset_num_clouds(_h); }
 private:
  Context* _ctx; lsm2d_cloudset* _h = nullptr;
};

// PointNormal2fProjectorPolar parameters (apps/synthetic_scene_generator.cpp:69-75, MULTI.json:71-97)
struct PointNormal2fProjectorPolar {
  int   param_canvas_cols = 721;
  float param_angle_col_min = -3.14159f, param_angle_col_max = 3.14159f;
  float param_range_min = 0.3f, param_range_max = 20.f;
  float col_offset = 0.f;
  lsm2d_projector abi() const { return {param_canvas_cols, param_angle_col_min, param_angle_col_max, param_range_min, param_range_max, col_offset}; }
};
using PointNormal2fProjectorPolarPtr = std::shared_ptr<PointNormal2fProjectorPolar>;

class CorrespondenceFinderProjective2f {
 public:
  explicit CorrespondenceFinderProjective2f(Context& ctx) : _ctx(ctx) {}
  float param_point_distance = 0.5f;   // .h:16-20
  float param_normal_cos = 0.8f;       // .h:21
  PointNormal2fProjectorPolarPtr param_projector{new PointNormal2fProjectorPolar};

  // the clouds are uploaded when set (the reference keeps raw pointers and a _fixed_changed_flag)
  void setFixed(const PointNormal2fVectorCloud* fixed) { _fixed.reset(fixed ? new CloudSet(_ctx, *fixed) : nullptr); }
  void setMoving(const PointNormal2fVectorCloud* moving) { _moving.reset(moving ? new CloudSet(_ctx, *moving) : nullptr); }
  void setLocalMapInSensor(const Vector3f& pose) { _local_map_in_sensor = pose; }
  void setCorrespondences(CorrespondenceVector* c) { _correspondences = c; }

  lsm2d_slice_params sliceParams() const {
    lsm2d_slice_params sp{};
    sp.finder = LSM2D_FINDER_PROJECTIVE; sp.projector = param_projector->abi();
    sp.point_distance = param_point_distance; sp.normal_cos = param_normal_cos;
    return sp;
  }

  void compute() {
    if (!param_projector) throw std::runtime_error("CorrespondenceFinderProjective2f::compute| Missing Projector");
    if (!_fixed) throw std::runtime_error("CorrespondenceFinderProjective2f::compute| Missing fixed!");
    if (!_moving) throw std::runtime_error("CorrespondenceFinderProjective2f::compute| Missing moving!");
    if (!_correspondences) throw std::runtime_error("CorrespondenceFinderProjective2f::compute| Missing correspondences!");
    const lsm2d_slice_params sp = sliceParams();
    _correspondences->resize((size_t) sp.projector.canvas_cols);        // .cpp:49
    int32_t k = 0;
    check(lsm2d_find_correspondences(_ctx.get(), &sp, _fixed->get(), 0, _moving->get(), 0, _local_map_in_sensor.data(),
                                     _correspondences->data(), (int32_t) _correspondences->size(), &k),
          "lsm2d_find_correspondences", _ctx.get());
    _correspondences->resize((size_t) k);                                 // .cpp:76
  }
 private:
  Context& _ctx;
  std::unique_ptr<CloudSet> _fixed, _moving;
  Vector3f _local_map_in_sensor{{0.f, 0.f, 0.f}};
  CorrespondenceVector* _correspondences = nullptr;
};

// CorrespondenceFinderKDTree2D (registration/correspondence_finder_kd_tree_2d.{h,cpp}) and CorrespondenceFinderNN2D
// (registration/correspondence_finder_nn_2d.{h,cpp}): same surface, different lsm2d_finder.
class CorrespondenceFinderPointQuery {
 public:
  CorrespondenceFinderPointQuery(Context& ctx, int finder) : _ctx(ctx), _finder(finder) {}
  float param_max_distance_m = 1e-2f;   // kd .h:23 (nn .h:20-24 defaults to 1)
  float param_resolution = 5e-2f;       // nn .h:25-29
  float param_normal_cos = 0.8f;
  float param_max_leaf_range = 1e-2f;   // kd .h:26-28 -- honoured by LSM2D_FINDER_KDTREE (search "kdtree")
  unsigned param_min_leaf_points = 20;  // kd .h:29-33
  void setFixed(const PointNormal2fVectorCloud* fixed) { _fixed.reset(fixed ? new CloudSet(_ctx, *fixed) : nullptr); }
  void setMoving(const PointNormal2fVectorCloud* moving) { _moving.reset(moving ? new CloudSet(_ctx, *moving) : nullptr); _n_moving = moving ? moving->size() : 0; }
  void setLocalMapInSensor(const Vector3f& pose) { _local_map_in_sensor = pose; }
  void setCorrespondences(CorrespondenceVector* c) { _correspondences = c; }
  lsm2d_slice_params sliceParams() const {
    lsm2d_slice_params sp{};
    sp.finder = _finder; sp.projector = PointNormal2fProjectorPolar().abi();
    sp.max_distance = param_max_distance_m; sp.resolution = param_resolution; sp.normal_cos = param_normal_cos;
    sp.kd_max_leaf_range = param_max_leaf_range; sp.kd_min_leaf_points = (int32_t) param_min_leaf_points;
    return sp;
  }
  void compute() {
    if (_finder == LSM2D_FINDER_DISTMAP && !(param_resolution > 0.f)) throw std::runtime_error("resolution must be > 0");      // nn .cpp:11-14
    if (!_fixed || !_moving || !_correspondences) throw std::runtime_error("CorrespondenceFinder::compute| missing fixed, moving or correspondences");
    const lsm2d_slice_params sp = sliceParams();
    _correspondences->resize(_n_moving);                                   // kd .cpp:10, nn .cpp:60
    int32_t k = 0;
    check(lsm2d_find_correspondences(_ctx.get(), &sp, _fixed->get(), 0, _moving->get(), 0, _local_map_in_sensor.data(),
                                     _correspondences->data(), (int32_t) _correspondences->size(), &k),
          "lsm2d_find_correspondences", _ctx.get());
    _correspondences->resize((size_t) k);
  }
 protected:
  Context& _ctx; int _finder;
 private:
  std::unique_ptr<CloudSet> _fixed, _moving; size_t _n_moving = 0;
  Vector3f _local_map_in_sensor{{0.f, 0.f, 0.f}};
  CorrespondenceVector* _correspondences = nullptr;
};
// search "kdtree" (default): the reference's own tree and single-leaf descent (approximate, honours max_leaf_range / min_leaf_points);
// "exact": an exact nearest-neighbour search on a uniform grid (LSM2D_FINDER_NN), for which the two leaf parameters have no meaning
struct CorrespondenceFinderKDTree2D : CorrespondenceFinderPointQuery {
  explicit CorrespondenceFinderKDTree2D(Context& ctx, const std::string& search = "kdtree") : CorrespondenceFinderPointQuery(ctx, finderOf(search)) {}
  void setSearch(const std::string& search) { _finder = finderOf(search); }
  static int finderOf(const std::string& search) {
    if (search == "kdtree") return LSM2D_FINDER_KDTREE;
    if (search == "exact") return LSM2D_FINDER_NN;
    throw std::runtime_error("CorrespondenceFinderKDTree2D| search must be \"kdtree\" or \"exact\"");
  }
};
struct CorrespondenceFinderNN2D : CorrespondenceFinderPointQuery {
  explicit CorrespondenceFinderNN2D(Context& ctx) : CorrespondenceFinderPointQuery(ctx, LSM2D_FINDER_DISTMAP) { param_max_distance_m = 1.f; }
};

// a single growable device cloud: the tracker's local map / the clipped scene
class ReservedCloud {
 public:
  ReservedCloud(Context& ctx, int64_t capacity) : _ctx(ctx) { check(lsm2d_cloudset_create_reserved(ctx.get(), capacity, &_h), "lsm2d_cloudset_create_reserved", ctx.get()); }
  ~ReservedCloud() { lsm2d_cloudset_destroy(_h); }
  ReservedCloud(const ReservedCloud&) = delete; ReservedCloud& operator=(const ReservedCloud&) = delete;
  lsm2d_cloudset* get() const { return _h; }
  void upload(const PointNormal2fVectorCloud& c) { check(lsm2d_cloudset_upload(_h, c.empty() ? nullptr : &c[0].x, (int64_t) c.size()), "lsm2d_cloudset_upload", _ctx.get()); }
  PointNormal2fVectorCloud download() const {
    PointNormal2fVectorCloud c((size_t) lsm2d_cloudset_num_points(_h)); int64_t n = 0;
    check(lsm2d_cloudset_download(_h, 0, c.empty() ? nullptr : &c[0].x, (int64_t) c.size(), &n), "lsm2d_cloudset_download", _ctx.get());
    c.resize((size_t) n); return c;
  }
  int64_t size() const { return lsm2d_cloudset_num_points(_h); }
 private:
  Context& _ctx; lsm2d_cloudset* _h = nullptr;
};

// SceneClipperProjective2D (mapping/scene_clipper_projective_2d.{h,cpp})
class SceneClipperProjective2D {
 public:
  explicit SceneClipperProjective2D(Context& ctx) : _ctx(ctx) {}
  PointNormal2fProjectorPolarPtr param_projector{new PointNormal2fProjectorPolar};
  void setFullScene(const ReservedCloud* scene) { _scene = scene; }
  void setClippedSceneInRobot(ReservedCloud* clipped) { _clipped = clipped; }
  void setRobotInLocalMap(const Vector3f& p) { _robot_in_local_map = p; }
  void setSensorInRobot(const Vector3f& p) { _sensor_in_robot = p; }
  int compute() {
    if (!_scene || !_clipped) throw std::runtime_error("SceneClipperProjective2D::compute| missing local OR global scene");     // .cpp:12-17
    if (!param_projector) throw std::runtime_error("SceneClipperProjective2D::compute| Missing Projector");                     // .cpp:19-21
    const lsm2d_projector pr = param_projector->abi(); int32_t n = -1;
    // asynchronous: the call only queues the work; the clipped set's size stays on the device until somebody asks (returns -1)
    check(lsm2d_clip_scene_voxelized(_ctx.get(), &pr, _scene->get(), 0, _robot_in_local_map.data(), _sensor_in_robot.data(),
                                     param_voxelize_resolution, _clipped->get(), asynchronous ? nullptr : &n, nullptr),
          "lsm2d_clip_scene", _ctx.get());
    return n;
  }
  bool asynchronous = false;
  float param_voxelize_resolution = 0.1f;                                  // the class default (.h:21-25); both shipped configurations set 0 (MULTI.json:673-683); > 0: .cpp:36-48
 private:
  Context& _ctx; const ReservedCloud* _scene = nullptr; ReservedCloud* _clipped = nullptr;
  Vector3f _robot_in_local_map{{0.f, 0.f, 0.f}}, _sensor_in_robot{{0.f, 0.f, 0.f}};
};

// MergerProjective2D (mapping/merger_projective_2d.{h,cpp})
class MergerProjective2D {
 public:
  explicit MergerProjective2D(Context& ctx) : _ctx(ctx) {}
  float param_merge_threshold = 0.2f;                                      // .h:12-16
  PointNormal2fProjectorPolarPtr param_projector{new PointNormal2fProjectorPolar};
  void setScene(ReservedCloud* scene) { _scene = scene; }
  void setMeasurement(const PointNormal2fVectorCloud* m) { _measurement.reset(m ? new CloudSet(_ctx, *m) : nullptr); _device_measurement = nullptr; }
  void setMeasurement(const ReservedCloud* m) { _measurement.reset(); _device_measurement = m; }     // a measurement that already lives on the device
  void setMeasurementInScene(const Vector3f& p) { _measurement_in_scene = p; }
  int compute() {
    if (!param_projector) throw std::runtime_error("MergerProjective2D::compute| Missing Projector");                            // .cpp:10-12
    const lsm2d_cloudset* meas = _device_measurement ? _device_measurement->get() : (_measurement ? _measurement->get() : nullptr);
    if (!_scene || !meas) throw std::runtime_error("MergerProjective2D::compute| missing scene or measurement");
    const lsm2d_projector pr = param_projector->abi(); int32_t size = -1;
    // asynchronous: queue only; the scene's new size (and the counts) stay on the device (returns -1)
    check(lsm2d_merge_scene(_ctx.get(), &pr, _scene->get(), meas, 0, _measurement_in_scene.data(), param_merge_threshold,
                            asynchronous ? nullptr : &size, asynchronous ? nullptr : counts.data()),
          "lsm2d_merge_scene", _ctx.get());
    return size;
  }
  // several device-resident measurements, each at its own pose, in order: one call (one launch when everything is small)
  int computeAll(const std::vector<const ReservedCloud*>& measurements, const std::vector<Vector3f>& poses) {
    if (!param_projector) throw std::runtime_error("MergerProjective2D::compute| Missing Projector");
    if (!_scene || measurements.empty() || measurements.size() != poses.size()) throw std::runtime_error("MergerProjective2D::compute| missing scene or measurement");
    std::vector<const lsm2d_cloudset*> sets; std::vector<float> p;
    for (size_t k = 0; k < measurements.size(); ++k) { sets.push_back(measurements[k]->get()); p.insert(p.end(), poses[k].begin(), poses[k].end()); }
    const lsm2d_projector pr = param_projector->abi(); int32_t size = -1;
    check(lsm2d_merge_scenes(_ctx.get(), &pr, _scene->get(), (int32_t) sets.size(), sets.data(), nullptr, p.data(), param_merge_threshold,
                             asynchronous ? nullptr : &size, nullptr), "lsm2d_merge_scenes", _ctx.get());
    return size;
  }
  bool asynchronous = false;
  std::array<int32_t, 3> counts{{0, 0, 0}};                                // new, merged, replaced
 private:
  Context& _ctx; ReservedCloud* _scene = nullptr; std::unique_ptr<CloudSet> _measurement; const ReservedCloud* _device_measurement = nullptr;
  Vector3f _measurement_in_scene{{0.f, 0.f, 0.f}};
};

struct RobustifierCauchy { float param_chi_threshold = 0.01f; };     // MULTI.json:153-158

// AlignerSliceProcessorLaser2D[WithSensor] (registration/aligner_slice_processor_laser_2d.h:7-42; MULTI.json:160-188)
struct AlignerSliceProcessorLaser2D {
  std::string param_fixed_slice_name = "points", param_moving_slice_name = "points";
  int param_min_num_correspondences = 0;
  std::shared_ptr<CorrespondenceFinderProjective2f> param_finder;
  std::shared_ptr<RobustifierCauchy> param_robustifier;
  Vector3f sensor_in_robot{{0.f, 0.f, 0.f}};                           // WithSensor: from the tf Platform
  lsm2d_slice_params sliceParams() const {
    if (!param_finder) throw std::runtime_error("AlignerSliceProcessorLaser2D| Missing finder");
    lsm2d_slice_params sp = param_finder->sliceParams();
    sp.robustifier = param_robustifier ? LSM2D_ROBUST_CAUCHY : LSM2D_ROBUST_NONE;
    sp.chi_threshold = param_robustifier ? param_robustifier->param_chi_threshold : 0.f;
    sp.min_num_correspondences = param_min_num_correspondences;
    for (int i = 0; i < 3; ++i) sp.sensor_in_robot[i] = sensor_in_robot[i];
    return sp;
  }
};
using AlignerSliceProcessorLaser2DPtr = std::shared_ptr<AlignerSliceProcessorLaser2D>;

class MultiAligner2D {
 public:
  enum Status { Success = 0, NotEnoughCorrespondences = 1, NotEnoughInliers = 2, Fail = 3 };
  using PropertyContainer = std::map<std::string, const PointNormal2fVectorCloud*>;   // slice name -> cloud
  explicit MultiAligner2D(Context& ctx) : _ctx(ctx) {}
  int param_max_iterations = 10, param_min_num_inliers = 10;             // MULTI.json:711,714
  float param_damping = 0.f;                                             // MULTI.json:254-259
  // the options the shipped aligners carry at their defaults (MULTI.json:606-610,627-630); semantics in include/lsm2d.h (restated from the
  // parameters' doc strings: the upstream class is not in the reference tree); the termination criterion exists as an epsilon (lsm2d.h)
  bool param_enable_inlier_only_runs = false, param_keep_only_inlier_correspondences = false;
  bool store_correspondences = false;      // compute() also fetches what the reference leaves in slice->correspondences() (a finder pass per slice)
  float param_termination_chi_epsilon = 0.f;                             // 0 = "termination_criteria" not set = max_iterations
  std::vector<AlignerSliceProcessorLaser2DPtr> param_slice_processors;

  void setFixed(const PropertyContainer* f) { _fixed = f; }
  void setMoving(const PropertyContainer* m) { _moving = m; }
  void setMovingInFixed(const Vector3f& x) { _moving_in_fixed = x; }
  void setPrior(const Vector3f& z, const std::array<float, 9>& omega) { _has_prior = true; for (int i = 0; i < 3; ++i) _prior.z[i] = z[i]; for (int i = 0; i < 9; ++i) _prior.omega[i] = omega[i]; }

  void compute() {
    if (!_fixed || !_moving) throw std::runtime_error("MultiAligner2D::compute| fixed or moving not set");
    const int ns = (int) param_slice_processors.size();
    if (ns < 1) throw std::runtime_error("MultiAligner2D::compute| no slice processors");
    std::vector<lsm2d_slice_params> sp; std::vector<std::unique_ptr<CloudSet>> own;
    std::vector<const lsm2d_cloudset*> fx, mv;
    for (const auto& s : param_slice_processors) {
      sp.push_back(s->sliceParams());
      auto fi = _fixed->find(s->param_fixed_slice_name); auto mi = _moving->find(s->param_moving_slice_name);
      if (fi == _fixed->end() || mi == _moving->end() || !fi->second || !mi->second) throw std::runtime_error("MultiAligner2D::compute| slice cloud missing");
      own.emplace_back(new CloudSet(_ctx, *fi->second)); fx.push_back(own.back()->get());
      own.emplace_back(new CloudSet(_ctx, *mi->second)); mv.push_back(own.back()->get());
    }
    lsm2d_batch b{}; b.n_alignments = 1; b.n_slices = ns; b.slices = sp.data(); b.fixed = fx.data(); b.moving = mv.data();
    b.init_pose = _moving_in_fixed.data(); b.prior = _has_prior ? &_prior : nullptr;
    lsm2d_aligner_params ap{param_max_iterations, param_min_num_inliers, param_damping, param_termination_chi_epsilon,
                            param_enable_inlier_only_runs ? 1 : 0, param_keep_only_inlier_correspondences ? 1 : 0};
    _stats.assign((size_t) lsm2d_stats_capacity(&ap), lsm2d_iteration_stats{});
    int32_t st = 0, its = 0;
    _pairs.assign((size_t) ns, {});
    if (store_correspondences) {
      size_t cap = 1;
      for (int s = 0; s < ns; ++s) {
        const size_t need = sp[(size_t) s].finder == LSM2D_FINDER_PROJECTIVE ? (size_t) sp[(size_t) s].projector.canvas_cols : (size_t) lsm2d_cloudset_num_points(mv[(size_t) s]);
        cap = need > cap ? need : cap;
      }
      std::vector<lsm2d_correspondence> buf(cap * (size_t) ns); std::vector<int32_t> cnt((size_t) ns, 0);
      check(lsm2d_align_batch_pairs(_ctx.get(), &ap, &b, _moving_in_fixed.data(), _information.data(), &st, &its, _stats.data(), buf.data(), (int32_t) cap, cnt.data()),
            "lsm2d_align_batch_pairs", _ctx.get());
      for (int s = 0; s < ns; ++s) _pairs[(size_t) s].assign(buf.begin() + (long) (cap * (size_t) s), buf.begin() + (long) (cap * (size_t) s) + cnt[(size_t) s]);
    } else {
      check(lsm2d_align_batch(_ctx.get(), &ap, &b, _moving_in_fixed.data(), _information.data(), &st, &its, _stats.data()), "lsm2d_align_batch", _ctx.get());
    }
    _stats.resize((size_t) its); _status = st;
  }
  // slice->correspondences() after compute() (apps/visual_test_aligner_2d.cpp:129-143); needs store_correspondences
  const std::vector<lsm2d_correspondence>& correspondences(size_t slice = 0) const { return _pairs.at(slice); }
  const Vector3f& movingInFixed() const { return _moving_in_fixed; }
  const std::array<float, 9>& informationMatrix() const { return _information; }
  const std::vector<lsm2d_iteration_stats>& iterationStats() const { return _stats; }
  int status() const { return _status; }
 private:
  Context& _ctx;
  const PropertyContainer* _fixed = nullptr; const PropertyContainer* _moving = nullptr;
  Vector3f _moving_in_fixed{{0.f, 0.f, 0.f}};
  std::array<float, 9> _information{};
  std::vector<lsm2d_iteration_stats> _stats;
  std::vector<std::vector<lsm2d_correspondence>> _pairs;
  lsm2d_prior _prior{}; bool _has_prior = false; int _status = 0;
};

// Batches of FRESH LaserMessages against one device-resident map, pipelined: per incoming message RawDataPreprocessorProjective2D::compute
// (sensor_processing/raw_data_preprocessor_projective_2d.cpp:13-51) feeding MultiAligner2D::compute (apps/visual_test_aligner_2d.cpp:123-156), for n_scans
// messages per push().  THREE scan sets in rotation, the order include/lsm2d.h recommends at lsm2d_align_batch_begin:  push(k) begins batch k - 1 (whose scans
// were preprocessed by the previous push), refills a set with the ranges of batch k (lsm2d_preprocess_scans_refill: their preprocessing has the whole launch of
// batch k - 1 to hide under), and waits for batch k - 2.  It returns true when pose / information / status / iterations hold a finished batch's results --
// retired() says which push they belong to -- bit for bit those of lsm2d_preprocess_scans + lsm2d_align_batch on the same ranges.
// At the end of the sequence:  while (stream.flush()) { ...results of one more batch... }.
// The caller keeps a pushed `ranges` buffer untouched until the results of ITS batch have come out (pinned memory is fetched asynchronously).
class LaserMessageBatchStream {
 public:
  LaserMessageBatchStream(Context& ctx, const lsm2d_preprocessor& pre, const CloudSet& map, const lsm2d_slice_params& slice, const lsm2d_aligner_params& aligner, int n_scans)
      : _ctx(ctx), _pre(pre), _map(map), _slice(slice), _aligner(aligner), _n(n_scans) {
    if (n_scans < 1) throw std::runtime_error("LaserMessageBatchStream| n_scans < 1");
    pose.resize((size_t) n_scans); information.resize((size_t) n_scans); status.resize((size_t) n_scans); iterations.resize((size_t) n_scans);
    _start.resize((size_t) n_scans);
  }
  ~LaserMessageBatchStream() {
    try { while (flush()) {} } catch (...) {}
    for (auto* s : _sets) if (s) lsm2d_cloudset_destroy(s);
  }
  LaserMessageBatchStream(const LaserMessageBatchStream&) = delete; LaserMessageBatchStream& operator=(const LaserMessageBatchStream&) = delete;
  bool push(const float* ranges /* [n_scans][n_beams] */, const Vector3f* init_pose /* [n_scans] */) {
    if (_pushed > _begun) begin_next();                                   // batch k - 1: its scans are on the device (or on their way)
    if (!_sets[0])                                                        // the first push makes the three sets (allocation; nothing is in flight yet), every later one refills
      for (auto& s : _sets) check(lsm2d_preprocess_scans(_ctx.get(), &_pre, ranges, _n, &s), "lsm2d_preprocess_scans", _ctx.get());
    else                                                                  // (the set batch k - 3 read: retired by the previous push)
      check(lsm2d_preprocess_scans_refill(_ctx.get(), &_pre, ranges, _n, _sets[_pushed % 3]), "lsm2d_preprocess_scans_refill", _ctx.get());
    _start.assign(init_pose, init_pose + _n);
    ++_pushed;
    return _begun - _retired >= 2 ? retire() : false;                     // batch k - 2
  }
  bool flush() {                      // ONE more batch per call, oldest first; false when none is left
    if (_pushed > _begun) begin_next();
    return _begun > _retired ? retire() : false;
  }
  long retired() const { return _retired - 1; }                           // the push (0-based) whose results the members hold; -1: none yet
  std::vector<Vector3f> pose; std::vector<std::array<float, 9>> information; std::vector<int32_t> status, iterations;
 private:
  void begin_next() {
    const lsm2d_cloudset* fx[1] = {_sets[_begun % 3]}; const lsm2d_cloudset* mv[1] = {_map.get()};
    lsm2d_batch b{}; b.n_alignments = _n; b.n_slices = 1; b.slices = &_slice; b.fixed = fx; b.moving = mv; b.init_pose = _start[0].data();
    check(lsm2d_align_batch_begin(_ctx.get(), &_aligner, &b, 0, &_pending[_begun & 1]), "lsm2d_align_batch_begin", _ctx.get());
    ++_begun;
  }
  bool retire() {
    lsm2d_pending* p = _pending[_retired & 1]; _pending[_retired & 1] = nullptr;     // wait() consumes the handle whatever it returns
    ++_retired;
    check(lsm2d_align_batch_wait(p, pose[0].data(), information[0].data(), status.data(), iterations.data(), nullptr), "lsm2d_align_batch_wait", _ctx.get());
    return true;
  }
  Context& _ctx; lsm2d_preprocessor _pre; const CloudSet& _map; lsm2d_slice_params _slice; lsm2d_aligner_params _aligner; int _n;
  lsm2d_cloudset* _sets[3] = {nullptr, nullptr, nullptr}; lsm2d_pending* _pending[2] = {nullptr, nullptr};
  std::vector<Vector3f> _start;                                             // the start poses of the batch pushed last (begun by the next push)
  long _pushed = 0, _begun = 0, _retired = 0;
};

// The candidate loop of MultiLoopDetectorBruteForce2D / MultiRelocalizer2D (MULTI.json:964-986, :749-769) over the GPUs of one node,
// in this process: one context and one host thread per device, the submap replicated device to device, candidates block-sharded,
// results in candidate order (lsm2d_sweep_* in include/lsm2d.h).  accept() is the reference's acceptance test (MULTI.json:979-985).
class LoopClosureSweep {
 public:
  explicit LoopClosureSweep(const std::vector<int>& device_ids) {
    std::vector<int32_t> ids(device_ids.begin(), device_ids.end());
    const int rc = lsm2d_sweep_create(ids.data(), (int32_t) ids.size(), &_sw);
    if (rc < 0) throw std::runtime_error(std::string("LoopClosureSweep| ") + lsm2d_status_string(rc) + ": " + lsm2d_sweep_last_error(nullptr));
  }
  ~LoopClosureSweep() { lsm2d_sweep_destroy(_sw); }
  LoopClosureSweep(const LoopClosureSweep&) = delete;
  LoopClosureSweep& operator=(const LoopClosureSweep&) = delete;
  int numDevices() const { return lsm2d_sweep_num_devices(_sw); }
  // relocalize_aligner's parameters (MULTI.json:602-630) and its one laser slice (MULTI.json:572-600,771-784: projective finder,
  // point_distance 1.414, normal_cos 0.8, Cauchy 0.05, min_num_correspondences 10)
  int param_max_iterations = 30, param_min_num_inliers = 10;
  lsm2d_slice_params param_slice = projectiveSlice(PointNormal2fProjectorPolar(), 1.414f, 0.8f, 0.05f, 10);
  static lsm2d_slice_params projectiveSlice(const PointNormal2fProjectorPolar& projector, float point_distance, float normal_cos,
                                            float cauchy_chi_threshold /* <= 0: no robustifier */, int min_num_correspondences) {
    lsm2d_slice_params sp{};
    sp.finder = LSM2D_FINDER_PROJECTIVE; sp.projector = projector.abi(); sp.point_distance = point_distance; sp.normal_cos = normal_cos;
    sp.robustifier = cauchy_chi_threshold > 0.f ? LSM2D_ROBUST_CAUCHY : LSM2D_ROBUST_NONE; sp.chi_threshold = cauchy_chi_threshold > 0.f ? cauchy_chi_threshold : 0.f;
    sp.min_num_correspondences = min_num_correspondences;
    return sp;
  }
  // acceptance thresholds (MULTI.json:964-986)
  int param_relocalize_min_inliers = 500; float param_relocalize_max_chi_inliers = 0.1f, param_relocalize_min_inliers_ratio = 0.8f;

  void setMap(const PointNormal2fVectorCloud& map) { checkSweep(lsm2d_sweep_set_map(_sw, map.empty() ? nullptr : &map[0].x, (int64_t) map.size()), "set_map"); }
  void setScans(const std::vector<PointNormal2fVectorCloud>& scans) {
    std::vector<float> packed; std::vector<int32_t> offs(1, 0);
    for (const auto& c : scans) { for (const auto& p : c) { packed.push_back(p.x); packed.push_back(p.y); packed.push_back(p.nx); packed.push_back(p.ny); } offs.push_back((int32_t) (packed.size() / 4)); }
    checkSweep(lsm2d_sweep_set_scans(_sw, packed.data(), offs.data(), (int32_t) scans.size()), "set_scans");
  }
  void setScansPacked(const float* xynn, const int32_t* offsets, int n_scans) { checkSweep(lsm2d_sweep_set_scans(_sw, xynn, offsets, n_scans), "set_scans"); }
  // candidates: (scan index, initial guess of moving-in-fixed); results in candidate order
  void compute(const std::vector<int32_t>& scan_index, const std::vector<Vector3f>& init_pose) {
    const int n = (int) init_pose.size();
    if ((int) scan_index.size() != n) throw std::runtime_error("LoopClosureSweep::compute| one scan index per candidate");
    const lsm2d_slice_params sp = param_slice;
    lsm2d_aligner_params ap{param_max_iterations, param_min_num_inliers, 0.f, 0.f, 0, 0};
    pose.assign((size_t) n, Vector3f{{0.f, 0.f, 0.f}}); information.assign((size_t) n, std::array<float, 9>{});
    status.assign((size_t) n, 0); iterations.assign((size_t) n, 0); last_stats.assign((size_t) n, lsm2d_iteration_stats{});
    static_assert(sizeof(Vector3f) == 3 * sizeof(float), "poses are packed");
    checkSweep(lsm2d_sweep_align(_sw, &ap, &sp, n, scan_index.data(), n ? init_pose[0].data() : nullptr, n ? pose[0].data() : nullptr,
                                 n ? information[0].data() : nullptr, status.data(), iterations.data(), last_stats.data()), "align");
  }
  bool accept(size_t i) const {
    const auto& st = last_stats[i];
    const float n_in = (float) st.n_inliers, n_c = (float) (st.n_correspondences > 0 ? st.n_correspondences : 1);
    return status[i] == LSM2D_SUCCESS && st.n_inliers >= param_relocalize_min_inliers &&
           st.chi_inliers / (n_in > 1.f ? n_in : 1.f) <= param_relocalize_max_chi_inliers && n_in / n_c >= param_relocalize_min_inliers_ratio;
  }
  std::vector<Vector3f> pose; std::vector<std::array<float, 9>> information;
  std::vector<int32_t> status, iterations; std::vector<lsm2d_iteration_stats> last_stats;
 private:
  void checkSweep(int rc, const char* where) const {
    if (rc < 0) throw std::runtime_error(std::string("LoopClosureSweep::") + where + "| " + lsm2d_status_string(rc) + ": " + lsm2d_sweep_last_error(_sw));
  }
  lsm2d_sweep* _sw = nullptr;
};

}  // namespace lsm2d_host
