// SRRG-side adapters for the two mapping steps either side of the aligner in the tracker (same caveat as
// correspondence_finder_hip_2d.h: compiles only inside a catkin workspace with the srrg2 stack; members whose upstream
// names could not be verified in this repository's container are tagged /*UPSTREAM*/).
//
//   SceneClipperHIP2D  sibling of SceneClipperProjective2D (mapping/scene_clipper_projective_2d.h:8-36, .cpp:11-65)
//   MergerHIP2D        sibling of MergerProjective2D       (mapping/merger_projective_2d.h:6-37,  .cpp:9-100)
//
// Both keep the local map in a reserved device cloud (lsm2d_cloudset_create_reserved) between calls, so the per-scan
// upload is the scan only; the host-side PointNormal2fVectorCloud the rest of the pipeline reads is refreshed with
// lsm2d_cloudset_download when a caller asks for it.
#pragma once
#include <lsm2d.h>
#include <srrg2_laser_slam_2d/mapping/merger_point_normal_2f.h>
#include <srrg2_laser_slam_2d/mapping/scene_clipper_point_normal_2f.h>
#include <srrg_config/property_configurable.h>
#include <srrg_pcl/point_projector_types.h>

namespace srrg2_laser_slam_2d {

  class SceneClipperHIP2D : public SceneClipperPointNormal2f {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    PARAM(srrg2_core::PropertyConfigurable_<srrg2_core::PointNormal2fProjectorPolar>,
          projector,
          "projector used to remap the points",
          srrg2_core::PointNormal2fProjectorPolarPtr(new srrg2_core::PointNormal2fProjectorPolar),
          nullptr);
    PARAM(srrg2_core::PropertyInt, device_id, "HIP device ordinal", 0, 0);
    virtual ~SceneClipperHIP2D() {
      lsm2d_cloudset_destroy(_scene_set);
      lsm2d_cloudset_destroy(_clipped_set);
      lsm2d_destroy(_ctx);
    }
    void compute() override {
      using namespace srrg2_core;
      if (!_clipped_scene_in_robot || !_full_scene) { // scene_clipper_projective_2d.cpp:12-17
        _status = Error;
        return;
      }
      if (!param_projector.value()) {
        throw std::runtime_error("SceneClipperHIP2D::compute| Missing Projector");
      }
      if (!_ctx && lsm2d_create(param_device_id.value(), nullptr, &_ctx) < 0) {
        throw std::runtime_error(lsm2d_last_error(nullptr));
      }
      auto projector = param_projector.value();
      lsm2d_projector pr{projector->param_canvas_cols.value(), projector->param_angle_col_min.value(), projector->param_angle_col_max.value(),
                         projector->param_range_min.value(), projector->param_range_max.value(), 0.f};
      // upload the full scene (a MergerHIP2D sharing the context can hand its device-resident scene over instead)
      std::vector<float> staging(4 * _full_scene->size());
      size_t k = 0;
      for (const auto& p : *_full_scene) {
        staging[k++] = p.coordinates().x(); staging[k++] = p.coordinates().y();
        staging[k++] = p.normal().x();      staging[k++] = p.normal().y();
      }
      lsm2d_cloudset_destroy(_scene_set); _scene_set = nullptr;
      if (lsm2d_cloudset_create(_ctx, staging.data(), nullptr, 1, (int64_t) _full_scene->size(), &_scene_set) < 0) {
        throw std::runtime_error(lsm2d_last_error(_ctx));
      }
      if (!_clipped_set && lsm2d_cloudset_create_reserved(_ctx, pr.canvas_cols, &_clipped_set) < 0) {
        throw std::runtime_error(lsm2d_last_error(_ctx));
      }
      const Vector3f r = geometry2d::t2v(_robot_in_local_map), s = geometry2d::t2v(_sensor_in_robot);
      const float robot[3] = {r.x(), r.y(), r.z()}, sensor[3] = {s.x(), s.y(), s.z()};
      int32_t n = 0;
      if (lsm2d_clip_scene(_ctx, &pr, _scene_set, 0, robot, sensor, _clipped_set, &n, nullptr) < 0) {
        throw std::runtime_error(lsm2d_last_error(_ctx));
      }
      staging.resize(4 * (size_t) n);
      int64_t got = 0;
      lsm2d_cloudset_download(_clipped_set, 0, staging.data(), n, &got);
      _clipped_scene_in_robot->resize(got);
      for (int64_t i = 0; i < got; ++i) {
        auto& p = (*_clipped_scene_in_robot)[i];
        p.coordinates() << staging[4 * i], staging[4 * i + 1];
        p.normal() << staging[4 * i + 2], staging[4 * i + 3];
      }
      _status = Successful;
    }
    // the clipped scene as a device cloud: what MultiAlignerHIP2D takes as `moving` without another upload
    const lsm2d_cloudset* clippedOnDevice() const { return _clipped_set; }

  protected:
    lsm2d_context* _ctx          = nullptr;
    lsm2d_cloudset* _scene_set   = nullptr;
    lsm2d_cloudset* _clipped_set = nullptr;
  };

  class MergerHIP2D : public MergerPointNormal2f {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    PARAM(srrg2_core::PropertyFloat, merge_threshold, "max distance for merging the points in the scene and the moving", 0.2f, 0);
    PARAM(srrg2_core::PropertyConfigurable_<srrg2_core::PointNormal2fProjectorPolar>,
          projector,
          "projector to compute correspondences",
          srrg2_core::PointNormal2fProjectorPolarPtr(new srrg2_core::PointNormal2fProjectorPolar),
          nullptr);
    PARAM(srrg2_core::PropertyInt, device_id, "HIP device ordinal", 0, 0);
    PARAM(srrg2_core::PropertyInt, capacity, "points the device-resident scene can hold", 1000000, 0);
    virtual ~MergerHIP2D() {
      lsm2d_cloudset_destroy(_scene_set);
      lsm2d_destroy(_ctx);
    }
    void compute() override {
      using namespace srrg2_core;
      if (!param_projector.value()) {
        throw std::runtime_error("MergerHIP2D::compute| Missing Projector"); // merger_projective_2d.cpp:10-12
      }
      if (!_ctx && lsm2d_create(param_device_id.value(), nullptr, &_ctx) < 0) {
        throw std::runtime_error(lsm2d_last_error(nullptr));
      }
      auto projector = param_projector.value();
      lsm2d_projector pr{projector->param_canvas_cols.value(), projector->param_angle_col_min.value(), projector->param_angle_col_max.value(),
                         projector->param_range_min.value(), projector->param_range_max.value(), 0.f};
      std::vector<float> staging;
      auto pack = [&](const PointNormal2fVectorCloud& cloud_) {
        staging.resize(4 * cloud_.size());
        size_t k = 0;
        for (const auto& p : cloud_) {
          staging[k++] = p.coordinates().x(); staging[k++] = p.coordinates().y();
          staging[k++] = p.normal().x();      staging[k++] = p.normal().y();
        }
      };
      if (!_scene_set) { // first call: seed the device scene with the host scene
        if (lsm2d_cloudset_create_reserved(_ctx, param_capacity.value(), &_scene_set) < 0) {
          throw std::runtime_error(lsm2d_last_error(_ctx));
        }
        pack(*_scene);
        lsm2d_cloudset_upload(_scene_set, staging.data(), (int64_t) _scene->size());
      }
      pack(*_measurement);
      lsm2d_cloudset* meas = nullptr;
      if (lsm2d_cloudset_create(_ctx, staging.data(), nullptr, 1, (int64_t) _measurement->size(), &meas) < 0) {
        throw std::runtime_error(lsm2d_last_error(_ctx));
      }
      const Vector3f m = geometry2d::t2v(_measurement_in_scene);
      const float mis[3]  = {m.x(), m.y(), m.z()};
      int32_t size = 0, counts[3];
      const int rc = lsm2d_merge_scene(_ctx, &pr, _scene_set, meas, 0, mis, param_merge_threshold.value(), &size, counts);
      lsm2d_cloudset_destroy(meas);
      if (rc < 0) {
        throw std::runtime_error(lsm2d_last_error(_ctx));
      }
      // mirror the result into the host cloud the rest of the pipeline reads
      staging.resize(4 * (size_t) size);
      int64_t got = 0;
      lsm2d_cloudset_download(_scene_set, 0, staging.data(), size, &got);
      _scene->resize(got);
      for (int64_t i = 0; i < got; ++i) {
        auto& p = (*_scene)[i];
        p.coordinates() << staging[4 * i], staging[4 * i + 1];
        p.normal() << staging[4 * i + 2], staging[4 * i + 3];
      }
      _status = MergerBase::Status::Success; // merger_projective_2d.cpp:99
    }

  protected:
    lsm2d_context* _ctx        = nullptr;
    lsm2d_cloudset* _scene_set = nullptr;
  };

} // namespace srrg2_laser_slam_2d
