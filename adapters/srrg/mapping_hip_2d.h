// SRRG-side adapters for the two mapping steps either side of the aligner in the tracker.
//
//   SceneClipperHIP2D  sibling of SceneClipperProjective2D (mapping/scene_clipper_projective_2d.h:8-36, .cpp:11-65)
//   MergerHIP2D        sibling of MergerProjective2D       (mapping/merger_projective_2d.h:6-37,  .cpp:9-100)
//
// Both keep their clouds in reserved device sets between calls (refilled, never re-allocated); the merger keeps the local map
// itself on the device, so the per-scan upload is the scan only, and mirrors the merged scene into the host cloud the rest of the
// pipeline reads.  Base-class members are the ones the reference's own implementations use (scene_clipper_projective_2d.cpp:12-64,
// merger_projective_2d.cpp:17-99).  Compile-checked and driven on the GPU against tests/cpp/adapter_shim in this repository.
#pragma once
#include "lsm2d_srrg_common.h"

#include <srrg2_laser_slam_2d/mapping/merger_point_normal_2f.h>
#include <srrg2_laser_slam_2d/mapping/scene_clipper_point_normal_2f.h>

namespace srrg2_laser_slam_2d {

  namespace hip_detail {
    inline void unpackCloud(const std::vector<float>& staging_, int64_t n_, srrg2_core::PointNormal2fVectorCloud* out_) {
      out_->resize((size_t) n_);
      for (int64_t i = 0; i < n_; ++i) {
        auto& p               = (*out_)[(size_t) i];
        p.coordinates().x()   = staging_[4 * i];
        p.coordinates().y()   = staging_[4 * i + 1];
        p.normal().x()        = staging_[4 * i + 2];
        p.normal().y()        = staging_[4 * i + 3];
      }
    }
  } // namespace hip_detail

  class SceneClipperHIP2D : public SceneClipperPointNormal2f {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    PARAM(srrg2_core::PropertyConfigurable_<srrg2_core::PointNormal2fProjectorPolar>,
          projector,
          "projector used to remap the points",
          srrg2_core::PointNormal2fProjectorPolarPtr(new srrg2_core::PointNormal2fProjectorPolar),
          nullptr);
    PARAM(srrg2_core::PropertyFloat, voxelize_resolution, "resolution used to decimate the points in the scan on a grid [meters]", 0.1, nullptr);
    PARAM(srrg2_core::PropertyInt, device_id, "HIP device ordinal", 0, 0);
    virtual ~SceneClipperHIP2D() {
      lsm2d_cloudset_destroy(_clipped_set);
      lsm2d_destroy(_ctx);
    }
    void compute() override {
      using namespace srrg2_core;
      const char* who = "SceneClipperHIP2D::compute";
      if (!_clipped_scene_in_robot || !_full_scene) { // scene_clipper_projective_2d.cpp:12-17
        _status = Error;
        return;
      }
      if (!param_projector.value()) {
        throw std::runtime_error(std::string(who) + "| Missing Projector");
      }
      if (!_ctx) {
        lsm2d_srrg::throwOnError(lsm2d_create(param_device_id.value(), nullptr, &_ctx), who, nullptr);
      }
      lsm2d_projector pr{};
      lsm2d_srrg::fillProjector(*param_projector.value(), &pr);
      _scene_dev.upload(_ctx, *_full_scene, who);
      if (!_clipped_set || _clipped_capacity < pr.canvas_cols) {
        lsm2d_cloudset_destroy(_clipped_set);
        _clipped_set = nullptr;
        lsm2d_srrg::throwOnError(lsm2d_cloudset_create_reserved(_ctx, pr.canvas_cols, &_clipped_set), who, _ctx);
        _clipped_capacity = pr.canvas_cols;
      }
      float robot[3], sensor[3];
      lsm2d_srrg::poseToArray(_robot_in_local_map, robot);
      lsm2d_srrg::poseToArray(_sensor_in_robot, sensor);
      int32_t n = 0;
      lsm2d_srrg::throwOnError(
        lsm2d_clip_scene_voxelized(_ctx, &pr, _scene_dev.set(), 0, robot, sensor, param_voxelize_resolution.value(), _clipped_set, &n, nullptr), who, _ctx);
      _staging.resize(4 * (size_t) (n > 0 ? n : 1));
      int64_t got = 0;
      lsm2d_srrg::throwOnError(lsm2d_cloudset_download(_clipped_set, 0, _staging.data(), n, &got), who, _ctx);
      hip_detail::unpackCloud(_staging, got, _clipped_scene_in_robot);
      _status = Successful;
    }
    // the clipped scene as a device cloud (what an aligner on the same device can take as `moving` without another upload)
    const lsm2d_cloudset* clippedOnDevice() const {
      return _clipped_set;
    }

  protected:
    lsm2d_context* _ctx          = nullptr;
    lsm2d_srrg::DeviceCloud _scene_dev;
    lsm2d_cloudset* _clipped_set = nullptr;
    int _clipped_capacity        = 0;
    std::vector<float> _staging;
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
      const char* who = "MergerHIP2D::compute";
      if (!param_projector.value()) {
        throw std::runtime_error(std::string(who) + "| Missing Projector"); // merger_projective_2d.cpp:10-12
      }
      if (!_scene || !_measurement) {
        throw std::runtime_error(std::string(who) + "| scene / measurement not set");
      }
      if (!_ctx) {
        lsm2d_srrg::throwOnError(lsm2d_create(param_device_id.value(), nullptr, &_ctx), who, nullptr);
      }
      lsm2d_projector pr{};
      lsm2d_srrg::fillProjector(*param_projector.value(), &pr);
      // the device scene is seeded from the host scene when the object changes or somebody else resized it (a new local map)
      if (!_scene_set || _scene_host != _scene || _scene_host_size != _scene->size()) {
        if (!_scene_set) {
          lsm2d_srrg::throwOnError(lsm2d_cloudset_create_reserved(_ctx, param_capacity.value(), &_scene_set), who, _ctx);
        }
        _staging.resize(4 * _scene->size());
        size_t k = 0;
        for (const auto& p : *_scene) {
          _staging[k++] = p.coordinates().x(); _staging[k++] = p.coordinates().y();
          _staging[k++] = p.normal().x();      _staging[k++] = p.normal().y();
        }
        lsm2d_srrg::throwOnError(lsm2d_cloudset_upload(_scene_set, _staging.data(), (int64_t) _scene->size()), who, _ctx);
        _scene_host = _scene;
      }
      _measurement_dev.upload(_ctx, *_measurement, who);
      float mis[3];
      lsm2d_srrg::poseToArray(_measurement_in_scene, mis);
      int32_t size = 0, counts[3];
      lsm2d_srrg::throwOnError(
        lsm2d_merge_scene(_ctx, &pr, _scene_set, _measurement_dev.set(), 0, mis, param_merge_threshold.value(), &size, counts), who, _ctx);
      // mirror the result into the host cloud the rest of the pipeline reads
      _staging.resize(4 * (size_t) (size > 0 ? size : 1));
      int64_t got = 0;
      lsm2d_srrg::throwOnError(lsm2d_cloudset_download(_scene_set, 0, _staging.data(), size, &got), who, _ctx);
      hip_detail::unpackCloud(_staging, got, _scene);
      _scene_host_size = _scene->size();
      _status          = MergerBase::Status::Success; // merger_projective_2d.cpp:99
    }

  protected:
    lsm2d_context* _ctx        = nullptr;
    lsm2d_cloudset* _scene_set = nullptr;
    const srrg2_core::PointNormal2fVectorCloud* _scene_host = nullptr;
    size_t _scene_host_size    = 0;
    lsm2d_srrg::DeviceCloud _measurement_dev;
    std::vector<float> _staging;
  };

} // namespace srrg2_laser_slam_2d
