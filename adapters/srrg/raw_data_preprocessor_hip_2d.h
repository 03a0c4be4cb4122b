// SRRG-side adapter for the step in front of the aligner:
//
//   RawDataPreprocessorHIP2D  sibling of RawDataPreprocessorProjective2D (sensor_processing/raw_data_preprocessor_projective_2d.h:16-56,
//                             .cpp:12-104): LaserMessage -> PointNormal2fVectorCloud (polar unprojection, sliding-window normals,
//                             voxelisation) on the device.
//
// Same PARAM names, property types and defaults as the reference class, so its configuration block (MULTI.json:488-515) loads with the
// class name swapped; the un-projector object is kept and set per message exactly as the reference does (.cpp:96-101) because other
// modules may share it through the configuration.  One scan per call: lsm2d_preprocess_scans with n_scans = 1, the cloud downloaded into
// the caller's measurement.  Compile-checked and driven on the GPU against tests/cpp/adapter_shim in this repository.
#pragma once
#include "lsm2d_srrg_common.h"

#include <srrg2_slam_interfaces/raw_data_preprocessors/raw_data_preprocessor.h>
#include <srrg_messages/messages/laser_message.h>
#include <srrg_pcl/normal_computator.h>
#include <srrg_pcl/point_unprojector_types.h>

#include <algorithm>

namespace srrg2_laser_slam_2d {

  class RawDataPreprocessorHIP2D : public srrg2_slam_interfaces::RawDataPreprocessor_<srrg2_core::PointNormal2fVectorCloud> {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    using BaseType             = srrg2_slam_interfaces::RawDataPreprocessor_<srrg2_core::PointNormal2fVectorCloud>;
    using MeasurementType      = typename BaseType::MeasurementType;
    using NormalComputatorType = srrg2_core::NormalComputator1DSlidingWindow<MeasurementType, 1>;

    PARAM(srrg2_core::PropertyConfigurable_<srrg2_core::PointNormal2fUnprojectorPolar>,
          unprojector,
          "un-projector used to compute the scan from the cloud",
          srrg2_core::PointNormal2fUnprojectorPolarPtr(new srrg2_core::PointNormal2fUnprojectorPolar()),
          nullptr);
    PARAM(srrg2_core::PropertyConfigurable_<NormalComputatorType>,
          normal_computator_sliding,
          "normal computator object",
          std::shared_ptr<NormalComputatorType>(new NormalComputatorType()),
          nullptr);
    PARAM(srrg2_core::PropertyFloat, range_min, "range_min [meters]", 0.0, nullptr);
    PARAM(srrg2_core::PropertyFloat, range_max, "range_max [meters]", 1000.0, nullptr);
    PARAM(srrg2_core::PropertyFloat, voxelize_resolution, "unproject voxelization resolution", 0.02, nullptr);
    PARAM(srrg2_core::PropertyString, scan_topic, "topic of the scan", "/scan", nullptr);
    PARAM(srrg2_core::PropertyInt, device_id, "HIP device ordinal", 0, 0);

    virtual ~RawDataPreprocessorHIP2D() {
      lsm2d_destroy(_ctx);
    }

    bool setRawData(srrg2_core::BaseSensorMessagePtr msg_) override {
      using namespace srrg2_core;
      if (!msg_) {
        throw std::runtime_error("RawDataPreprocessorHIP2D::setMeasurement|measurement is not set"); // .cpp:54-57
      }
      BaseType::setRawData(msg_);
      _status = Error;
      LaserMessagePtr laser_message = srrg2_slam_interfaces::extractMessage<LaserMessage>(msg_, param_scan_topic.value());
      if (!laser_message) {
        return false; // .cpp:64-69: not this module's topic
      }
      _ranges = &laser_message->ranges.value();
      if (!param_unprojector.value()) {
        throw std::runtime_error("RawDataPreprocessorHIP2D::_processLaserMessage|missing unprojector"); // .cpp:92-95
      }
      // .cpp:81-85: the limits of the message and of the module, and the sensor matrix the un-projector gets
      _range_max = std::min(laser_message->range_max.value(), param_range_max.value());
      _range_min = std::max(laser_message->range_min.value(), param_range_min.value());
      _angle_max = laser_message->angle_max.value();
      _angle_min = laser_message->angle_min.value();
      const float sensor_res = (_angle_max - _angle_min) / (float) _ranges->size();
      Matrix2f sensor_matrix(Matrix2f::Identity()); // .cpp:89-90, Eigen's comma initialiser as the reference writes it
      sensor_matrix << 1.f / sensor_res, (float) _ranges->size() / 2.f, 0, 0;
      PointNormal2fUnprojectorPolarPtr unprojector = param_unprojector.value();
      unprojector->param_range_min.setValue(_range_min); // .cpp:96-101: a shared un-projector sees the same values as with the reference module
      unprojector->param_range_max.setValue(_range_max);
      unprojector->param_angle_max.setValue(_angle_max);
      unprojector->param_angle_min.setValue(_angle_min);
      unprojector->setCameraMatrix(sensor_matrix);
      _status = Ready;
      return true;
    }

    void compute() override {
      const char* who = "RawDataPreprocessorHIP2D::compute";
      if (!_meas || !_raw_data || !_ranges) { // .cpp:13-17
        _status = Error;
        return;
      }
      if (!param_unprojector.value()) {
        throw std::runtime_error(std::string(who) + "| missing unprojector");
      }
      if (!param_normal_computator_sliding.value()) {
        throw std::runtime_error(std::string(who) + "| missing normal computator");
      }
      if (!_ctx) {
        lsm2d_srrg::throwOnError(lsm2d_create(param_device_id.value(), nullptr, &_ctx), who, nullptr);
      }
      lsm2d_preprocessor pp{};
      pp.n_beams               = (int32_t) _ranges->size();
      pp.angle_min             = _angle_min;
      pp.angle_max             = _angle_max;
      pp.range_min             = _range_min;
      pp.range_max             = _range_max;
      pp.normal_point_distance = param_normal_computator_sliding->param_normal_point_distance.value();
      pp.normal_min_points     = param_normal_computator_sliding->param_normal_min_points.value();
      pp.voxelize_resolution   = param_voxelize_resolution.value();
      lsm2d_cloudset* set = nullptr;
      lsm2d_srrg::throwOnError(lsm2d_preprocess_scans(_ctx, &pp, _ranges->data(), 1, &set), who, _ctx);
      const int64_t n = lsm2d_cloudset_num_points(set);
      _staging.resize(4 * (size_t) (n > 0 ? n : 1));
      int64_t got = 0;
      const int rc = lsm2d_cloudset_download(set, 0, _staging.data(), n, &got);
      lsm2d_cloudset_destroy(set);
      lsm2d_srrg::throwOnError(rc, who, _ctx);
      _meas->clear();
      _meas->reserve(_ranges->size()); // .cpp:42-43
      _meas->resize((size_t) got);
      for (int64_t i = 0; i < got; ++i) {
        auto& p             = (*_meas)[(size_t) i];
        p.coordinates().x() = _staging[4 * i];
        p.coordinates().y() = _staging[4 * i + 1];
        p.normal().x()      = _staging[4 * i + 2];
        p.normal().y()      = _staging[4 * i + 3];
      }
      _status = Ready;
    }

  protected:
    std::vector<float>* _ranges = nullptr; // laser scan fields - mandatory (.h:51-52)
    float _range_min = 0.f, _range_max = 0.f, _angle_min = 0.f, _angle_max = 0.f;
    lsm2d_context* _ctx = nullptr;
    std::vector<float> _staging;
  };

  using RawDataPreprocessorHIP2DPtr = std::shared_ptr<RawDataPreprocessorHIP2D>;

} // namespace srrg2_laser_slam_2d
