// See correspondence_finder_hip_2d.h.  Written against the upstream API exactly as the reference uses it in-tree
// (registration/correspondence_finder_projective_2d.cpp:18-77 for the base-class members _fixed, _moving,
// _correspondences, _local_map_in_sensor, _fixed_changed_flag; registration/correspondence_finder_kd_tree_2d.cpp:15-25
// for PointNormal2f accessors and the Correspondence ctor).  NOT compiled in this repository's container.
#include "correspondence_finder_hip_2d.h"
#include <srrg_geometry/geometry2d.h>

namespace srrg2_laser_slam_2d {
  using namespace srrg2_core;

  static void lsm2dThrow(int rc_, const char* where_, const lsm2d_context* ctx_) {
    if (rc_ < 0) {
      throw std::runtime_error(std::string("CorrespondenceFinderHIP2D::") + where_ + "| " +
                               lsm2d_status_string(rc_) + " " + lsm2d_last_error(ctx_));
    }
  }

  CorrespondenceFinderHIP2D::CorrespondenceFinderHIP2D() {
  }

  CorrespondenceFinderHIP2D::~CorrespondenceFinderHIP2D() {
    lsm2d_cloudset_destroy(_fixed_set);
    lsm2d_cloudset_destroy(_moving_set);
    lsm2d_destroy(_ctx);
  }

  void CorrespondenceFinderHIP2D::_upload(const PointNormal2fVectorCloud& cloud_, lsm2d_cloudset*& set_) {
    // PointNormal2f -> (x, y, nx, ny); invalid points are dropped on upload, so indices refer to the
    // compacted cloud only when the cloud holds invalid points (the preprocessor never emits them:
    // sensor_processing/raw_data_preprocessor_projective_2d.cpp:42-47)
    _staging.resize(4 * cloud_.size());
    size_t k = 0;
    for (const auto& p : cloud_) {
      _staging[k++] = p.coordinates().x();
      _staging[k++] = p.coordinates().y();
      _staging[k++] = p.normal().x();
      _staging[k++] = p.normal().y();
    }
    lsm2d_cloudset_destroy(set_);
    set_ = nullptr;
    lsm2dThrow(lsm2d_cloudset_create(_ctx, _staging.data(), nullptr, 1, (int64_t) cloud_.size(), &set_), "upload", _ctx);
  }

  void CorrespondenceFinderHIP2D::compute() {
    PointNormal2fProjectorPolarPtr projector = this->param_projector.value();
    if (!projector) {
      throw std::runtime_error("CorrespondenceFinderHIP2D::compute| Missing Projector");
    }
    if (!_fixed) {
      throw std::runtime_error("CorrespondenceFinderHIP2D::compute| Missing fixed!");
    }
    if (!_moving) {
      throw std::runtime_error("CorrespondenceFinderHIP2D::compute| Missing moving!");
    }
    if (!_ctx) {
      lsm2dThrow(lsm2d_create(param_device_id.value(), nullptr, &_ctx), "create", nullptr);
    }
    if (this->_fixed_changed_flag || !_fixed_set) {
      _upload(*_fixed, _fixed_set);
      _fixed_changed_flag = false;
    }
    // the base class has no dirty flag for moving: re-upload when the cloud object or its size changes
    if (!_moving_set || _moving_uploaded != (const void*) _moving || _moving_uploaded_size != _moving->size()) {
      _upload(*_moving, _moving_set);
      _moving_uploaded      = (const void*) _moving;
      _moving_uploaded_size = _moving->size();
    }
    _projector_changed_flag = false;

    lsm2d_slice_params sp{};
    sp.finder                = LSM2D_FINDER_PROJECTIVE;
    sp.projector.canvas_cols = projector->param_canvas_cols.value();
    sp.projector.angle_min   = projector->param_angle_col_min.value();
    sp.projector.angle_max   = projector->param_angle_col_max.value();
    sp.projector.range_min   = projector->param_range_min.value();
    sp.projector.range_max   = projector->param_range_max.value();
    sp.projector.col_offset  = 0.f;
    sp.point_distance        = param_point_distance.value();
    sp.normal_cos            = param_normal_cos.value();

    const Vector3f v = geometry2d::t2v(_local_map_in_sensor);
    const float pose[3] = {v.x(), v.y(), v.z()};
    _pairs.resize(sp.projector.canvas_cols);
    int32_t k = 0;
    lsm2dThrow(lsm2d_find_correspondences(_ctx, &sp, _fixed_set, 0, _moving_set, 0, pose, _pairs.data(), (int32_t) _pairs.size(), &k),
               "compute", _ctx);
    _correspondences->resize(k);
    for (int32_t i = 0; i < k; ++i) {
      _correspondences->at(i) = Correspondence(_pairs[i].fixed_idx, _pairs[i].moving_idx);
    }
  }

} // namespace srrg2_laser_slam_2d
