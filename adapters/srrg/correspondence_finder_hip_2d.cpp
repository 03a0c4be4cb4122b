// See correspondence_finder_hip_2d.h.  Written against the upstream API exactly as the reference uses it in-tree
// (registration/correspondence_finder_projective_2d.cpp:18-77 for the base-class members _fixed, _moving, _correspondences,
// _local_map_in_sensor, _fixed_changed_flag; registration/correspondence_finder_kd_tree_2d.cpp:15-25 for the PointNormal2f
// accessors and the Correspondence ctor).
#include "correspondence_finder_hip_2d.h"

namespace srrg2_laser_slam_2d {
  using namespace srrg2_core;
  using lsm2d_srrg::throwOnError;

  CorrespondenceFinderHIPBase::~CorrespondenceFinderHIPBase() {
    // the context is shared (lsm2d_srrg::sharedContext) or a HipContext configurable's: whoever holds it last destroys it.  The device clouds (members) go
    // after this body; a set may outlive its context (lsm2d.h), so the order is safe either way
  }

  int64_t CorrespondenceFinderHIPBase::contextUploads() const {
    int64_t v = 0;
    if (_ctx) {
      lsm2d_get_option(_ctx, "uploads", &v);
    }
    return v;
  }

  void CorrespondenceFinderHIPBase::compute() {
    const std::string who = std::string(className()) + "::compute";
    // the reference's own preconditions and messages (registration/correspondence_finder_projective_2d.cpp:21-31)
    if (!_fixed) {
      throw std::runtime_error(who + "| Missing fixed!");
    }
    if (!_moving) {
      throw std::runtime_error(who + "| Missing moving!");
    }
    if (!_correspondences) {
      throw std::runtime_error(who + "| Missing correspondence vector!");
    }
    lsm2d_slice_params sp{};
    fillSliceParams(&sp); // throws on a missing projector / bad parameters, before anything touches the device
    if (!_ctx) {
      if (param_context.value()) {
        _ctx = param_context.value()->handle(who);
      } else {
        _shared = lsm2d_srrg::sharedContext(param_device_id.value(), who);
        _ctx    = _shared->ctx;
      }
    }
    // fixed: behind the base class's dirty flag, like the reference's cached canvas / tree (:37-44; kd_tree_2d.cpp:6-9)
    if (_fixed_changed_flag || !_fixed_dev.set() || _fixed_dev.host() != _fixed) {
      _fixed_dev.upload(_ctx, *_fixed, className());
      _fixed_changed_flag = false;
    }
    // moving: behind a CONTENT check (round 5).  The base class has no dirty flag for it as far as the reference's tree shows, and pointer + size say nothing
    // about contents -- the tracker's clipped scene is one object, cleared and refilled every step, often to the same size -- so the packed floats are compared,
    // byte for byte, with the ones uploaded last (round 6; a 64-bit hash before): an unchanged cloud (the reference's own aligner loop: twenty compute() calls on one local map) is not copied again, a changed one is.
    _moving_dev.uploadIfChanged(_ctx, *_moving, className());

    float pose[3];
    lsm2d_srrg::poseToArray(_local_map_in_sensor, pose);
    const size_t capacity = sp.finder == LSM2D_FINDER_PROJECTIVE ? (size_t) sp.projector.canvas_cols : _moving->size();
    _pairs.resize(capacity > 0 ? capacity : 1);
    int32_t k = 0;
    throwOnError(lsm2d_find_correspondences(_ctx, &sp, _fixed_dev.set(), 0, _moving_dev.set(), 0, pose, _pairs.data(), (int32_t) _pairs.size(), &k),
                 who, _ctx);
    _correspondences->resize(k);
    for (int32_t i = 0; i < k; ++i) {
      (*_correspondences)[i] = Correspondence(_pairs[i].fixed_idx, _pairs[i].moving_idx);
    }
  }

  void CorrespondenceFinderHIP2D::fillSliceParams(lsm2d_slice_params* sp_) const {
    if (!param_projector.value()) {
      throw std::runtime_error("CorrespondenceFinderHIP2D::compute| Missing Projector");
    }
    sp_->finder = LSM2D_FINDER_PROJECTIVE;
    lsm2d_srrg::fillProjector(*param_projector.value(), &sp_->projector);
    sp_->point_distance = param_point_distance.value();
    sp_->normal_cos     = param_normal_cos.value();
  }

  void CorrespondenceFinderKDTreeHIP2D::fillSliceParams(lsm2d_slice_params* sp_) const {
    const std::string& search = param_search.value();
    if (search == "kdtree") {
      sp_->finder = LSM2D_FINDER_KDTREE;
    } else if (search == "exact") {
      sp_->finder = LSM2D_FINDER_NN;
    } else {
      throw std::runtime_error("CorrespondenceFinderKDTreeHIP2D::compute| search must be \"kdtree\" or \"exact\", got \"" + search + "\"");
    }
    sp_->max_distance       = param_max_distance_m.value();
    sp_->normal_cos         = param_normal_cos.value();
    sp_->kd_max_leaf_range  = param_max_leaf_range.value();
    sp_->kd_min_leaf_points = (int32_t) param_min_leaf_points.value();
  }

  void CorrespondenceFinderNNHIP2D::fillSliceParams(lsm2d_slice_params* sp_) const {
    if (!(param_resolution.value() > 0.f)) { // registration/correspondence_finder_nn_2d.cpp:11-18
      throw std::runtime_error("CorrespondenceFinderNNHIP2D::compute| resolution must be positive");
    }
    sp_->finder       = LSM2D_FINDER_DISTMAP;
    sp_->max_distance = param_max_distance_m.value();
    sp_->resolution   = param_resolution.value();
    sp_->normal_cos   = param_normal_cos.value();
  }

} // namespace srrg2_laser_slam_2d
