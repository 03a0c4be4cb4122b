// Shared plumbing of the SRRG-side adapters (adapters/srrg/*): device clouds that live as long as the adapter object, parameter
// translation from the reference's finder / projector / robustifier objects to the C ABI's plain structs, error conversion.
#pragma once
#include "upstream_access.h"

#include <srrg2_laser_slam_2d/registration/correspondence_finder_kd_tree_2d.h>
#include <srrg2_laser_slam_2d/registration/correspondence_finder_nn_2d.h>
#include <srrg2_laser_slam_2d/registration/correspondence_finder_projective_2d.h>
#include <srrg_pcl/point_projector_types.h>
#include <srrg_solver/solver_core/robustifier.h>

#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace lsm2d_srrg {

  // C-ABI errors become the exceptions the reference's modules throw (registration/correspondence_finder_projective_2d.cpp:21-31)
  inline void throwOnError(int rc_, const std::string& where_, const lsm2d_context* ctx_) {
    if (rc_ < 0) {
      throw std::runtime_error(where_ + "| " + lsm2d_status_string(rc_) + ": " + lsm2d_last_error(ctx_));
    }
  }

  // The device context the HIP modules of one process share: ONE device context -- one HIP stream, one set of staging buffers -- per device ordinal, created by the
  // first module that needs it and destroyed with the last one.  (Round 4: every finder sibling owned a context and a stream of its own -- a two-laser MULTI
  // configuration created three.)  A context is not thread-safe; the reference's compute path is single-threaded (finding 1), and so are its modules.
  struct SharedContext {
    lsm2d_context* ctx = nullptr;
    ~SharedContext() {
      lsm2d_destroy(ctx);
    }
  };
  inline std::shared_ptr<SharedContext> sharedContext(int device_id_, const std::string& who_) {
    static std::mutex mutex;
    static std::map<int, std::weak_ptr<SharedContext>> registry;
    std::lock_guard<std::mutex> lock(mutex);
    if (auto alive = registry[device_id_].lock()) {
      return alive;
    }
    auto fresh = std::make_shared<SharedContext>();
    throwOnError(lsm2d_create(device_id_, nullptr, &fresh->ctx), who_ + " create", nullptr);
    registry[device_id_] = fresh;
    return fresh;
  }
  // A context of its own, as a configurable: modules whose `context` PARAM points at the same HipContext share IT instead of the process-wide one
  // (a configuration that wants two independent streams on one device writes two of these).
  class HipContext : public srrg2_core::Configurable {
  public:
    PARAM(srrg2_core::PropertyInt, device_id, "HIP device ordinal", 0, 0);
    PARAM(srrg2_core::PropertyInt,
          sum_order,
          "0: H, b and the chi2 statistics are added in trees (fast); 1: pair after pair in the reference's order -- the aligner then equals the sequential fp32 "
          "restatement of the reference's factor loop bit for bit, at about 1.5 x the time (lsm2d.h, option sum_order)",
          0,
          0);
    ~HipContext() {
      lsm2d_destroy(_ctx);
    }
    lsm2d_context* handle(const std::string& who_) {
      if (!_ctx) {
        throwOnError(lsm2d_create(param_device_id.value(), nullptr, &_ctx), who_ + " create", nullptr);
        throwOnError(lsm2d_set_option(_ctx, "sum_order", param_sum_order.value() ? 1 : 0), who_ + " sum_order", _ctx);
      }
      return _ctx;
    }

  private:
    lsm2d_context* _ctx = nullptr;
  };
  using HipContextPtr = std::shared_ptr<HipContext>;

  // One PointNormal2fVectorCloud on the device, in a reserved set that is REFILLED (lsm2d_cloudset_upload: a copy into pinned
  // memory, unpacked by the kernel that reads it) instead of being created and destroyed per call.  Owns the set.
  class DeviceCloud {
  public:
    DeviceCloud() {
    }
    DeviceCloud(const DeviceCloud&) = delete;
    DeviceCloud& operator=(const DeviceCloud&) = delete;
    ~DeviceCloud() {
      lsm2d_cloudset_destroy(_set);
    }
    // PointNormal2f -> (x, y, nx, ny).  The preprocessor never emits invalid points
    // (sensor_processing/raw_data_preprocessor_projective_2d.cpp:42-47), so indices are the host cloud's.
    // uploadIfChanged: the same, unless the device already holds exactly these values -- same object, same size, and the packed floats compare EQUAL BYTE FOR BYTE
    // with what was uploaded last (a host-side copy of it is kept; round 5 trusted a 64-bit hash alone: a collision would have matched against a stale cloud, where
    // the reference re-projects the cloud it is handed every call).  The reference's own finders re-PROJECT the moving cloud every compute() but never re-copy it
    // (registration/correspondence_finder_projective_2d.cpp:37-48); under its aligner loop -- twenty compute() calls on an unchanged local map -- this sibling
    // uploads once too (1.6 MB over the host link per iteration at 100k points otherwise).  A cloud changed IN PLACE differs and is uploaded.
    // Returns true when an upload was queued.
    bool uploadIfChanged(lsm2d_context* ctx_, const PointNormal2fVectorCloud& cloud_, const char* who_) {
      return uploadImpl(ctx_, cloud_, who_, true);
    }
    void upload(lsm2d_context* ctx_, const PointNormal2fVectorCloud& cloud_, const char* who_) {
      uploadImpl(ctx_, cloud_, who_, false);
    }

  private:
    bool uploadImpl(lsm2d_context* ctx_, const PointNormal2fVectorCloud& cloud_, const char* who_, bool skip_if_same_) {
      const size_t n = cloud_.size();
      _staging.resize(4 * n);
      size_t k       = 0;
      for (const auto& p : cloud_) {
        _staging[k++] = p.coordinates().x(); _staging[k++] = p.coordinates().y(); _staging[k++] = p.normal().x(); _staging[k++] = p.normal().y();
      }
      if (skip_if_same_ && _set && _ctx == ctx_ && _host == &cloud_ && _n == n && _uploaded.size() == _staging.size() &&
          (n == 0 || memcmp(_uploaded.data(), _staging.data(), sizeof(float) * 4 * n) == 0)) {
        return false;
      }
      if (!_set || _ctx != ctx_ || n > _capacity) {
        lsm2d_cloudset_destroy(_set);
        _set      = nullptr;
        _capacity = n + n / 2 + 1024;
        throwOnError(lsm2d_cloudset_create_reserved(ctx_, (int64_t) _capacity, &_set), std::string(who_) + " reserve", ctx_);
      }
      throwOnError(lsm2d_cloudset_upload(_set, _staging.data(), (int64_t) n), std::string(who_) + " upload", ctx_);
      _host = &cloud_; _ctx = ctx_; _n = n;
      _uploaded.swap(_staging);      // (lsm2d_cloudset_upload copied the values into the set's pinned buffer: the vector is ours again)
      return true;
    }

  public:
    lsm2d_cloudset* set() const {
      return _set;
    }
    const PointNormal2fVectorCloud* host() const {
      return _host;
    }

  private:
    lsm2d_cloudset* _set                  = nullptr;
    size_t _capacity                      = 0;
    const PointNormal2fVectorCloud* _host = nullptr;
    const lsm2d_context* _ctx             = nullptr;
    size_t _n                             = 0;
    std::vector<float> _staging, _uploaded;      // this call's packed values; what the device holds
  };

  inline void fillProjector(const PointNormal2fProjectorPolar& projector_, lsm2d_projector* out_) {
    out_->canvas_cols = projector_.param_canvas_cols.value();
    out_->angle_min   = projector_.param_angle_col_min.value();
    out_->angle_max   = projector_.param_angle_col_max.value();
    out_->range_min   = projector_.param_range_min.value();
    out_->range_max   = projector_.param_range_max.value();
    out_->col_offset  = 0.f;
  }

  // HIP siblings of the reference finders implement this, so an aligner slice configured with one of THEM translates too
  class SliceParamsSource {
  public:
    virtual ~SliceParamsSource() {
    }
    virtual void fillSliceParams(lsm2d_slice_params* sp_) const = 0;
  };

  // finder object of a slice -> finder part of lsm2d_slice_params.  The three finders the reference registers
  // (instances.cpp:27-29) and their HIP siblings are understood; anything else is an error, never silently skipped.
  template <typename FinderPtr_>
  inline void fillFinderParams(const FinderPtr_& finder_, lsm2d_slice_params* sp_, const char* who_) {
    using namespace srrg2_laser_slam_2d;
    if (!finder_) {
      throw std::runtime_error(std::string(who_) + "| slice without a correspondence finder");
    }
    if (auto src = dynamic_cast<const SliceParamsSource*>(finder_.get())) {
      src->fillSliceParams(sp_);
    } else if (auto proj = dynamic_cast<const CorrespondenceFinderProjective2f*>(finder_.get())) {
      if (!proj->param_projector.value()) {
        throw std::runtime_error(std::string(who_) + "| Missing Projector");
      }
      sp_->finder = LSM2D_FINDER_PROJECTIVE;
      fillProjector(*proj->param_projector.value(), &sp_->projector);
      sp_->point_distance = proj->param_point_distance.value();
      sp_->normal_cos     = proj->param_normal_cos.value();
    } else if (auto kd = dynamic_cast<const CorrespondenceFinderKDTree2D*>(finder_.get())) {
      sp_->finder             = LSM2D_FINDER_KDTREE; // the reference's own tree and descent, with its own parameters
      sp_->max_distance       = kd->param_max_distance_m.value();
      sp_->normal_cos         = kd->param_normal_cos.value();
      sp_->kd_max_leaf_range  = kd->param_max_leaf_range.value();
      sp_->kd_min_leaf_points = (int32_t) kd->param_min_leaf_points.value();
    } else if (auto nn = dynamic_cast<const CorrespondenceFinderNN2D*>(finder_.get())) {
      sp_->finder       = LSM2D_FINDER_DISTMAP;
      sp_->max_distance = nn->param_max_distance_m.value();
      sp_->resolution   = nn->param_resolution.value();
      sp_->normal_cos   = nn->param_normal_cos.value();
    } else {
      throw std::runtime_error(std::string(who_) + "| unsupported correspondence finder type in a laser slice");
    }
  }

  template <typename RobustifierPtr_>
  inline void fillRobustifier(const RobustifierPtr_& robustifier_, lsm2d_slice_params* sp_, const char* who_) {
    sp_->robustifier   = LSM2D_ROBUST_NONE;
    sp_->chi_threshold = 0.f;
    if (!robustifier_) {
      return; // "#pointer": -1 (MULTI.json:184-187)
    }
    if (auto cauchy = dynamic_cast<const srrg2_solver::RobustifierCauchy*>(robustifier_.get())) {
      sp_->robustifier   = LSM2D_ROBUST_CAUCHY;
      sp_->chi_threshold = cauchy->param_chi_threshold.value();
      return;
    }
    throw std::runtime_error(std::string(who_) + "| unsupported robustifier type (only RobustifierCauchy runs on the device)");
  }

  inline void poseToArray(const Isometry2f& T_, float out_[3]) {
    const Vector3f v = geometry2d::t2v(T_);
    out_[0]          = v.x();
    out_[1]          = v.y();
    out_[2]          = v.z();
  }
} // namespace lsm2d_srrg
