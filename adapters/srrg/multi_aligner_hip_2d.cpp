// See multi_aligner_hip_2d.h.  NOT compiled in this repository's container (srrg2 stack absent); the calls into the
// upstream base class are limited to what the reference itself uses in-tree:
//   param_max_iterations / param_min_num_inliers / param_slice_processors   MULTI.json:700-732
//   slice->param_fixed_slice_name / param_moving_slice_name / param_finder / param_robustifier /
//   param_min_num_correspondences                                            MULTI.json:160-188
//   setMovingInFixed / movingInFixed                                          apps/visual_test_aligner_2d.cpp:126,145
// Members whose upstream names could not be verified here are marked  /*UPSTREAM*/ .
#include "multi_aligner_hip_2d.h"
#include <srrg_geometry/geometry2d.h>
#include <srrg_solver/solver_core/robustifier.h>

namespace srrg2_laser_slam_2d {
  using namespace srrg2_core;
  using namespace srrg2_slam_interfaces;

  MultiAlignerHIP2D::~MultiAlignerHIP2D() {
    lsm2d_destroy(_ctx);
  }

  static lsm2d_cloudset* uploadCloud(lsm2d_context* ctx_, const PointNormal2fVectorCloud& cloud_) {
    std::vector<float> staging(4 * cloud_.size());
    size_t k = 0;
    for (const auto& p : cloud_) {
      staging[k++] = p.coordinates().x();
      staging[k++] = p.coordinates().y();
      staging[k++] = p.normal().x();
      staging[k++] = p.normal().y();
    }
    lsm2d_cloudset* set = nullptr;
    if (lsm2d_cloudset_create(ctx_, staging.data(), nullptr, 1, (int64_t) cloud_.size(), &set) < 0) {
      throw std::runtime_error(std::string("MultiAlignerHIP2D| upload: ") + lsm2d_last_error(ctx_));
    }
    return set;
  }

  void MultiAlignerHIP2D::compute() {
    if (!_ctx && lsm2d_create(param_device_id.value(), nullptr, &_ctx) < 0) {
      throw std::runtime_error(std::string("MultiAlignerHIP2D::compute| ") + lsm2d_last_error(nullptr));
    }
    std::vector<lsm2d_slice_params> slices;
    std::vector<lsm2d_cloudset*> fixed_sets, moving_sets;
    lsm2d_prior prior{};
    bool has_prior = false;

    for (size_t s = 0; s < param_slice_processors.size(); ++s) {
      auto laser = std::dynamic_pointer_cast<AlignerSliceProcessorLaser2D>(param_slice_processors.value(s));
      auto laser_ws = std::dynamic_pointer_cast<AlignerSliceProcessorLaser2DWithSensor>(param_slice_processors.value(s));
      if (!laser && !laser_ws) {
        // non-laser cue (e.g. AlignerSliceOdom2DPrior, MULTI.json:402-422): expressed as the lsm2d prior
        // e = t2v(Z^-1 X) with information Omega.  /*UPSTREAM*/ accessor names of the prior slice:
        //   Isometry2f Z; Matrix3f omega;  -> fill prior.z = t2v(Z), prior.omega row-major, has_prior = true
        continue;
      }
      lsm2d_slice_params sp{};
      auto fill = [&](auto& slice_) {
        auto finder = std::dynamic_pointer_cast<CorrespondenceFinderProjective2f>(slice_->param_finder.value());
        if (!finder || !finder->param_projector.value()) {
          throw std::runtime_error("MultiAlignerHIP2D::compute| laser slice without a projective finder");
        }
        auto projector           = finder->param_projector.value();
        sp.finder                = LSM2D_FINDER_PROJECTIVE;
        sp.projector.canvas_cols = projector->param_canvas_cols.value();
        sp.projector.angle_min   = projector->param_angle_col_min.value();
        sp.projector.angle_max   = projector->param_angle_col_max.value();
        sp.projector.range_min   = projector->param_range_min.value();
        sp.projector.range_max   = projector->param_range_max.value();
        sp.point_distance        = finder->param_point_distance.value();
        sp.normal_cos            = finder->param_normal_cos.value();
        sp.min_num_correspondences = slice_->param_min_num_correspondences.value();
        if (auto cauchy = std::dynamic_pointer_cast<srrg2_solver::RobustifierCauchy>(slice_->param_robustifier.value())) {
          sp.robustifier   = LSM2D_ROBUST_CAUCHY;
          sp.chi_threshold = cauchy->param_chi_threshold.value();
        }
        // clouds by slice name out of the fixed / moving property containers
        // (apps/visual_test_aligner_2d.cpp:108-118): /*UPSTREAM*/ slice_->fixed() / slice_->moving() after bind
        fixed_sets.push_back(uploadCloud(_ctx, *slice_->fixed()));
        moving_sets.push_back(uploadCloud(_ctx, *slice_->moving()));
      };
      if (laser_ws) {
        fill(laser_ws);
        // WithSensor: sensor_in_robot from the tf Platform (registration/aligner_slice_processor_laser_2d_impl.cpp:7-10)
        const Vector3f sv = geometry2d::t2v(laser_ws->sensorInRobot() /*UPSTREAM*/);
        sp.sensor_in_robot[0] = sv.x(); sp.sensor_in_robot[1] = sv.y(); sp.sensor_in_robot[2] = sv.z();
      } else {
        fill(laser);
      }
      slices.push_back(sp);
    }

    const Vector3f x0  = geometry2d::t2v(movingInFixed());
    float pose[3]      = {x0.x(), x0.y(), x0.z()};
    float information[9];
    int32_t status = 0, iterations = 0;
    std::vector<const lsm2d_cloudset*> fx(fixed_sets.begin(), fixed_sets.end()), mv(moving_sets.begin(), moving_sets.end());
    lsm2d_batch batch{};
    batch.n_alignments = 1;
    batch.n_slices     = (int32_t) slices.size();
    batch.slices       = slices.data();
    batch.fixed        = fx.data();
    batch.moving       = mv.data();
    batch.init_pose    = pose;
    batch.prior        = has_prior ? &prior : nullptr;
    lsm2d_aligner_params ap{param_max_iterations.value(), param_min_num_inliers.value(), 0.f};
    std::vector<lsm2d_iteration_stats> stats(std::max(1, ap.max_iterations));
    const int rc = lsm2d_align_batch(_ctx, &ap, &batch, pose, information, &status, &iterations, stats.data());
    for (auto* s : fixed_sets) { lsm2d_cloudset_destroy(s); }
    for (auto* s : moving_sets) { lsm2d_cloudset_destroy(s); }
    if (rc < 0) {
      throw std::runtime_error(std::string("MultiAlignerHIP2D::compute| ") + lsm2d_last_error(_ctx));
    }
    setMovingInFixed(geometry2d::v2t(Vector3f(pose[0], pose[1], pose[2])));
    // status / information matrix / iteration stats back into the base class  /*UPSTREAM*/ member names:
    //   _status = Success | NotEnoughCorrespondences | NotEnoughInliers | Fail ; _information_matrix ; _iteration_stats
  }

} // namespace srrg2_laser_slam_2d
