// See multi_aligner_hip_2d.h.  Calls into the upstream base class are limited to what the reference itself uses in-tree:
//   param_max_iterations / param_min_num_inliers / param_slice_processors    MULTI.json:700-732
//   slice->param_fixed_slice_name / param_moving_slice_name / param_finder / param_robustifier /
//   param_min_num_correspondences                                             MULTI.json:160-188
//   setMovingInFixed / movingInFixed / iterationStats                          apps/visual_test_aligner_2d.cpp:126,145,156
// everything else goes through adapters/srrg/upstream_access.h (tagged UPSTREAM there) or _writeBack() below.
#include "multi_aligner_hip_2d.h"

#include <functional>
#include <iterator>
#include <set>

namespace srrg2_laser_slam_2d {
  using namespace srrg2_core;
  using namespace srrg2_slam_interfaces;
  using lsm2d_srrg::throwOnError;

  MultiAlignerHIP2D::~MultiAlignerHIP2D() {
    _device_clouds.clear();
    lsm2d_destroy(_ctx);
  }

  void MultiAlignerHIP2D::_writeBack(int status_, const float information_[9], int iterations_, const std::vector<lsm2d_iteration_stats>& stats_) {
    // UPSTREAM: member names of the aligner base class behind status() / informationMatrix() / iterationStats()
    switch (status_) {
      case LSM2D_SUCCESS: _status = Success; break;
      case LSM2D_NOT_ENOUGH_CORRESPONDENCES: _status = NotEnoughCorrespondences; break;
      case LSM2D_NOT_ENOUGH_INLIERS: _status = NotEnoughInliers; break;
      default: _status = Fail; break; // LSM2D_SINGULAR_H: the reference's linear solver would fail here
    }
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) {
        _information_matrix(r, c) = information_[3 * r + c];
      }
    }
    _iteration_stats.clear();
    for (int it = 0; it < iterations_; ++it) {
      srrg2_solver::IterationStats st;
      st.iteration    = it;
      st.num_inliers  = stats_[it].n_inliers;
      st.num_outliers = stats_[it].n_outliers;
      st.chi_inliers  = stats_[it].chi_inliers;
      st.chi_outliers = stats_[it].chi_outliers;
      _iteration_stats.push_back(st);
    }
  }

  void MultiAlignerHIP2D::compute() {
    const char* who = "MultiAlignerHIP2D::compute";
    if (!_fixed || !_moving) {
      throw std::runtime_error(std::string(who) + "| fixed / moving scene not set");
    }
    // the upstream aligner's remaining options (MULTI.json:606-610,704-708,627-630,729-731) go to the device loop as they are
    // (lsm2d_aligner_params; semantics in include/lsm2d.h).  A termination_criteria OBJECT is translated into the epsilon it carries
    // (upstream_access.h: terminationCriteriaEpsilon); one without such a property is refused -- never silently dropped.
    float chi_epsilon = param_termination_chi_epsilon.value();
    if (!(chi_epsilon >= 0.f)) {
      throw std::runtime_error(std::string(who) + "| termination_chi_epsilon must be >= 0");
    }
    if (lsm2d_srrg::terminationCriteriaSet(*this)) {
      const float from_object = lsm2d_srrg::terminationCriteriaEpsilon(*this);
      if (!(from_object >= 0.f)) {
        throw std::runtime_error(std::string(who) + "| the termination_criteria object carries no float property 'epsilon': it cannot be translated for the device");
      }
      if (chi_epsilon > 0.f && chi_epsilon != from_object) {
        throw std::runtime_error(std::string(who) + "| termination_chi_epsilon and the termination_criteria object's epsilon disagree");
      }
      chi_epsilon = from_object;
    }
    if (!_ctx) {
      throwOnError(lsm2d_create(param_device_id.value(), nullptr, &_ctx), std::string(who) + " create", nullptr);
    }
    std::vector<lsm2d_slice_params> slices;
    std::vector<const lsm2d_cloudset*> fixed_sets, moving_sets;
    struct PublishTarget {      // where a laser slice's pairs go when publish_correspondences is on
      std::function<CorrespondenceVector&()> correspondences;
      size_t n_moving;
    };
    std::vector<PublishTarget> laser_slices;
    lsm2d_prior prior{};
    bool has_prior = false;
    std::set<const PointNormal2fVectorCloud*> refreshed; // a cloud shared by several slices is uploaded once per compute()

    auto deviceCloud = [&](PointNormal2fVectorCloud* cloud_) -> lsm2d_cloudset* {
      auto& slot = _device_clouds[cloud_];
      if (!slot) {
        slot.reset(new lsm2d_srrg::DeviceCloud);
      }
      if (!refreshed.count(cloud_)) { // contents may have changed in place since the last call: always refill (pinned copy)
        slot->upload(_ctx, *cloud_, who);
        refreshed.insert(cloud_);
      }
      return slot->set();
    };

    for (size_t s = 0; s < param_slice_processors.size(); ++s) {
      auto processor = param_slice_processors.value(s);
      auto laser     = std::dynamic_pointer_cast<AlignerSliceProcessorLaser2D>(processor);
      auto laser_ws  = std::dynamic_pointer_cast<AlignerSliceProcessorLaser2DWithSensor>(processor);
      if (laser || laser_ws) {
        lsm2d_slice_params sp{};
        auto fill = [&](const auto& slice_) {
          PointNormal2fVectorCloud* fixed  = lsm2d_srrg::cloudInContainer(_fixed, slice_->param_fixed_slice_name.value());
          PointNormal2fVectorCloud* moving = lsm2d_srrg::cloudInContainer(_moving, slice_->param_moving_slice_name.value());
          if (!fixed || !moving) {
            throw std::runtime_error(std::string(who) + "| slice '" + slice_->param_fixed_slice_name.value() + "' / '" +
                                     slice_->param_moving_slice_name.value() + "' not found in the fixed / moving scene");
          }
          lsm2d_srrg::fillFinderParams(slice_->param_finder.value(), &sp, who);
          lsm2d_srrg::fillRobustifier(slice_->param_robustifier.value(), &sp, who);
          sp.min_num_correspondences = slice_->param_min_num_correspondences.value();
          fixed_sets.push_back(deviceCloud(fixed));
          moving_sets.push_back(deviceCloud(moving));
          lsm2d_srrg::publishSliceBinding(slice_, fixed, moving);
          laser_slices.push_back(PublishTarget{[slice_]() -> CorrespondenceVector& { return lsm2d_srrg::sliceCorrespondences(slice_); }, moving->size()});
        };
        if (laser_ws) {
          fill(laser_ws);
          // WithSensor: the estimate lives in the robot frame, the fixed scan in the sensor frame
          // (registration/aligner_slice_processor_laser_2d_impl.cpp:7-10)
          lsm2d_srrg::poseToArray(lsm2d_srrg::sensorInRobot(laser_ws), sp.sensor_in_robot);
        } else {
          fill(laser);
        }
        slices.push_back(sp);
        continue;
      }
      if (auto odom = std::dynamic_pointer_cast<AlignerSliceOdom2DPrior>(processor)) {
        // the odometry cue (MULTI.json:402-422): both scenes carry the robot's odometry pose under the slice names; the factor's
        // measurement is the moving scene's origin seen from the fixed one, Z = odom_fixed^-1 * odom_moving, its error e = t2v(Z^-1 X)
        if (has_prior) {
          throw std::runtime_error(std::string(who) + "| more than one prior slice: the device path takes one");
        }
        if (odom->param_robustifier.value()) {
          throw std::runtime_error(std::string(who) + "| a robustifier on the odometry prior slice is not supported on the device");
        }
        Isometry2f odom_fixed, odom_moving;
        if (!lsm2d_srrg::isometryInContainer(_fixed, odom->param_fixed_slice_name.value(), &odom_fixed) ||
            !lsm2d_srrg::isometryInContainer(_moving, odom->param_moving_slice_name.value(), &odom_moving)) {
          throw std::runtime_error(std::string(who) + "| odometry slice '" + odom->param_fixed_slice_name.value() + "' not found in the fixed / moving scene");
        }
        lsm2d_srrg::poseToArray(odom_fixed.inverse() * odom_moving, prior.z);
        lsm2d_srrg::priorInformation(prior.omega);
        has_prior = true;
        continue;
      }
      throw std::runtime_error(std::string(who) + "| slice processor " + std::to_string(s) +
                               " is neither a laser slice nor the odometry prior: refusing to run without it");
    }
    if (slices.empty()) {
      throw std::runtime_error(std::string(who) + "| no laser slice");
    }
    // device clouds of host clouds this call did not see are released (a tracker hands over a fresh measurement cloud object
    // now and then: the cache must not grow with them)
    for (auto it = _device_clouds.begin(); it != _device_clouds.end();) {
      it = refreshed.count(it->first) ? std::next(it) : _device_clouds.erase(it);
    }

    float pose[3], information[9];
    lsm2d_srrg::poseToArray(movingInFixed(), pose);
    int32_t status = 0, iterations = 0;
    lsm2d_batch batch{};
    batch.n_alignments = 1;
    batch.n_slices     = (int32_t) slices.size();
    batch.slices       = slices.data();
    batch.fixed        = fixed_sets.data();
    batch.moving       = moving_sets.data();
    batch.init_pose    = pose;
    batch.prior        = has_prior ? &prior : nullptr;
    lsm2d_aligner_params ap{};
    ap.max_iterations  = param_max_iterations.value();
    ap.min_num_inliers = param_min_num_inliers.value();
    ap.damping         = 0.f; // GN, MULTI.json:254-259
    ap.termination_chi_epsilon          = chi_epsilon;
    ap.enable_inlier_only_runs          = lsm2d_srrg::inlierOnlyRunsEnabled(*this) ? 1 : 0;
    ap.keep_only_inlier_correspondences = lsm2d_srrg::keepOnlyInlierCorrespondences(*this) ? 1 : 0;
    std::vector<lsm2d_iteration_stats> stats((size_t) lsm2d_stats_capacity(&ap));
    // with publish_correspondences the call also hands back what the reference leaves in slice->correspondences(): the pairs of the last
    // iteration (only its inliers under keep_only_inlier_correspondences)
    std::vector<lsm2d_correspondence> pairs;
    std::vector<int32_t> n_pairs(slices.size(), 0);
    size_t pair_capacity = 0;
    if (param_publish_correspondences.value()) {
      for (size_t s = 0; s < slices.size(); ++s) {
        const size_t need = slices[s].finder == LSM2D_FINDER_PROJECTIVE ? (size_t) slices[s].projector.canvas_cols : laser_slices[s].n_moving;
        pair_capacity     = need > pair_capacity ? need : pair_capacity;
      }
      pair_capacity = pair_capacity > 0 ? pair_capacity : 1;
      pairs.resize(pair_capacity * slices.size());
    }
    throwOnError(lsm2d_align_batch_pairs(_ctx, &ap, &batch, pose, information, &status, &iterations, stats.data(),
                                         pairs.empty() ? nullptr : pairs.data(), (int32_t) pair_capacity, n_pairs.data()),
                 who, _ctx);
    _last_status     = status;
    _last_iterations = iterations;
    setMovingInFixed(geometry2d::v2t(Vector3f(pose[0], pose[1], pose[2])));
    _writeBack(status, information, iterations, stats);

    if (param_publish_correspondences.value()) {
      for (size_t s = 0; s < laser_slices.size(); ++s) {
        CorrespondenceVector& out          = laser_slices[s].correspondences();
        const lsm2d_correspondence* mine   = pairs.data() + s * pair_capacity;
        out.resize((size_t) n_pairs[s]);
        for (int32_t i = 0; i < n_pairs[s]; ++i) {
          out[i] = Correspondence(mine[i].fixed_idx, mine[i].moving_idx);
        }
      }
    }
  }

} // namespace srrg2_laser_slam_2d
