// SRRG-side adapter for plugin interface #2: MultiAlignerHIP2D derives from the upstream
// srrg2_slam_interfaces::MultiAligner2D and overrides compute(): instead of {finder->compute(); solver->compute()} x
// max_iterations on the host (driver apps/visual_test_aligner_2d.cpp:123-156) it makes ONE lsm2d_align_batch
// call that runs the whole loop on the device.  It consumes the same configuration -- max_iterations, min_num_inliers,
// slice_processors (configurations/stage_segway_double_config_MULTI.json:700-732): every AlignerSliceProcessorLaser2D[WithSensor]
// becomes an lsm2d slice (finder, robustifier, min_num_correspondences, sensor extrinsics), the AlignerSliceOdom2DPrior
// (MULTI.json:402-422) becomes the lsm2d_prior; a slice processor of any other type is an ERROR, never skipped.  enable_inlier_only_runs /
// keep_only_inlier_correspondences (MULTI.json:606-610) go to the device loop as they are, a termination_criteria object (:627-630) is
// translated into the epsilon it carries (one that carries none is refused).  After the call
// the pose, the status, the information matrix and the iteration statistics are written back into the base class.
// Device clouds persist across calls (reserved sets, refilled), one per distinct host cloud.
#pragma once
#include "lsm2d_srrg_common.h"

#include <srrg2_laser_slam_2d/registration/aligner_slice_processor_laser_2d.h>
#include <srrg2_slam_interfaces/registration/aligners/multi_aligner.h>

#include <map>

namespace srrg2_laser_slam_2d {

  class MultiAlignerHIP2D : public srrg2_slam_interfaces::MultiAligner2D {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    using BaseType = srrg2_slam_interfaces::MultiAligner2D;
    PARAM(srrg2_core::PropertyInt, device_id, "HIP device ordinal", 0, 0);
    PARAM(srrg2_core::PropertyInt,
          publish_correspondences,
          "1: after compute() every laser slice's correspondences() holds the pairs of the last iteration, as the reference leaves them (only "
          "its inliers with keep_only_inlier_correspondences; one extra finder pass per slice; callers use them for drawing only: "
          "apps/visual_test_aligner_2d.cpp:129-143)",
          0,
          0);
    PARAM(srrg2_core::PropertyFloat,
          termination_chi_epsilon,
          "device-side termination criterion: stop after an iteration whose total chi2 differs from the previous one's by less than this "
          "ratio (0: run max_iterations, what an unset termination_criteria means; a termination_criteria OBJECT with a float property "
          "'epsilon' is translated into this value)",
          0.f,
          0);
    virtual ~MultiAlignerHIP2D();
    void compute() override;
    // per-alignment outcome of the last compute(), in the C ABI's terms (lsm2d_status >= 0)
    int lastDeviceStatus() const {
      return _last_status;
    }
    int lastIterations() const {
      return _last_iterations;
    }

  protected:
    void _writeBack(int status_, const float information_[9], int iterations_, const std::vector<lsm2d_iteration_stats>& stats_);
    lsm2d_context* _ctx = nullptr;
    // one device cloud per host cloud object (two laser slices usually share the moving cloud: MULTI.json:372-400,160-188)
    std::map<const srrg2_core::PointNormal2fVectorCloud*, std::unique_ptr<lsm2d_srrg::DeviceCloud>> _device_clouds;
    int _last_status = -1, _last_iterations = 0;
  };

  using MultiAlignerHIP2DPtr = std::shared_ptr<MultiAlignerHIP2D>;
} // namespace srrg2_laser_slam_2d
