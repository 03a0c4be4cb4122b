// SRRG-side adapter for the aligner (same caveat as correspondence_finder_hip_2d.h: needs the srrg2 stack).
//
// MultiAlignerHIP2D derives from the upstream srrg2_slam_interfaces::MultiAligner2D and overrides compute():
// instead of {finder->compute(); solver->compute()} x max_iterations on the host
// (driver apps/visual_test_aligner_2d.cpp:123-156) it makes ONE lsm2d_align_batch call that runs the
// whole loop on the device.  It consumes the same configuration: max_iterations, min_num_inliers, slice_processors
// (configurations/stage_segway_double_config_MULTI.json:700-732); laser slices
// (AlignerSliceProcessorLaser2D[WithSensor]) become lsm2d slices, an AlignerSliceOdom2DPrior becomes the lsm2d_prior.
#pragma once
#include <lsm2d.h>
#include <srrg2_laser_slam_2d/registration/aligner_slice_processor_laser_2d.h>
#include <srrg2_slam_interfaces/registration/aligners/multi_aligner.h>

namespace srrg2_laser_slam_2d {

  class MultiAlignerHIP2D : public srrg2_slam_interfaces::MultiAligner2D {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    using BaseType = srrg2_slam_interfaces::MultiAligner2D;
    PARAM(srrg2_core::PropertyInt, device_id, "HIP device ordinal", 0, 0);
    virtual ~MultiAlignerHIP2D();
    void compute() override;

  protected:
    lsm2d_context* _ctx = nullptr;
  };

  using MultiAlignerHIP2DPtr = std::shared_ptr<MultiAlignerHIP2D>;
} // namespace srrg2_laser_slam_2d
