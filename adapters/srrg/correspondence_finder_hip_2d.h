// SRRG-side adapter (compiled ONLY inside a catkin workspace that has srrg2_core / srrg2_slam_interfaces;
// it cannot be compiled in the build container of this repository -- see INTEGRATION.md).
//
// CorrespondenceFinderHIP2D is a drop-in sibling of CorrespondenceFinderProjective2f
// (srrg2_laser_slam_2d/src/srrg2_laser_slam_2d/registration/correspondence_finder_projective_2d.h:9-37):
// same base class (registration/correspondence_finder_normal_2f.h:9-13), same PARAMs, same compute() contract;
// the body forwards to the C ABI (include/lsm2d.h) instead of walking two projector canvases on the CPU.
#pragma once
#include <lsm2d.h>
#include <srrg2_laser_slam_2d/registration/correspondence_finder_normal_2f.h>
#include <srrg_config/property_configurable.h>
#include <srrg_pcl/point_projector_types.h>

namespace srrg2_laser_slam_2d {

  class CorrespondenceFinderHIP2D : public CorrespondenceFinderNormal2f {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    using BaseType = CorrespondenceFinderNormal2f;
    using ThisType = CorrespondenceFinderHIP2D;

    PARAM(srrg2_core::PropertyFloat, point_distance, "max distance between corresponding points", 0.5, 0);
    PARAM(srrg2_core::PropertyFloat, normal_cos, "min cosinus between normals", 0.8, 0);
    PARAM(srrg2_core::PropertyConfigurable_<srrg2_core::PointNormal2fProjectorPolar>,
          projector,
          "projector whose parameters (canvas_cols, angle_col_min/max, range_min/max) define the polar canvas",
          srrg2_core::PointNormal2fProjectorPolarPtr(new srrg2_core::PointNormal2fProjectorPolar),
          &_projector_changed_flag);
    PARAM(srrg2_core::PropertyInt, device_id, "HIP device ordinal", 0, 0);

    CorrespondenceFinderHIP2D();
    virtual ~CorrespondenceFinderHIP2D();
    void compute() override;

  protected:
    void _upload(const srrg2_core::PointNormal2fVectorCloud& cloud_, lsm2d_cloudset*& set_);
    bool _projector_changed_flag = true;
    lsm2d_context* _ctx          = nullptr;
    lsm2d_cloudset* _fixed_set   = nullptr;
    lsm2d_cloudset* _moving_set  = nullptr;
    const void* _moving_uploaded = nullptr; // identity + size of the cloud last uploaded
    size_t _moving_uploaded_size = 0;
    std::vector<float> _staging;
    std::vector<lsm2d_correspondence> _pairs;
  };

  using CorrespondenceFinderHIP2DPtr = std::shared_ptr<CorrespondenceFinderHIP2D>;
} // namespace srrg2_laser_slam_2d
