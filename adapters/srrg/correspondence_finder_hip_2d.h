// SRRG-side adapters for plugin interface #1: drop-in siblings of the three finders the reference
// registers (src/srrg2_laser_slam_2d/instances.cpp:27-29).  Same base class (registration/correspondence_finder_normal_2f.h:9-13),
// same PARAM names and defaults, same compute() contract; the body forwards to the C ABI (include/lsm2d.h).
//
//   CorrespondenceFinderHIP2D        sibling of CorrespondenceFinderProjective2f  (registration/correspondence_finder_projective_2d.h:9-37)
//   CorrespondenceFinderKDTreeHIP2D  sibling of CorrespondenceFinderKDTree2D      (registration/correspondence_finder_kd_tree_2d.h:12-47)
//   CorrespondenceFinderNNHIP2D      sibling of CorrespondenceFinderNN2D          (registration/correspondence_finder_nn_2d.h:11-67)
//
// Compiled inside a catkin workspace that has the srrg2 stack; in THIS repository the same sources are compiled and driven on the
// GPU against stand-in headers (tests/cpp/adapter_shim, tests/cpp/adapter_driver.cpp).
#pragma once
#include "lsm2d_srrg_common.h"

#include <srrg2_laser_slam_2d/registration/correspondence_finder_normal_2f.h>

namespace srrg2_laser_slam_2d {

  // what the three siblings share: the device context, the two device clouds, the call
  class CorrespondenceFinderHIPBase : public CorrespondenceFinderNormal2f, public lsm2d_srrg::SliceParamsSource {
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    using BaseType = CorrespondenceFinderNormal2f;
    PARAM(srrg2_core::PropertyInt, device_id, "HIP device ordinal", 0, 0);
    virtual ~CorrespondenceFinderHIPBase();
    void compute() override;

  protected:
    virtual const char* className() const = 0;
    lsm2d_context* _ctx = nullptr;
    lsm2d_srrg::DeviceCloud _fixed_dev, _moving_dev;
    std::vector<lsm2d_correspondence> _pairs;
  };

  class CorrespondenceFinderHIP2D : public CorrespondenceFinderHIPBase {
  public:
    PARAM(srrg2_core::PropertyFloat, point_distance, "max distance between corresponding points", 0.5, 0);
    PARAM(srrg2_core::PropertyFloat, normal_cos, "min cosinus between normals", 0.8, 0);
    PARAM(srrg2_core::PropertyConfigurable_<srrg2_core::PointNormal2fProjectorPolar>,
          projector,
          "projector whose parameters (canvas_cols, angle_col_min/max, range_min/max) define the polar canvas",
          srrg2_core::PointNormal2fProjectorPolarPtr(new srrg2_core::PointNormal2fProjectorPolar),
          0);
    void fillSliceParams(lsm2d_slice_params* sp_) const override;

  protected:
    const char* className() const override {
      return "CorrespondenceFinderHIP2D";
    }
  };

  class CorrespondenceFinderKDTreeHIP2D : public CorrespondenceFinderHIPBase {
  public:
    PARAM(srrg2_core::PropertyFloat, max_distance_m, "max distance for correspondences [meters]", 1e-2, 0);
    // kept so that a configuration written for CorrespondenceFinderKDTree2D loads unchanged; the device search is an exact
    // nearest-neighbour search on a uniform grid, for which they have no meaning (PARITY.md section 5 measures the difference)
    PARAM(srrg2_core::PropertyFloat, max_leaf_range, "unused by the device search (exact NN)", 1e-2, 0);
    PARAM(srrg2_core::PropertyUnsignedInt, min_leaf_points, "unused by the device search (exact NN)", 20, 0);
    PARAM(srrg2_core::PropertyFloat, normal_cos, "min cosinus between normals", 0.8, 0);
    void fillSliceParams(lsm2d_slice_params* sp_) const override;

  protected:
    const char* className() const override {
      return "CorrespondenceFinderKDTreeHIP2D";
    }
  };

  class CorrespondenceFinderNNHIP2D : public CorrespondenceFinderHIPBase {
  public:
    PARAM(srrg2_core::PropertyFloat, max_distance_m, "max distance for correspondences [meters]", 1, 0);
    PARAM(srrg2_core::PropertyFloat, resolution, "resolution of the distance map [m/pixel]", 0.05, 0);
    PARAM(srrg2_core::PropertyFloat, normal_cos, "min cosinus between normals", 0.8, 0);
    void fillSliceParams(lsm2d_slice_params* sp_) const override;

  protected:
    const char* className() const override {
      return "CorrespondenceFinderNNHIP2D";
    }
  };

  using CorrespondenceFinderHIP2DPtr       = std::shared_ptr<CorrespondenceFinderHIP2D>;
  using CorrespondenceFinderKDTreeHIP2DPtr = std::shared_ptr<CorrespondenceFinderKDTreeHIP2D>;
  using CorrespondenceFinderNNHIP2DPtr     = std::shared_ptr<CorrespondenceFinderNNHIP2D>;
} // namespace srrg2_laser_slam_2d
