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
    // unset (the default): the ONE context every HIP module of this process shares on that device (lsm2d_srrg::sharedContext); set: the context of that
    // configurable -- the finder siblings of one aligner share a stream and their staging buffers either way
    PARAM(srrg2_core::PropertyConfigurable_<lsm2d_srrg::HipContext>, context, "device context shared with other HIP modules (unset: one per process and device)", nullptr, 0);
    virtual ~CorrespondenceFinderHIPBase();
    // host-to-device cloud uploads the finder's context has queued so far (lsm2d option "uploads"): what the upload-once test of the adapter driver reads
    int64_t contextUploads() const;
    void compute() override;

  protected:
    virtual const char* className() const = 0;
    lsm2d_context* _ctx = nullptr;                               // borrowed: _shared or param_context own it
    std::shared_ptr<lsm2d_srrg::SharedContext> _shared;
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
    PARAM(srrg2_core::PropertyFloat, max_leaf_range, "maximum range for a leaf of the KDTree [meters]", 1e-2, 0);
    PARAM(srrg2_core::PropertyUnsignedInt, min_leaf_points, "minimum number of points in a leaf of the KDTree", 20, 0);
    PARAM(srrg2_core::PropertyFloat, normal_cos, "min cosinus between normals", 0.8, 0);
    // "kdtree" (default): the reference's own structure on the device -- KDTree2D(coordinates, max_leaf_range, min_leaf_points) built in
    // reset() (registration/correspondence_finder_kd_tree_2d.cpp:31-38) and findNeighbor's single-leaf descent (.cpp:18-19), restated as
    // SURVEY.md App. A.4 believes upstream implements them: a configuration written for CorrespondenceFinderKDTree2D keeps its meaning.
    // "exact": an exact nearest-neighbour search on a uniform grid (never a farther neighbour than the tree's; the two leaf
    // parameters are not used by it).
    PARAM(srrg2_core::PropertyString, search, "device search: kdtree (the reference's tree, approximate) | exact (uniform grid)", "kdtree", 0);
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
