// upstream_access.h -- every place where the SRRG-side adapters depend on an srrg2_core / srrg2_slam_interfaces name that the
// reference tree itself does not show.  The reference uses, in-tree, only a thin slice of the upstream API
// (apps/visual_test_aligner_2d.cpp:102-156, registration/correspondence_finder_projective_2d.cpp:18-77); whatever the adapters
// need beyond that is funnelled through the accessors below and tagged UPSTREAM, so a maintainer holding the real stack has ONE
// header to adjust.  tests/cpp/adapter_shim/ implements exactly these names for the compile-and-run check in this repository.
#pragma once
#include <lsm2d.h>
#include <srrg2_slam_interfaces/registration/aligners/multi_aligner.h>
#include <srrg_config/property_configurable.h>
#include <srrg_geometry/geometry2d.h>
#include <srrg_pcl/point_types.h>

#include <stdexcept>
#include <string>

namespace lsm2d_srrg {
  using namespace srrg2_core;

  // a value bound to a slice name in one of the aligner's property containers: the reference builds them as
  // Property_<PointNormal2fVectorCloud*>("points", "", &container) (apps/visual_test_aligner_2d.cpp:108-118).
  // UPSTREAM: PropertyContainerBase::property(name) returning the PropertyBase* registered under `name`.
  template <typename T>
  inline bool valueInContainer(PropertyContainerBase* container_, const std::string& name_, T* out_) {
    if (!container_) {
      return false;
    }
    auto* p = dynamic_cast<Property_<T>*>(container_->property(name_));
    if (!p) {
      return false;
    }
    *out_ = p->value();
    return true;
  }
  inline PointNormal2fVectorCloud* cloudInContainer(PropertyContainerBase* container_, const std::string& name_) {
    PointNormal2fVectorCloud* cloud = nullptr;
    return valueInContainer<PointNormal2fVectorCloud*>(container_, name_, &cloud) ? cloud : nullptr;
  }
  // the odometry poses an AlignerSliceOdom2DPrior reads ("odom" in both containers, MULTI.json:402-422).
  // UPSTREAM: stored as Property_<Isometry2f> (by value) -- a pointer-valued property is tried as well.
  inline bool isometryInContainer(PropertyContainerBase* container_, const std::string& name_, Isometry2f* out_) {
    if (valueInContainer<Isometry2f>(container_, name_, out_)) {
      return true;
    }
    Isometry2f* ptr = nullptr;
    if (valueInContainer<Isometry2f*>(container_, name_, &ptr) && ptr) {
      *out_ = *ptr;
      return true;
    }
    return false;
  }
  // WithSensor slices: the sensor pose in the robot frame, which upstream reads from the tf Platform by base_frame_id / frame_id
  // (registration/aligner_slice_processor_laser_2d_impl.cpp:7-10, apps/visual_test_aligner_2d.cpp:96-107).
  // UPSTREAM: accessor name on the slice after it has been bound to the platform.
  template <typename SlicePtr_>
  inline Isometry2f sensorInRobot(const SlicePtr_& slice_) {
    return slice_->sensorInRobot();
  }
  // the slice's view of its clouds and pairs that callers read after compute() (apps/visual_test_aligner_2d.cpp:129-143:
  // slice->fixed(), slice->moving(), slice->correspondences()).  UPSTREAM: the member names behind those accessors.
  template <typename SlicePtr_>
  inline void publishSliceBinding(const SlicePtr_& slice_, PointNormal2fVectorCloud* fixed_, PointNormal2fVectorCloud* moving_) {
    slice_->_fixed_slice  = fixed_;
    slice_->_moving_slice = moving_;
  }
  template <typename SlicePtr_>
  inline CorrespondenceVector& sliceCorrespondences(const SlicePtr_& slice_) {
    return slice_->_correspondences;
  }
  // aligner options the shipped configurations carry (MULTI.json:606-610,627-630,704-708,729-731), handed on to the device loop
  // (lsm2d_aligner_params.enable_inlier_only_runs / keep_only_inlier_correspondences / termination_chi_epsilon).
  // UPSTREAM: the PARAM types behind "enable_inlier_only_runs" / "keep_only_inlier_correspondences" (bool or int) and the class behind
  // "termination_criteria" -- only `.value()`, a conversion to bool and a property looked up by name are used.
  template <typename Aligner_>
  inline bool inlierOnlyRunsEnabled(const Aligner_& aligner_) {
    return (bool) aligner_.param_enable_inlier_only_runs.value();
  }
  template <typename Aligner_>
  inline bool keepOnlyInlierCorrespondences(const Aligner_& aligner_) {
    return (bool) aligner_.param_keep_only_inlier_correspondences.value();
  }
  template <typename Aligner_>
  inline bool terminationCriteriaSet(const Aligner_& aligner_) {
    return (bool) aligner_.param_termination_criteria.value();
  }
  // The criteria OBJECT as an epsilon.  The aligner-side class is not in the reference tree; the one in-tree trace of such a criterion is the
  // solver's SimpleTerminationCriteria with its float property "epsilon" -- "ratio of decay of chi2 between iteration" (MULTI.json:218-223) --
  // which is exactly what lsm2d_aligner_params.termination_chi_epsilon implements.  A criteria object that carries a float property of that
  // name is translated; any other object returns a negative value and MultiAlignerHIP2D refuses it (it cannot know what it asks for).
  // UPSTREAM: Configurable::property(name) and the PropertyFloat type behind it.
  template <typename Aligner_>
  inline float terminationCriteriaEpsilon(const Aligner_& aligner_) {
    auto criteria = aligner_.param_termination_criteria.value();
    if (!criteria) {
      return 0.f;
    }
    auto* eps = dynamic_cast<srrg2_core::PropertyFloat*>(criteria->property("epsilon"));
    return eps ? eps->value() : -1.f;
  }
  // information matrix of the odometry prior factor.  UPSTREAM: the prior slice's factor carries its own information matrix; the
  // shipped configuration sets none (MULTI.json:402-422), i.e. the factor's default, taken to be identity.
  inline void priorInformation(float omega_row_major_[9]) {
    for (int i = 0; i < 9; ++i) {
      omega_row_major_[i] = (i % 4 == 0) ? 1.f : 0.f;
    }
  }
} // namespace lsm2d_srrg
