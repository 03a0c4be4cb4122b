// srrg_shim.h -- TEST INFRASTRUCTURE ONLY, never shipped, never part of the product.
//
// Minimal stand-ins for the parts of the srrg2 stack (srrg2_core, srrg2_solver, srrg2_slam_interfaces -- absent from this image
// and from the GPU box) and of the reference package's own class declarations that the SRRG-side adapter sources under
// adapters/srrg/ are written against.  They exist for ONE purpose: so that those translation units are compiled (here, by g++) and
// driven on the GPU (tests/cpp/adapter_driver.cpp) instead of being un-compiled text.  Only what the adapters touch is declared,
// with the names the reference itself uses in-tree:
//   base-class members of a finder     registration/correspondence_finder_projective_2d.cpp:18-77, ..._kd_tree_2d.cpp:5-29
//   PARAM / property accessors          registration/correspondence_finder_kd_tree_2d.h:23-34, ..._nn_2d.h:20-30
//   aligner / slice driving surface     apps/visual_test_aligner_2d.cpp:102-156
//   property containers                 apps/visual_test_aligner_2d.cpp:108-118
// Names that could NOT be confirmed from the reference tree are the ones adapters/srrg/upstream_access.h isolates; the shim
// implements exactly those accessors, so a maintainer with the real stack adjusts one header.
// Nothing here restates reference ALGORITHMS: every compute() of a shim base class is empty or aborts.
#pragma once
#include <cmath>
#include <cstddef>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define PARAM(TYPE, NAME, DESC, DEFAULT, FLAG) TYPE param_##NAME = TYPE(#NAME, DESC, this, DEFAULT, FLAG)
#define PARAM_VECTOR(TYPE, NAME, DESC, FLAG) TYPE param_##NAME = TYPE(#NAME, DESC, this, FLAG)

namespace srrg2_core {
  // ---- the handful of Eigen types / operations the adapters use.  The stand-ins IMITATE EIGEN'S PUBLIC API (accessors x() / y() / z(),
  // operator()(row, col), the comma initialiser `m << a, b, c, d`, Identity(), setZero(), Isometry2f::linear() / translation() /
  // inverse() / operator*) and keep their storage PRIVATE (names ending in __shim), so an adapter that compiles here uses nothing Eigen
  // does not have: tests/test_abi.py::test_adapters_use_no_shim_only_member greps for it as well.
  template <int N_>
  class ShimCommaInitializer_ {      // what `matrix << a, b, ...` returns (Eigen::CommaInitializer): row-major fill
  public:
    ShimCommaInitializer_(float* data_, float first_) : _d__shim(data_) { _d__shim[_k__shim++] = first_; }
    template <typename S_> ShimCommaInitializer_& operator,(S_ v_) { if (_k__shim < N_) _d__shim[_k__shim++] = (float) v_; return *this; }
  private:
    float* _d__shim; int _k__shim = 0;
  };
  class Vector2f {
  public:
    Vector2f() {}
    Vector2f(float x_, float y_) { _v__shim[0] = x_; _v__shim[1] = y_; }
    float& x() { return _v__shim[0]; } float& y() { return _v__shim[1]; }
    const float& x() const { return _v__shim[0]; } const float& y() const { return _v__shim[1]; }
    float& operator()(int i_) { return _v__shim[i_]; } const float& operator()(int i_) const { return _v__shim[i_]; }
  private:
    float _v__shim[2] = {0, 0};
  };
  class Vector3f {
  public:
    Vector3f() {}
    Vector3f(float x_, float y_, float z_) { _v__shim[0] = x_; _v__shim[1] = y_; _v__shim[2] = z_; }
    float& x() { return _v__shim[0]; } float& y() { return _v__shim[1]; } float& z() { return _v__shim[2]; }
    const float& x() const { return _v__shim[0]; } const float& y() const { return _v__shim[1]; } const float& z() const { return _v__shim[2]; }
    float& operator()(int i_) { return _v__shim[i_]; } const float& operator()(int i_) const { return _v__shim[i_]; }
  private:
    float _v__shim[3] = {0, 0, 0};
  };
  template <int R_>
  class ShimSquareMatrix_ {           // Matrix2f / Matrix3f: row-major storage behind Eigen's accessors
  public:
    float& operator()(int r_, int c_) { return _m__shim[R_ * r_ + c_]; }
    const float& operator()(int r_, int c_) const { return _m__shim[R_ * r_ + c_]; }
    void setIdentity() { for (int i = 0; i < R_ * R_; ++i) _m__shim[i] = (i % (R_ + 1) == 0) ? 1.f : 0.f; }
    void setZero() { for (int i = 0; i < R_ * R_; ++i) _m__shim[i] = 0.f; }
    static ShimSquareMatrix_ Identity() { ShimSquareMatrix_ r; r.setIdentity(); return r; }
    static ShimSquareMatrix_ Zero() { return ShimSquareMatrix_(); }
    template <typename S_> ShimCommaInitializer_<R_ * R_> operator<<(S_ first_) { return ShimCommaInitializer_<R_ * R_>(_m__shim, (float) first_); }
  private:
    float _m__shim[R_ * R_] = {};
  };
  using Matrix2f = ShimSquareMatrix_<2>;
  using Matrix3f = ShimSquareMatrix_<3>;
  class Isometry2f {       // Eigen::Transform<float, 2, Isometry>: R = linear(), t = translation()
  public:
    static Isometry2f Identity() { return Isometry2f(); }
    void setIdentity() { *this = Isometry2f(); }
    Matrix2f linear() const { Matrix2f r; r(0, 0) = _c__shim; r(0, 1) = -_s__shim; r(1, 0) = _s__shim; r(1, 1) = _c__shim; return r; }
    Vector2f translation() const { return Vector2f(_tx__shim, _ty__shim); }
    Isometry2f inverse() const {
      Isometry2f r; r._c__shim = _c__shim; r._s__shim = -_s__shim;
      r._tx__shim = -(_c__shim * _tx__shim + _s__shim * _ty__shim); r._ty__shim = -(-_s__shim * _tx__shim + _c__shim * _ty__shim); return r;
    }
    Isometry2f operator*(const Isometry2f& o) const {
      Isometry2f r; r._c__shim = _c__shim * o._c__shim - _s__shim * o._s__shim; r._s__shim = _s__shim * o._c__shim + _c__shim * o._s__shim;
      r._tx__shim = _c__shim * o._tx__shim - _s__shim * o._ty__shim + _tx__shim; r._ty__shim = _s__shim * o._tx__shim + _c__shim * o._ty__shim + _ty__shim; return r;
    }
  private:
    float _c__shim = 1.f, _s__shim = 0.f, _tx__shim = 0.f, _ty__shim = 0.f;
    friend Isometry2f shimMakeIsometry(float c_, float s_, float tx_, float ty_);
  };
  inline Isometry2f shimMakeIsometry(float c_, float s_, float tx_, float ty_) { Isometry2f T; T._c__shim = c_; T._s__shim = s_; T._tx__shim = tx_; T._ty__shim = ty_; return T; }
  namespace geometry2d {      // srrg_geometry/geometry2d.h: v2t / t2v as the reference calls them (apps/visual_test_aligner_2d.cpp:126,145)
    inline Vector3f t2v(const Isometry2f& T) { const Matrix2f R = T.linear(); const Vector2f t = T.translation(); return Vector3f(t.x(), t.y(), std::atan2(R(1, 0), R(0, 0))); }
    inline Isometry2f v2t(const Vector3f& v) { return shimMakeIsometry(std::cos(v.z()), std::sin(v.z()), v.x(), v.y()); }
  } // namespace geometry2d

  // ---- point cloud types (srrg_pcl)
  class PointNormal2f {
  public:
    Vector2f& coordinates() { return _c__shim; } const Vector2f& coordinates() const { return _c__shim; }
    Vector2f& normal() { return _n__shim; } const Vector2f& normal() const { return _n__shim; }
  private:
    Vector2f _c__shim, _n__shim;
  };
  using PointNormal2fVectorCloud = std::vector<PointNormal2f>;
  struct Correspondence {
    int fixed_idx = -1, moving_idx = -1; float response = 0.f;
    Correspondence() {}
    Correspondence(int f_, int m_, float r_ = 0.f) : fixed_idx(f_), moving_idx(m_), response(r_) {}
  };
  using CorrespondenceVector = std::vector<Correspondence>;

  // ---- properties / configurables (srrg_property, srrg_config)
  class PropertyBase { public: virtual ~PropertyBase() {} };
  // containers holding named properties: the dynamic ones of apps/visual_test_aligner_2d.cpp:108-118, and every Configurable (a PARAM
  // registers itself with its owner under its name: what ConfigurableManager reads and writes a configuration file through)
  class PropertyContainerBase {
  public:
    virtual ~PropertyContainerBase() {}
    std::map<std::string, PropertyBase*> _props;
    PropertyBase* property(const std::string& name_) const { auto it = _props.find(name_); return it == _props.end() ? nullptr : it->second; }
  };
  using PropertyContainerDynamic = PropertyContainerBase;
  class Configurable : public PropertyContainerBase { public: virtual ~Configurable() {} };
  template <typename T>
  class Property_ : public PropertyBase {
  public:
    Property_(const char* name_, const char*, void*, const T& def_, bool* flag_ = nullptr) : _name(name_), _value(def_), _flag(flag_) {}
    template <typename Owner_, typename = typename std::enable_if<std::is_base_of<PropertyContainerBase, Owner_>::value>::type>
    Property_(const char* name_, const char*, Owner_* owner_, const T& def_, bool* flag_ = nullptr) : _name(name_), _value(def_), _flag(flag_) {
      if (owner_) static_cast<PropertyContainerBase*>(owner_)->_props[_name] = this;
    }
    Property_(const std::string& name_, const std::string&, class PropertyContainerBase* owner_);
    const T& value() const { return _value; }
    T& value() { return _value; }
    void setValue(const T& v_) { _value = v_; if (_flag) *_flag = true; }
    const std::string& name() const { return _name; }
  protected:
    std::string _name; T _value{}; bool* _flag = nullptr;
  };
  using PropertyFloat = Property_<float>;
  using PropertyInt = Property_<int>;
  using PropertyBool = Property_<bool>;
  using PropertyUnsignedInt = Property_<unsigned int>;
  using PropertyString = Property_<std::string>;
  template <typename C>
  class PropertyConfigurable_ {
  public:
    PropertyConfigurable_(const char*, const char*, void*, std::shared_ptr<C> def_, bool* flag_ = nullptr) : _value(def_), _flag(flag_) {}
    std::shared_ptr<C> value() const { return _value; }
    C* operator->() const { return _value.get(); }
    template <typename D> void setValue(std::shared_ptr<D> v_) { _value = v_; if (_flag) *_flag = true; }
  protected:
    std::shared_ptr<C> _value; bool* _flag = nullptr;
  };
  template <typename C>
  class PropertyConfigurableVector_ {
  public:
    PropertyConfigurableVector_(const char*, const char*, void*, bool* = nullptr) {}
    size_t size() const { return _v.size(); }
    std::shared_ptr<C> value(size_t i) const { return _v[i]; }
    template <typename D> void pushBack(std::shared_ptr<D> v_) { _v.push_back(v_); }
  protected:
    std::vector<std::shared_ptr<C>> _v;
  };
  template <typename T>
  Property_<T>::Property_(const std::string& name_, const std::string&, PropertyContainerBase* owner_) : _name(name_) { if (owner_) owner_->_props[name_] = this; }

  // ---- projector parameters (srrg_pcl/point_projector_types.h): only the PARAMs the reference sets (apps/synthetic_scene_generator.cpp:69-75)
  class PointNormal2fProjectorPolar : public Configurable {
  public:
    PARAM(PropertyInt, canvas_cols, "", 721, 0);
    PARAM(PropertyInt, canvas_rows, "", 1, 0);
    PARAM(PropertyFloat, angle_col_min, "", -3.14159f, 0);
    PARAM(PropertyFloat, angle_col_max, "", 3.14159f, 0);
    PARAM(PropertyFloat, range_min, "", 0.3f, 0);
    PARAM(PropertyFloat, range_max, "", 20.f, 0);
  };
  using PointNormal2fProjectorPolarPtr = std::shared_ptr<PointNormal2fProjectorPolar>;

  // ---- what the raw-data preprocessor touches (sensor_processing/raw_data_preprocessor_projective_2d.{h,cpp}): the laser message's fields
  // (:78-85), the un-projector's PARAMs it sets per message (:96-101), the sliding-window normal computator's PARAMs (MULTI.json:845-853)
  class BaseSensorMessage { public: virtual ~BaseSensorMessage() {} };
  using BaseSensorMessagePtr = std::shared_ptr<BaseSensorMessage>;
  class LaserMessage : public BaseSensorMessage {
  public:
    Property_<std::string> topic{"topic", "", nullptr, std::string("/scan")};
    Property_<std::vector<float>> ranges{"ranges", "", nullptr, std::vector<float>()};
    PropertyFloat range_min{"range_min", "", nullptr, 0.f}, range_max{"range_max", "", nullptr, 0.f};
    PropertyFloat angle_min{"angle_min", "", nullptr, 0.f}, angle_max{"angle_max", "", nullptr, 0.f};
  };
  using LaserMessagePtr = std::shared_ptr<LaserMessage>;
  class PointNormal2fUnprojectorPolar : public Configurable {
  public:
    PARAM(PropertyFloat, range_min, "", 0.3f, 0);
    PARAM(PropertyFloat, range_max, "", 20.f, 0);
    PARAM(PropertyFloat, angle_min, "", -3.14159f, 0);
    PARAM(PropertyFloat, angle_max, "", 3.14159f, 0);
    void setCameraMatrix(const Matrix2f& m_) { _camera_matrix = m_; }
    Matrix2f _camera_matrix;
  };
  using PointNormal2fUnprojectorPolarPtr = std::shared_ptr<PointNormal2fUnprojectorPolar>;
  template <typename Cloud_, int idx_>
  class NormalComputator1DSlidingWindow : public Configurable {
  public:
    PARAM(PropertyInt, normal_min_points, "min number of points to compute a normal", 5, 0);
    PARAM(PropertyFloat, normal_point_distance, "max normal point distance", 0.3f, 0);
  };
} // namespace srrg2_core

namespace srrg2_solver {
  class RobustifierBase : public srrg2_core::Configurable {};
  class RobustifierCauchy : public RobustifierBase {
  public:
    PARAM(srrg2_core::PropertyFloat, chi_threshold, "threshold of chi after which the kernel is active", 1.f, 0);
  };
  struct IterationStats {     // what aligner->iterationStats() prints (apps/visual_test_aligner_2d.cpp:156)
    int iteration = 0, num_inliers = 0, num_outliers = 0; float chi_inliers = 0.f, chi_outliers = 0.f;
  };
  using IterationStatsVector = std::vector<IterationStats>;
} // namespace srrg2_solver

namespace srrg2_slam_interfaces {
  using namespace srrg2_core;
  // CorrespondenceFinder_ : members as the reference's subclasses use them (registration/correspondence_finder_projective_2d.cpp:25-49)
  template <typename Est_, typename Fixed_, typename Moving_>
  class CorrespondenceFinder_ : public Configurable {
  public:
    using EstimateType = Est_; using FixedType = Fixed_; using MovingType = Moving_;
    virtual void setFixed(FixedType* f_) { _fixed = f_; _fixed_changed_flag = true; }
    virtual void setMoving(MovingType* m_) { _moving = m_; _moving_changed_flag = true; }
    void setLocalMapInSensor(const EstimateType& e_) { _local_map_in_sensor = e_; }
    void setCorrespondences(CorrespondenceVector* c_) { _correspondences = c_; }
    virtual void compute() = 0;
    virtual void reset() {}
  protected:
    FixedType* _fixed = nullptr; MovingType* _moving = nullptr; CorrespondenceVector* _correspondences = nullptr;
    EstimateType _local_map_in_sensor = EstimateType::Identity();
    bool _fixed_changed_flag = true, _moving_changed_flag = true;
  };
  class AlignerBase : public Configurable {
  public:
    enum Status { Fail = 0, NotEnoughCorrespondences = 1, NotEnoughInliers = 2, Success = 3 };
  };
  // slice processors: what apps/visual_test_aligner_2d.cpp:102-143 and MULTI.json:160-188 show
  class AlignerSliceProcessorBase : public Configurable {
  public:
    PARAM(PropertyString, fixed_slice_name, "name of the slice in the fixed scene", "", 0);
    PARAM(PropertyString, moving_slice_name, "name of the slice in the moving scene", "", 0);
    PARAM(PropertyString, base_frame_id, "", "", 0);
    PARAM(PropertyString, frame_id, "", "", 0);
    PARAM(PropertyConfigurable_<srrg2_solver::RobustifierBase>, robustifier, "robustifier used on this slice", nullptr, 0);
  };
  template <typename Fixed_, typename Moving_>
  class AlignerSliceProcessorCloud_ : public AlignerSliceProcessorBase {
  public:
    using FinderType = CorrespondenceFinder_<Isometry2f, Fixed_, Moving_>;
    PARAM(PropertyConfigurable_<FinderType>, finder, "correspondence finder used in this cue", nullptr, 0);
    PARAM(PropertyInt, min_num_correspondences, "minimum number of correspondences in this slice", 0, 0);
    Fixed_* fixed() { return _fixed_slice; } Moving_* moving() { return _moving_slice; }
    const CorrespondenceVector& correspondences() const { return _correspondences; }
    CorrespondenceVector _correspondences; Fixed_* _fixed_slice = nullptr; Moving_* _moving_slice = nullptr;
    Isometry2f _sensor_in_robot = Isometry2f::Identity();
    const Isometry2f& sensorInRobot() const { return _sensor_in_robot; }
  };
  class AlignerSliceOdom2DPrior : public AlignerSliceProcessorBase {};      // MULTI.json:402-422: fixed / moving slices "odom" hold Isometry2f
  class AlignerTerminationCriteriaBase : public Configurable {};      // what "termination_criteria" points to (MULTI.json:627-630,729-731: unset in both shipped aligners)
  class MultiAligner2D : public AlignerBase {
  public:
    PARAM(PropertyInt, max_iterations, "maximum number of iterations", 10, 0);
    PARAM(PropertyInt, min_num_inliers, "minimum number of inliers", 10, 0);
    // the aligner options the shipped configurations carry next to the two above (MULTI.json:606-610,704-708,627-630,729-731)
    PARAM(PropertyBool, enable_inlier_only_runs, "toggles additional inlier only runs if sufficient inliers are available", false, 0);
    PARAM(PropertyBool, keep_only_inlier_correspondences, "toggles removal of correspondences which factors are not inliers in the last iteration", false, 0);
    PARAM(PropertyConfigurable_<AlignerTerminationCriteriaBase>, termination_criteria, "termination criteria, not set=max iterations", nullptr, 0);
    PARAM_VECTOR(PropertyConfigurableVector_<AlignerSliceProcessorBase>, slice_processors, "slices", 0);
    virtual void setFixed(PropertyContainerBase* f_) { _fixed = f_; }
    virtual void setMoving(PropertyContainerBase* m_) { _moving = m_; }
    void setMovingInFixed(const Isometry2f& e_) { _moving_in_fixed = e_; }
    const Isometry2f& movingInFixed() const { return _moving_in_fixed; }
    virtual void compute() = 0;
    Status status() const { return _status; }
    const Matrix3f& informationMatrix() const { return _information_matrix; }
    const srrg2_solver::IterationStatsVector& iterationStats() const { return _iteration_stats; }
  protected:
    PropertyContainerBase* _fixed = nullptr; PropertyContainerBase* _moving = nullptr;
    Isometry2f _moving_in_fixed = Isometry2f::Identity();
    Status _status = Fail; Matrix3f _information_matrix; srrg2_solver::IterationStatsVector _iteration_stats;
  };
} // namespace srrg2_slam_interfaces

namespace srrg2_slam_interfaces {
  // mapping bases: members as scene_clipper_projective_2d.cpp:12-64 and merger_projective_2d.cpp:17-99 use them
  class MergerBase : public Configurable { public: enum Status { Error = 0, Success = 1 }; };
  template <typename Est_, typename Scene_, typename Meas_>
  class Merger_ : public MergerBase {
  public:
    void setScene(Scene_* s_) { _scene = s_; } void setMeasurement(Meas_* m_) { _measurement = m_; }
    void setMeasurementInScene(const Est_& e_) { _measurement_in_scene = e_; }
    virtual void compute() = 0;
    Status status() const { return _status; }
  protected:
    Scene_* _scene = nullptr; Meas_* _measurement = nullptr; Est_ _measurement_in_scene = Est_::Identity(); Status _status = Error;
  };
  // raw-data preprocessor base (srrg2_slam_interfaces/raw_data_preprocessors/raw_data_preprocessor.h): the members the reference's
  // implementation uses (raw_data_preprocessor_projective_2d.cpp:13-17,42-51,59-60: _meas, _raw_data, _status, setRawData)
  template <typename T>
  std::shared_ptr<T> extractMessage(srrg2_core::BaseSensorMessagePtr msg_, const std::string& topic_) {
    auto m = std::dynamic_pointer_cast<T>(msg_);
    return (m && m->topic.value() == topic_) ? m : nullptr;
  }
  template <typename Meas_>
  class RawDataPreprocessor_ : public Configurable {
  public:
    using MeasurementType = Meas_;
    enum Status { Error = 0, Ready = 1 };
    virtual ~RawDataPreprocessor_() {}
    virtual bool setRawData(srrg2_core::BaseSensorMessagePtr msg_) { _raw_data = msg_; return true; }
    void setMeas(Meas_* meas_) { _meas = meas_; }
    Status status() const { return _status; }
    virtual void compute() = 0;
  protected:
    Meas_* _meas = nullptr; srrg2_core::BaseSensorMessagePtr _raw_data; Status _status = Error;
  };
  template <typename Est_, typename Scene_>
  class SceneClipper_ : public Configurable {
  public:
    using EstimateType = Est_; using ThisType = SceneClipper_<Est_, Scene_>;
    enum Status { Error = 0, Successful = 1 };
    void setFullScene(Scene_* s_) { _full_scene = s_; } void setClippedSceneInRobot(Scene_* s_) { _clipped_scene_in_robot = s_; }
    void setRobotInLocalMap(const Est_& e_) { _robot_in_local_map = e_; } void setSensorInRobot(const Est_& e_) { _sensor_in_robot = e_; }
    virtual void compute() = 0;
    Status status() const { return _status; }
  protected:
    Scene_* _full_scene = nullptr; Scene_* _clipped_scene_in_robot = nullptr;
    Est_ _robot_in_local_map = Est_::Identity(), _sensor_in_robot = Est_::Identity(); Status _status = Error;
  };
} // namespace srrg2_slam_interfaces

namespace srrg2_laser_slam_2d {
  using namespace srrg2_core;
  using srrg2_slam_interfaces::MergerBase;
  using MergerPointNormal2f = srrg2_slam_interfaces::Merger_<Isometry2f, PointNormal2fVectorCloud, PointNormal2fVectorCloud>;
  using SceneClipperPointNormal2f = srrg2_slam_interfaces::SceneClipper_<Isometry2f, PointNormal2fVectorCloud>;
  // the reference package's own declarations the adapters derive from / recognise (interface only: PARAM names and defaults
  // from registration/correspondence_finder_normal_2f.h:9-13, ..._projective_2d.h:16-26, ..._kd_tree_2d.h:23-34, ..._nn_2d.h:20-30,
  // registration/aligner_slice_processor_laser_2d.h:7-42); their compute() bodies are NOT restated
  using CorrespondenceFinderNormal2f = srrg2_slam_interfaces::CorrespondenceFinder_<Isometry2f, PointNormal2fVectorCloud, PointNormal2fVectorCloud>;
  class CorrespondenceFinderProjective2f : public CorrespondenceFinderNormal2f {
  public:
    PARAM(PropertyFloat, point_distance, "", 0.5f, 0);
    PARAM(PropertyFloat, normal_cos, "", 0.8f, 0);
    PARAM(PropertyConfigurable_<PointNormal2fProjectorPolar>, projector, "", PointNormal2fProjectorPolarPtr(new PointNormal2fProjectorPolar), 0);
    void compute() override { throw std::logic_error("shim: the reference finder is not restated"); }
  };
  class CorrespondenceFinderKDTree2D : public CorrespondenceFinderNormal2f {
  public:
    PARAM(PropertyFloat, max_distance_m, "", 1e-2f, 0);
    PARAM(PropertyFloat, max_leaf_range, "", 1e-2f, 0);
    PARAM(PropertyUnsignedInt, min_leaf_points, "", 20, 0);
    PARAM(PropertyFloat, normal_cos, "", 0.8f, 0);
    void compute() override { throw std::logic_error("shim: the reference finder is not restated"); }
  };
  class CorrespondenceFinderNN2D : public CorrespondenceFinderNormal2f {
  public:
    PARAM(PropertyFloat, max_distance_m, "", 1.f, 0);
    PARAM(PropertyFloat, resolution, "", 0.05f, 0);
    PARAM(PropertyFloat, normal_cos, "", 0.8f, 0);
    void compute() override { throw std::logic_error("shim: the reference finder is not restated"); }
  };
  class AlignerSliceProcessorLaser2D : public srrg2_slam_interfaces::AlignerSliceProcessorCloud_<PointNormal2fVectorCloud, PointNormal2fVectorCloud> {};
  class AlignerSliceProcessorLaser2DWithSensor : public srrg2_slam_interfaces::AlignerSliceProcessorCloud_<PointNormal2fVectorCloud, PointNormal2fVectorCloud> {};
} // namespace srrg2_laser_slam_2d
