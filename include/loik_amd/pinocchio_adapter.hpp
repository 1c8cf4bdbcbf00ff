// loik_amd/pinocchio_adapter.hpp -- from a Pinocchio model / Eigen arguments to the types of loik_amd/loik.hpp.
//
// What a caller of the reference holds is a `pinocchio::Model` (built by urdf::buildModel, /root/reference/tests/loik-loid.cpp:
// 108-113) and Eigen arguments (`Mat6x6 H_ref`, `Motion v_ref`, `PINOCCHIO_ALIGNED_STD_VECTOR(Mat6x6) Ais`, `DVec lb`, ...,
// loik-loid-optimized.hpp:335-361).  This header converts them.  It is written against the INTERFACE of those types, as
// templates, so that it compiles and is tested in this repository without Pinocchio or Eigen (tests/cpp/test_adapter.cpp drives
// it with a model type of the same shape); with Pinocchio included before this header, `to_loik_amd(const pinocchio::Model&)`
// is available directly.
//
//   members used of the model type M:   njoints, nq, nv (int-like), parents[i], names[i], joints[i], jointPlacements[i]
//   of a joint model  M::joints[i]:     shortname() -> std::string ("JointModelRZ", "JointModelRevoluteUnaligned", ...),
//                                        idx_q(), idx_v()
//   of a placement    jointPlacements[i]: rotation()(r, c), translation()[k]              (pinocchio::SE3)
//   AxisOf(joint, shortname) -> something indexable [0..2]: the axis of an unaligned joint (JointModelRevoluteUnaligned::axis);
//                         for a JointModelUniversal indexable [0..5]: (axis1, axis2)
//   PitchOf(joint, shortname) -> double: JointModelHelical*::m_pitch (only called for helical joints; the 3-argument overloads
//                         of to_loik_amd refuse a model with a helical joint)
//   SubJointsOf(joint) -> a range of (sub-joint model, placement) pairs (.first / .second) of a JointModelComposite
//                         (JointModelComposite::joints[k], ::jointPlacements[k]); any supported joint type but a composite
// JointModelUniversal(axis1, axis2) -- M = R(axis1, q0) R(axis2, q1), S(q) = [R(axis2, q1)^T axis1 | axis2] -- is handed over
// as what it is: the composite of RevoluteUnaligned(axis1) and RevoluteUnaligned(axis2) with identity placements (same q, same
// nu; tests/test_composite.py::test_universal_joint_is_the_composite_of_its_two_revolute_joints).
#pragma once

#include <array>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "loik_amd/loik.hpp"

namespace loik_amd {

// (adapter-internal marker, never crosses the C-ABI: a universal joint leaves this header as a LOIKB_J_COMPOSITE)
enum { LOIKB_J_UNIVERSAL_AS_COMPOSITE = 1000 };

// joint type of include/loik_amd_models.h for a Pinocchio joint short name; LOIKB_J_NONE for an unknown one
inline int joint_type_of(const std::string& n)
{
  static const struct { const char* name; int type; } table[] = {
      {"JointModelRX", LOIKB_J_RX}, {"JointModelRY", LOIKB_J_RY}, {"JointModelRZ", LOIKB_J_RZ},
      {"JointModelPX", LOIKB_J_PX}, {"JointModelPY", LOIKB_J_PY}, {"JointModelPZ", LOIKB_J_PZ},
      {"JointModelRevoluteUnaligned", LOIKB_J_RU}, {"JointModelPrismaticUnaligned", LOIKB_J_PU},
      {"JointModelFreeFlyer", LOIKB_J_FREEFLYER},        // nq 7 (t, quat xyzw), nv 6
      {"JointModelSpherical", LOIKB_J_SPHERICAL},        // nq 4 (quat xyzw), nv 3
      {"JointModelTranslation", LOIKB_J_TRANSLATION},    // nq 3, nv 3
      {"JointModelSphericalZYX", LOIKB_J_SPHERICAL_ZYX}, // nq 3 (z, y, x angles), nv 3
      {"JointModelPlanar", LOIKB_J_PLANAR},              // nq 4 (x, y, cos, sin), nv 3
      {"JointModelRUBX", LOIKB_J_RUBX}, {"JointModelRUBY", LOIKB_J_RUBY}, {"JointModelRUBZ", LOIKB_J_RUBZ},  // nq 2 (cos, sin)
      {"JointModelRevoluteUnboundedUnaligned", LOIKB_J_RUBU},  // nq 2 (cos, sin), axis from axis_of
      {"JointModelComposite", LOIKB_J_COMPOSITE},        // loikb_model_desc.comp_*
      {"JointModelUniversal", LOIKB_J_UNIVERSAL_AS_COMPOSITE},  // -> composite of two unaligned revolute joints (below)
      {"JointModelHX", LOIKB_J_HX}, {"JointModelHY", LOIKB_J_HY}, {"JointModelHZ", LOIKB_J_HZ},  // helical: pitch from pitch_of
      {"JointModelHelicalUnaligned", LOIKB_J_HU},
  };
  for (const auto& e : table)
    if (n == e.name) return e.type;
  return LOIKB_J_NONE;
}

namespace detail {
template <class SE3Like>
void push_placement(std::vector<double>& out, const SE3Like& P)
{
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out.push_back(P.rotation()(r, c));  // row-major here, whatever Eigen stores
  for (int k = 0; k < 3; ++k) out.push_back(P.translation()[k]);
}
struct NoPlacement {  // (never read: NoComposite throws)
  struct Rot { double operator()(int, int) const { return 0.0; } };
  Rot rotation() const { return Rot(); }
  const double* translation() const { static const double z[3] = {0.0, 0.0, 0.0}; return z; }
};
struct NoComposite {
  template <class J>
  std::vector<std::pair<J, NoPlacement>> operator()(const J&) const
  {
    throw std::runtime_error("loik_amd: the model has a JointModelComposite: pass a SubJointsOf functor to to_loik_amd");
  }
};
}  // namespace detail

struct NoPitch {
  template <class J>
  double operator()(const J&, const std::string& n) const
  {
    throw std::runtime_error("loik_amd: the model has a helical joint (" + n + "): pass a PitchOf functor to to_loik_amd");
  }
};

template <class PinocchioModel, class AxisOf, class SubJointsOf, class PitchOf>
Model to_loik_amd(const PinocchioModel& m, AxisOf axis_of, SubJointsOf sub_joints_of, PitchOf pitch_of)
{
  Model o;
  bool any_helical = false, any_helical_sub = false;
  o.njoints = static_cast<int>(m.njoints); o.nq = static_cast<int>(m.nq); o.nv = static_cast<int>(m.nv);
  bool any_composite = false;
  for (std::size_t i = 0; i < static_cast<std::size_t>(m.njoints); ++i) {
    o.parents.push_back(static_cast<int>(m.parents[i]));
    o.idx_q.push_back(i ? static_cast<int>(m.joints[i].idx_q()) : 0);
    o.idx_v.push_back(i ? static_cast<int>(m.joints[i].idx_v()) : 0);
    const std::string n = i ? m.joints[i].shortname() : std::string();
    int t = i ? joint_type_of(n) : LOIKB_J_NONE;
    if (i && t == LOIKB_J_NONE)
      throw std::runtime_error("loik_amd: joint type '" + n + "' of joint '" + m.names[i] +
                               "' is not supported (mimic)");
    double ax[3] = {0.0, 0.0, 0.0};
    if (t == LOIKB_J_RU || t == LOIKB_J_PU || t == LOIKB_J_RUBU || t == LOIKB_J_HU) {
      const auto a = axis_of(m.joints[i], n);
      for (int k = 0; k < 3; ++k) ax[k] = a[k];
    }
    o.pitch.push_back((t >= LOIKB_J_HX && t <= LOIKB_J_HU) ? static_cast<double>(pitch_of(m.joints[i], n)) : 0.0);
    any_helical = any_helical || (t >= LOIKB_J_HX && t <= LOIKB_J_HU);
    o.comp_first.push_back(static_cast<int>(o.comp_jtype.size()));
    int count = 0;
    static const double ident[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    // the two revolute joints of a universal joint; `first_placement`: 12 doubles (row-major R, t) of the first one
    auto push_universal = [&](const auto& joint, const std::string& jn, const double* first_placement) {
      const auto a = axis_of(joint, jn);  // (axis1, axis2)
      for (int h = 0; h < 2; ++h) {
        o.comp_jtype.push_back(LOIKB_J_RU);
        o.comp_axis.insert(o.comp_axis.end(), {a[3 * h], a[3 * h + 1], a[3 * h + 2]});
        const double* P = h ? ident : first_placement;
        o.comp_placement.insert(o.comp_placement.end(), P, P + 12);
        ++count;
      }
    };
    if (t == LOIKB_J_UNIVERSAL_AS_COMPOSITE) {
      any_composite = true;
      push_universal(m.joints[i], n, ident);
      t = LOIKB_J_COMPOSITE;
    } else if (t == LOIKB_J_COMPOSITE) {
      any_composite = true;
      for (const auto& sub : sub_joints_of(m.joints[i])) {
        const std::string sn = sub.first.shortname();
        const int st = joint_type_of(sn);
        if (st == LOIKB_J_NONE || st == LOIKB_J_COMPOSITE)
          throw std::runtime_error("loik_amd: sub-joint '" + sn + "' of the composite joint '" + m.names[i] + "' is not supported");
        if (st == LOIKB_J_UNIVERSAL_AS_COMPOSITE) {
          std::vector<double> P;
          detail::push_placement(P, sub.second);
          push_universal(sub.first, sn, P.data());
          continue;
        }
        double sa[3] = {0.0, 0.0, 0.0};
        if (st == LOIKB_J_RU || st == LOIKB_J_PU || st == LOIKB_J_RUBU || st == LOIKB_J_HU) {
          const auto a = axis_of(sub.first, sn);
          for (int k = 0; k < 3; ++k) sa[k] = a[k];
        }
        o.comp_jtype.push_back(st);
        o.comp_axis.insert(o.comp_axis.end(), {sa[0], sa[1], sa[2]});
        detail::push_placement(o.comp_placement, sub.second);
        o.comp_pitch.resize(o.comp_jtype.size(), 0.0);
        if (st >= LOIKB_J_HX && st <= LOIKB_J_HU) { o.comp_pitch.back() = static_cast<double>(pitch_of(sub.first, sn)); any_helical_sub = true; }
        ++count;
      }
    }
    o.comp_count.push_back(count);
    o.jtype.push_back(t);
    o.axis.insert(o.axis.end(), {ax[0], ax[1], ax[2]});
    detail::push_placement(o.jointPlacements, m.jointPlacements[i]);
    o.names.push_back(m.names[i]);
  }
  if (!any_composite) { o.comp_first.clear(); o.comp_count.clear(); }
  if (!any_helical) o.pitch.clear();
  if (any_helical_sub) o.comp_pitch.resize(o.comp_jtype.size(), 0.0); else o.comp_pitch.clear();
  return o;
}

template <class PinocchioModel, class AxisOf, class SubJointsOf>
Model to_loik_amd(const PinocchioModel& m, AxisOf axis_of, SubJointsOf sub_joints_of)
{
  return to_loik_amd(m, axis_of, sub_joints_of, NoPitch());
}

template <class PinocchioModel, class AxisOf>
Model to_loik_amd(const PinocchioModel& m, AxisOf axis_of)
{
  return to_loik_amd(m, axis_of, detail::NoComposite());
}

// Eigen's 6x6 (column-major by default) -> row-major array
template <class Matrix6>
Mat6x6 to_rowmajor(const Matrix6& M)
{
  Mat6x6 o;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) o[6 * r + c] = M(r, c);
  return o;
}
// pinocchio::Motion (toVector() = [linear; angular]) or any 6-vector with operator[]
template <class MotionLike>
auto to_vec6(const MotionLike& v) -> decltype(v.toVector(), Vec6())
{
  Vec6 o;
  const auto x = v.toVector();
  for (int k = 0; k < 6; ++k) o[k] = x[k];
  return o;
}
template <class VectorLike>
auto to_vec6(const VectorLike& v) -> decltype(v[0], Vec6())
{
  Vec6 o;
  for (int k = 0; k < 6; ++k) o[k] = v[k];
  return o;
}
// Eigen::VectorXd -> DVec
template <class VectorLike>
DVec to_dvec(const VectorLike& v)
{
  DVec o(static_cast<std::size_t>(v.size()));
  for (std::size_t k = 0; k < o.size(); ++k) o[k] = v[static_cast<decltype(v.size())>(k)];
  return o;
}
// aligned_vector<Mat6x6> / <Vec6> -> std::vector of the array types
template <class Matrices>
std::vector<Mat6x6> to_rowmajor_list(const Matrices& Ms)
{
  std::vector<Mat6x6> o;
  for (const auto& M : Ms) o.push_back(to_rowmajor(M));
  return o;
}
template <class Vectors>
std::vector<Vec6> to_vec6_list(const Vectors& vs)
{
  std::vector<Vec6> o;
  for (const auto& v : vs) o.push_back(to_vec6(v));
  return o;
}

#ifdef PINOCCHIO_MAJOR_VERSION  // <pinocchio/...> was included before this header
inline Model to_loik_amd(const pinocchio::Model& m)
{
  auto axis_of = [](const pinocchio::JointModel& j, const std::string& n) {
    std::array<double, 6> a{};
    auto put = [&a](const auto& v, int at) { for (int k = 0; k < 3; ++k) a[at + k] = v[k]; };
    if (n == "JointModelRevoluteUnaligned") put(boost::get<pinocchio::JointModelRevoluteUnaligned>(j.toVariant()).axis, 0);
    else if (n == "JointModelRevoluteUnboundedUnaligned") put(boost::get<pinocchio::JointModelRevoluteUnboundedUnaligned>(j.toVariant()).axis, 0);
    else if (n == "JointModelUniversal") {
      const auto& u = boost::get<pinocchio::JointModelUniversal>(j.toVariant());
      put(u.axis1, 0); put(u.axis2, 3);
    } else if (n == "JointModelHelicalUnaligned") put(boost::get<pinocchio::JointModelHelicalUnaligned>(j.toVariant()).axis, 0);
    else put(boost::get<pinocchio::JointModelPrismaticUnaligned>(j.toVariant()).axis, 0);
    return a;
  };
  auto sub_joints_of = [](const pinocchio::JointModel& j) {
    const auto& c = boost::get<pinocchio::JointModelComposite>(j.toVariant());
    std::vector<std::pair<pinocchio::JointModel, pinocchio::SE3>> subs;
    for (std::size_t k = 0; k < c.joints.size(); ++k) subs.emplace_back(pinocchio::JointModel(c.joints[k]), c.jointPlacements[k]);
    return subs;
  };
  auto pitch_of = [](const pinocchio::JointModel& j, const std::string& n) -> double {
    if (n == "JointModelHX") return boost::get<pinocchio::JointModelHX>(j.toVariant()).m_pitch;
    if (n == "JointModelHY") return boost::get<pinocchio::JointModelHY>(j.toVariant()).m_pitch;
    if (n == "JointModelHZ") return boost::get<pinocchio::JointModelHZ>(j.toVariant()).m_pitch;
    return boost::get<pinocchio::JointModelHelicalUnaligned>(j.toVariant()).m_pitch;
  };
  return to_loik_amd(m, axis_of, sub_joints_of, pitch_of);
}
#endif

}  // namespace loik_amd
