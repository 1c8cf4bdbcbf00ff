// CPU test of include/loik_amd/pinocchio_adapter.hpp: the conversion from a Pinocchio-shaped model and Eigen-shaped arguments
// to the wrapper's types.  Pinocchio / Eigen are not in this image; `shape::` below has the INTERFACE the adapter reads of
// pinocchio::Model / JointModel / SE3 (member names, index types, column-major rotation storage like Eigen) and nothing else.
// Round trip: built-in table -> shape::Model -> to_loik_amd -> must equal the table.  No GPU, no library call that launches.
#include "loik_amd/pinocchio_adapter.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

using namespace loik_amd;

static int failures = 0;
#define CHECK(cond)                                                              \
  do {                                                                           \
    if (!(cond)) { ++failures; std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)

namespace shape {
struct Rotation {  // Eigen::Matrix3d: column-major storage, (r, c) access
  double d[9];
  double operator()(int r, int c) const { return d[3 * c + r]; }
};
struct Translation {
  double d[3];
  double operator[](int k) const { return d[k]; }
};
struct SE3 {
  Rotation R;
  Translation t;
  const Rotation& rotation() const { return R; }
  const Translation& translation() const { return t; }
};
struct JointModel {
  std::string sn;
  int iq = 0, iv = 0;
  double ax[6] = {0, 0, 0, 0, 0, 0};   // (axis; JointModelUniversal: axis1, axis2)
  double m_pitch = 0.0;                // JointModelHelical*
  std::vector<std::pair<JointModel, SE3>> subs;  // JointModelComposite::joints / ::jointPlacements
  std::string shortname() const { return sn; }
  int idx_q() const { return iq; }
  int idx_v() const { return iv; }
};
struct Model {
  int njoints = 0, nq = 0, nv = 0;
  std::vector<std::size_t> parents;  // JointIndex
  std::vector<std::string> names;
  std::vector<JointModel> joints;
  std::vector<SE3> jointPlacements;
};
struct Matrix6 {  // column-major 6x6
  double d[36];
  double operator()(int r, int c) const { return d[6 * c + r]; }
};
struct Vector6 {
  double d[6];
  double operator[](int k) const { return d[k]; }
};
struct Motion {
  Vector6 v;
  Vector6 toVector() const { return v; }
};
struct VectorX {
  std::vector<double> d;
  long size() const { return (long)d.size(); }
  double operator[](long k) const { return d[(std::size_t)k]; }
};
}  // namespace shape

static const char* short_name(int t)
{
  static const char* names[] = {"", "JointModelRX", "JointModelRY", "JointModelRZ", "JointModelPX", "JointModelPY", "JointModelPZ",
                                "JointModelRevoluteUnaligned", "JointModelPrismaticUnaligned", "JointModelFreeFlyer",
                                "JointModelSpherical", "JointModelTranslation", "JointModelSphericalZYX", "JointModelPlanar",
                                "JointModelRUBX", "JointModelRUBY", "JointModelRUBZ", "JointModelComposite",
                                "JointModelRevoluteUnboundedUnaligned", "JointModelHX", "JointModelHY", "JointModelHZ",
                                "JointModelHelicalUnaligned"};
  return names[t];
}

static shape::Model pinocchio_shaped(const Model& m)
{
  shape::Model p;
  p.njoints = m.njoints; p.nq = m.nq; p.nv = m.nv;
  for (int i = 0; i < m.njoints; ++i) {
    p.parents.push_back((std::size_t)m.parents[i]);
    p.names.push_back(m.names[i]);
    shape::JointModel j;
    j.sn = short_name(m.jtype[i]); j.iq = m.idx_q[i]; j.iv = m.idx_v[i];
    for (int k = 0; k < 3; ++k) j.ax[k] = m.axis[3 * i + k];
    if (!m.pitch.empty()) j.m_pitch = m.pitch[i];
    p.joints.push_back(j);
    shape::SE3 P;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) P.R.d[3 * c + r] = m.jointPlacements[12 * i + 3 * r + c];
    for (int k = 0; k < 3; ++k) P.t.d[k] = m.jointPlacements[12 * i + 9 + k];
    p.jointPlacements.push_back(P);
  }
  return p;
}

static shape::SE3 se3_of(const double* P12)
{
  shape::SE3 P;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) P.R.d[3 * c + r] = P12[3 * r + c];
  for (int k = 0; k < 3; ++k) P.t.d[k] = P12[9 + k];
  return P;
}

static void round_trip(const Model& m)
{
  shape::Model p = pinocchio_shaped(m);
  for (int i = 1; i < m.njoints; ++i)
    if (m.jtype[i] == LOIKB_J_COMPOSITE)
      for (int k = 0; k < m.comp_count[i]; ++k) {
        const int e = m.comp_first[i] + k;
        shape::JointModel sj;
        sj.sn = short_name(m.comp_jtype[e]);
        for (int c = 0; c < 3; ++c) sj.ax[c] = m.comp_axis[3 * e + c];
        if (!m.comp_pitch.empty()) sj.m_pitch = m.comp_pitch[e];
        p.joints[i].subs.emplace_back(sj, se3_of(&m.comp_placement[12 * e]));
      }
  const Model o = to_loik_amd(p, [](const shape::JointModel& j, const std::string&) { return j.ax; },
                              [](const shape::JointModel& j) { return j.subs; },
                              [](const shape::JointModel& j, const std::string&) { return j.m_pitch; });
  CHECK(o.pitch == m.pitch);
  CHECK(o.comp_pitch == m.comp_pitch);
  CHECK(o.comp_first == m.comp_first && o.comp_count == m.comp_count && o.comp_jtype == m.comp_jtype);
  CHECK(o.comp_axis == m.comp_axis && o.comp_placement == m.comp_placement);
  CHECK(o.njoints == m.njoints && o.nq == m.nq && o.nv == m.nv);
  CHECK(o.parents == m.parents && o.jtype == m.jtype && o.idx_q == m.idx_q && o.idx_v == m.idx_v);
  CHECK(o.names == m.names && o.jointPlacements == m.jointPlacements);
  for (int i = 1; i < m.njoints; ++i)
    for (int k = 0; k < 3; ++k) {
      // aligned joints carry no axis in Pinocchio: the adapter leaves it zero, the library derives it from the type
      const bool unaligned = m.jtype[i] == LOIKB_J_RU || m.jtype[i] == LOIKB_J_PU || m.jtype[i] == LOIKB_J_RUBU || m.jtype[i] == LOIKB_J_HU;
      CHECK(o.axis[3 * i + k] == (unaligned ? m.axis[3 * i + k] : 0.0));
    }
}

int main()
{
  for (const char* name : {"talos32", "talos32_freeflyer", "talos44", "panda7", "panda9"}) round_trip(Model::Builtin(name));
  {  // a model with every joint type, unaligned axes, rotated placements
    Model m;
    const int types[] = {LOIKB_J_NONE, LOIKB_J_FREEFLYER, LOIKB_J_RU, LOIKB_J_PU, LOIKB_J_SPHERICAL, LOIKB_J_TRANSLATION,
                         LOIKB_J_SPHERICAL_ZYX, LOIKB_J_PLANAR, LOIKB_J_RUBY, LOIKB_J_PZ, LOIKB_J_RUBU, LOIKB_J_COMPOSITE,
                         LOIKB_J_HY, LOIKB_J_HU};   // (two helical joints: aligned and unaligned, with their pitch)
    const int nqs[] = {0, 7, 1, 1, 4, 3, 3, 4, 2, 1, 2, 6, 1, 1}, nvs[] = {0, 6, 1, 1, 3, 3, 3, 3, 1, 1, 1, 5, 1, 1};
    m.njoints = 14;
    for (int i = 0; i < m.njoints; ++i) {
      m.parents.push_back(i ? (i - 1) / 2 : 0);
      m.jtype.push_back(types[i]);
      m.idx_q.push_back(m.nq); m.idx_v.push_back(m.nv);
      m.nq += nqs[i]; m.nv += nvs[i];
      const double a[3] = {std::sin(1.0 + i), std::cos(2.0 * i), 0.3};
      const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
      const bool un = types[i] == LOIKB_J_RU || types[i] == LOIKB_J_PU || types[i] == LOIKB_J_RUBU || types[i] == LOIKB_J_HU;
      m.pitch.push_back(types[i] >= LOIKB_J_HX ? 0.1 * i - 1.0 : 0.0);
      for (int k = 0; k < 3; ++k) m.axis.push_back(un ? a[k] / n : 0.0);
      const double c = std::cos(0.3 * i), s = std::sin(0.3 * i);
      const double P[12] = {c, -s, 0, s, c, 0, 0, 0, 1, 0.1 * i, -0.2, 0.05 * i};  // Rz(0.3 i): not symmetric -> order matters
      m.jointPlacements.insert(m.jointPlacements.end(), P, P + 12);
      m.names.push_back(i ? "joint_" + std::to_string(i) : "universe");
      // joint 11: a composite of RU, HelicalUnaligned (with its pitch), Spherical (nq 4, nv 3) with rotated internal placements
      m.comp_first.push_back(i == 11 ? 0 : (i < 11 ? 0 : 3));
      m.comp_count.push_back(i == 11 ? 3 : 0);
    }
    m.comp_jtype = {LOIKB_J_RU, LOIKB_J_HU, LOIKB_J_SPHERICAL};
    m.comp_axis = {0.6, 0.0, 0.8, 0.0, 0.8, -0.6, 0, 0, 0};
    m.comp_pitch = {0.0, 0.07, 0.0};
    for (int k = 0; k < 3; ++k) {
      const double c = std::cos(0.7 + k), s = std::sin(0.7 + k);
      const double P[12] = {1, 0, 0, 0, c, -s, 0, s, c, 0.01 * k, 0.2, -0.1};  // Rx
      m.comp_placement.insert(m.comp_placement.end(), P, P + 12);
    }
    round_trip(m);
    shape::Model p = pinocchio_shaped(m);
    p.joints[3].sn = "JointModelMimic";
    bool thrown = false;
    try { (void)to_loik_amd(p, [](const shape::JointModel& j, const std::string&) { return j.ax; }); }
    catch (const std::runtime_error& e) { thrown = std::strstr(e.what(), "JointModelMimic") && std::strstr(e.what(), "joint_3"); }
    CHECK(thrown);
    {  // JointModelUniversal(axis1, axis2): leaves the adapter as the composite of RevoluteUnaligned(axis1), (axis2), identity
       // placements; as a sub-joint of a composite: the same two in place, the first with the sub-joint's placement
      shape::Model u = pinocchio_shaped(m);
      u.joints[2].sn = "JointModelUniversal";     // (joint 2 was 1-DoF: the coordinates behind it move by one)
      const double axes[6] = {0.0, 0.6, 0.8, 1.0, 0.0, 0.0};
      for (int k = 0; k < 6; ++k) u.joints[2].ax[k] = axes[k];
      for (int i = 3; i < u.njoints; ++i) { u.joints[i].iq += 1; u.joints[i].iv += 1; }
      u.nq += 1; u.nv += 1;
      shape::JointModel su; su.sn = "JointModelUniversal";
      for (int k = 0; k < 6; ++k) su.ax[k] = axes[5 - k];
      shape::JointModel sp; sp.sn = "JointModelPZ";
      const double Psub[12] = {0, -1, 0, 1, 0, 0, 0, 0, 1, 0.3, 0.2, 0.1};
      u.joints[11].subs = {{sp, se3_of(Psub)}, {su, se3_of(Psub)}};   // nq 3, nv 3
      u.nq += 3 - 6; u.nv += 3 - 5;
      const Model o = to_loik_amd(u, [](const shape::JointModel& j, const std::string&) { return j.ax; },
                                  [](const shape::JointModel& j) { return j.subs; },
                                  [](const shape::JointModel& j, const std::string&) { return j.m_pitch; });
      CHECK(o.jtype[2] == LOIKB_J_COMPOSITE && o.comp_count[2] == 2 && o.comp_first[2] == 0);
      CHECK(o.comp_jtype[0] == LOIKB_J_RU && o.comp_jtype[1] == LOIKB_J_RU);
      for (int k = 0; k < 6; ++k) CHECK(o.comp_axis[k] == axes[k]);
      for (int h = 0; h < 2; ++h)
        for (int k = 0; k < 12; ++k) CHECK(o.comp_placement[12 * h + k] == ((k < 9 && k % 4 == 0) ? 1.0 : 0.0));
      CHECK(o.comp_first[11] == 2 && o.comp_count[11] == 3);
      CHECK(o.comp_jtype[2] == LOIKB_J_PZ && o.comp_jtype[3] == LOIKB_J_RU && o.comp_jtype[4] == LOIKB_J_RU);
      for (int k = 0; k < 6; ++k) CHECK(o.comp_axis[9 + k] == axes[5 - k]);
      for (int k = 0; k < 12; ++k) {
        CHECK(o.comp_placement[12 * 3 + k] == Psub[k]);
        CHECK(o.comp_placement[12 * 4 + k] == ((k < 9 && k % 4 == 0) ? 1.0 : 0.0));
      }
      CHECK(o.nq == u.nq && o.nv == u.nv && o.idx_q[3] == u.joints[3].iq);
    }
    thrown = false;   // a helical joint needs the PitchOf functor
    try {
      shape::Model hp = pinocchio_shaped(m);
      for (int i = 1; i < m.njoints; ++i)
        if (m.jtype[i] == LOIKB_J_COMPOSITE) hp.joints[i].sn = "JointModelRZ";
      (void)to_loik_amd(hp, [](const shape::JointModel& j, const std::string&) { return j.ax; });
    } catch (const std::runtime_error& e) { thrown = std::strstr(e.what(), "PitchOf") != nullptr; }
    CHECK(thrown);
    p.joints[3].sn = "JointModelComposite";   // a composite needs the SubJointsOf functor
    thrown = false;
    try { (void)to_loik_amd(p, [](const shape::JointModel& j, const std::string&) { return j.ax; }); }
    catch (const std::runtime_error& e) { thrown = std::strstr(e.what(), "SubJointsOf") != nullptr; }
    CHECK(thrown);
  }
  {  // argument conversions: column-major in, row-major out; Motion through toVector(); lists; VectorXd
    shape::Matrix6 M;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) M.d[6 * c + r] = 10.0 * r + c;
    const Mat6x6 R = to_rowmajor(M);
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) CHECK(R[6 * r + c] == 10.0 * r + c);
    shape::Motion v{{{1, 2, 3, 4, 5, 6}}};
    const Vec6 a = to_vec6(v), b = to_vec6(v.v);
    for (int k = 0; k < 6; ++k) CHECK(a[k] == k + 1 && b[k] == k + 1);
    CHECK(to_rowmajor_list(std::vector<shape::Matrix6>{M, M}).size() == 2);
    CHECK(to_vec6_list(std::vector<shape::Vector6>{v.v})[0][5] == 6.0);
    const DVec d = to_dvec(shape::VectorX{{0.5, -0.5, 2.0}});
    CHECK(d.size() == 3 && d[2] == 2.0);
  }
  std::printf(failures ? "%d CHECKS FAILED\n" : "all adapter checks passed\n", failures);
  return failures ? 1 : 0;
}
