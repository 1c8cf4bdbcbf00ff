// loik_amd/loik.hpp -- C++17 host-side mirror of the reference's solver interface over the C-ABI (loik_amd.h).
//
// Same class / method names, argument order and error behaviour as
//   loik::FirstOrderLoikOptimizedTpl<double>   (/root/reference/include/loik/loik-loid-optimized.hpp:22-808)
//   loik::IkIdDataTypeOptimizedTpl<double>     (/root/reference/include/loik/loik-loid-data-optimized.hpp:62-379)
//   loik::IkIdSolverBaseTpl<double>            (/root/reference/include/loik/task-solver-base.hpp:21-174)
// so a caller of the reference switches by changing the namespace and the vector types (no Eigen / Pinocchio in
// this image: 6-vectors are std::array<double,6> in Pinocchio's [linear; angular] order, 6x6 matrices are row-major
// std::array<double,36>; INTEGRATION.md shows the two-line Eigen/Pinocchio adapters).
//
// The one addition is the batch: `batch` independent problem instances are solved by one object on one MI355X.
// With batch == 1 every call has exactly the reference's single-instance meaning.  For batch > 1 the per-instance
// inputs are instance-major arrays (q[batch*nq], bis[batch*nc], ...); results land in the caller-owned data
// object, as upstream (the solver keeps `IkIdData&`, loik-loid-optimized.hpp:763).
// All compute happens in libloik_amd.so on the GPU; there is no CPU path behind this header.
#pragma once

#include "../loik_amd.h"

#include <array>
#include <map>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <vector>

namespace loik_amd {

// task-solver-base.hpp:13-18
enum ADMMPenaltyUpdateStrat { DEFAULT = 0, OSQP = 1, MAXEIGENVALUE = 3 };

using Index = std::size_t;
using DVec = std::vector<double>;
using Vec6 = std::array<double, 6>;
using Mat6x6 = std::array<double, 36>;  // row-major
using Motion = Vec6;                    // [linear; angular]

inline Mat6x6 Identity6()
{
  Mat6x6 m{};
  for (int k = 0; k < 6; ++k) m[7 * k] = 1.0;
  return m;
}

// the members of pinocchio::Model the hot path reads (loik-loid-optimized.hxx:46-47, :118-119, :257-265)
struct Model {
  int njoints = 0, nq = 0, nv = 0;
  std::vector<int> parents, jtype, idx_q, idx_v;
  std::vector<double> axis;             // [njoints][3]
  std::vector<double> jointPlacements;  // [njoints][12]: R row-major, t
  std::vector<std::string> names;
  // JointModelComposite (empty when the model has none): joint i of type LOIKB_J_COMPOSITE = sub-joints comp_first[i] ..
  // comp_first[i] + comp_count[i] - 1 of comp_jtype / comp_axis / comp_placement (include/loik_amd_models.h)
  std::vector<int> comp_first, comp_count, comp_jtype;
  std::vector<double> comp_axis, comp_placement;
  std::vector<double> comp_pitch;  // [n_sub] the same for helical sub-joints of composites (empty: none)
  std::vector<double> pitch;  // [njoints] JointModelHelical*::m_pitch (LOIKB_J_HX .. HU); empty when the model has no helical joint

  static Model Builtin(const std::string& name)
  {
    loikb_model_desc d{};
    if (loikb_builtin_model(name.c_str(), &d, nullptr, nullptr) != 0) throw std::runtime_error("unknown built-in model " + name);
    Model m;
    m.njoints = d.njoints; m.nq = d.nq; m.nv = d.nv;
    m.parents.assign(d.parents, d.parents + d.njoints);
    m.jtype.assign(d.jtype, d.jtype + d.njoints);
    m.idx_q.assign(d.idx_q, d.idx_q + d.njoints);
    m.idx_v.assign(d.idx_v, d.idx_v + d.njoints);
    m.axis.assign(d.axis, d.axis + 3 * d.njoints);
    m.jointPlacements.assign(d.placement, d.placement + 12 * d.njoints);
    for (int i = 0; i < d.njoints; ++i) m.names.emplace_back(loikb_builtin_joint_name(name.c_str(), i));
    return m;
  }
  Index getJointId(const std::string& n) const
  {
    for (std::size_t i = 0; i < names.size(); ++i)
      if (names[i] == n) return i;
    return names.size();
  }
  loikb_model_desc desc() const
  {
    loikb_model_desc d{};
    d.njoints = njoints; d.nq = nq; d.nv = nv;
    d.parents = parents.data(); d.jtype = jtype.data(); d.axis = axis.data();
    d.idx_q = idx_q.data(); d.idx_v = idx_v.data(); d.placement = jointPlacements.data();
    if (!comp_jtype.empty()) {
      d.comp_first = comp_first.data(); d.comp_count = comp_count.data(); d.comp_jtype = comp_jtype.data();
      d.comp_axis = comp_axis.data(); d.comp_placement = comp_placement.data();
    }
    if (!pitch.empty()) d.pitch = pitch.data();
    if (!comp_pitch.empty()) d.comp_pitch = comp_pitch.data();
    return d;
  }
};

class FirstOrderLoikOptimized;

// LoikSolverInfo of the reference (loik-loid-optimized.hpp:47-127; base lists task-solver-base.hpp:25-52): the lists a solver
// constructed with logging = true fills, for one instance.  Upstream keeps the struct protected; get_solver_info() hands it out.
struct LoikSolverInfo {
  std::vector<int> iter_list_, tail_solve_iter_list_;
  std::vector<double> primal_residual_task_list_, primal_residual_slack_list_, primal_residual_list_, dual_residual_nu_list_,
      dual_residual_v_list_, dual_residual_list_, mu_list_, mu_eq_list_, mu_ineq_list_;
  int Size() const { return (int)iter_list_.size(); }
};

// A member of the data object that is fetched from the device the first time it is read after a solve (His, pis, Aty,
// liMi: large, rarely read -- upstream's tests read His / pis, tests/loik-loid.cpp:597-615).  Reads like the std::vector it
// wraps: operator[], data(), size(), begin()/end(), implicit conversion to const DVec&.
class LazyField {
public:
  LazyField() = default;
  LazyField(int field, std::size_t n) : field_(field), buf_(n, 0.0) {}
  const DVec& get() const;
  operator const DVec&() const { return get(); }
  double operator[](std::size_t i) const { return get()[i]; }
  const double* data() const { return get().data(); }
  std::size_t size() const { return buf_.size(); }
  DVec::const_iterator begin() const { return get().begin(); }
  DVec::const_iterator end() const { return get().end(); }

private:
  friend class FirstOrderLoikOptimized;
  friend struct IkIdDataOptimized;
  int field_ = -1;
  mutable DVec buf_;
  mutable unsigned long long seen_ = 0;                  // generation of the solver state the buffer holds
  const FirstOrderLoikOptimized* owner_ = nullptr;       // set by the solver that references the data object
  const unsigned long long* generation_ = nullptr;
};

// caller-owned result / state object (public members, as upstream).  Instance b, joint i (1..nb), component k:
//   z[b*nv + j], nu[...], w[...];  vis[(b*nb + (i-1))*6 + k], fis likewise;  yis[(b*nc + c)*6 + k], Aty likewise;
//   His[(b*nb + (i-1))*21 + sym(r,c)] (upper triangle, row-major: H_i is symmetric);  pis like vis;
//   liMi[(b*nb + (i-1))*12 + ..] = R row-major, t.
// z, nu, w, vis, fis, yis are copied after every solve (selectable: FirstOrderLoikOptimized::set_fetch); His, pis, Aty, liMi
// are fetched on first access.
struct IkIdDataOptimized {
  IkIdDataOptimized(const Model& model, int num_eq_c_, int batch_ = 1)
  : batch(batch_), nb(model.njoints - 1), nv(model.nv), num_eq_c(num_eq_c_),
    nu(static_cast<std::size_t>(batch_) * model.nv, 0.0), z(nu.size(), 0.0), w(nu.size(), 0.0),
    vis(static_cast<std::size_t>(batch_) * (model.njoints - 1) * 6, 0.0), fis(vis.size(), 0.0),
    yis(static_cast<std::size_t>(batch_) * num_eq_c_ * 6, 0.0),
    His(LOIKB_F_HIS, static_cast<std::size_t>(batch_) * (model.njoints - 1) * 21),
    pis(LOIKB_F_PIS, static_cast<std::size_t>(batch_) * (model.njoints - 1) * 6),
    Aty(LOIKB_F_ATY, static_cast<std::size_t>(batch_) * num_eq_c_ * 6),
    liMi(LOIKB_F_LIMI, static_cast<std::size_t>(batch_) * (model.njoints - 1) * 12)
  {
  }
  int batch, nb, nv, num_eq_c;
  DVec nu, z, w;  // z is the answer: box-projected joint velocity (loik-loid-optimized.hpp:333)
  DVec vis, fis, yis;
  LazyField His, pis, Aty, liMi;
  // full symmetric 6x6 of link i (1..nb) of instance b out of the packed His
  Mat6x6 His_full(int i, int b = 0) const
  {
    const DVec& h = His.get();
    const double* p = h.data() + (static_cast<std::size_t>(b) * nb + (i - 1)) * 21;
    Mat6x6 m{};
    int k = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c, ++k) { m[6 * r + c] = p[k]; m[6 * c + r] = p[k]; }
    return m;
  }
};

class FirstOrderLoikOptimized {
public:
  using IkIdData = IkIdDataOptimized;

  // the reference's 17 constructor arguments, same order (loik-loid-optimized.hpp:129-134), then batch / device
  FirstOrderLoikOptimized(const int max_iter, const double& tol_abs, const double& tol_rel, const double& tol_primal_inf,
                          const double& tol_dual_inf, const double& rho, const double& mu,
                          const double& mu_equality_scale_factor, const ADMMPenaltyUpdateStrat& mu_update_strat,
                          const int num_eq_c, const int eq_c_dim, const Model& model, IkIdData& ik_id_data,
                          const bool warm_start, const double tol_tail_solve, const bool verbose, const bool logging,
                          const int device = 0, const int flags = 0, const int eq_c_capacity = 0)
  : model_(model), ik_id_data_(ik_id_data), batch_(ik_id_data.batch), nc_(num_eq_c)
  {
    loikb_options o{};
    o.max_iter = max_iter; o.tol_abs = tol_abs; o.tol_rel = tol_rel; o.tol_primal_inf = tol_primal_inf;
    o.tol_dual_inf = tol_dual_inf; o.rho = rho; o.mu = mu; o.mu_equality_scale_factor = mu_equality_scale_factor;
    o.mu_update_strat = mu_update_strat; o.num_eq_c = num_eq_c; o.eq_c_dim = eq_c_dim; o.warm_start = warm_start;
    o.tol_tail_solve = tol_tail_solve; o.verbose = verbose; o.logging = logging;
    o.batch = batch_; o.device = device; o.precision = LOIKB_F64; o.flags = flags;
    o.eq_c_capacity = eq_c_capacity;  // room for AddEqConstraint (0: num_eq_c slots, as upstream sizes yis / Aty)
    const loikb_model_desc d = model_.desc();
    if (loikb_version() != LOIKB_VERSION)  // (struct layouts of this header against the loaded library's)
      throw std::runtime_error("loik_amd: libloik_amd.so has ABI version " + std::to_string(loikb_version()) + ", this header is version " +
                               std::to_string(LOIKB_VERSION) + ": rebuild the library");
    check(loikb_create(&d, &o, &h_));
    max_iter_ = max_iter; rho_ = rho; tol_tail_solve_ = tol_tail_solve; tol_primal_inf_ = tol_primal_inf; tol_dual_inf_ = tol_dual_inf;
    for (LazyField* f : {&ik_id_data_.His, &ik_id_data_.pis, &ik_id_data_.Aty, &ik_id_data_.liMi}) {
      f->owner_ = this;
      f->generation_ = &generation_;
    }
  }
  ~FirstOrderLoikOptimized() { loikb_destroy(h_); }
  FirstOrderLoikOptimized(const FirstOrderLoikOptimized&) = delete;
  FirstOrderLoikOptimized& operator=(const FirstOrderLoikOptimized&) = delete;

  // loik-loid-optimized.hpp:335-361
  void SolveInit(const DVec& q, const Mat6x6& H_ref, const Motion& v_ref, const std::vector<Index>& active_task_constraint_ids,
                 const std::vector<Mat6x6>& Ais, const std::vector<Vec6>& bis, const DVec& lb, const DVec& ub)
  {
    Args a(*this, q, active_task_constraint_ids, Ais, bis, lb, ub);
    check(loikb_solve_init(h_, q.data(), H_ref.data(), v_ref.data(), a.ids.data(), (int)a.ids.size(), a.A.data(),
                           a.b.data(), lb.data(), ub.data(), a.nbound, a.flags));
  }
  // loik-loid-optimized.hpp:368-455
  void Solve()
  {
    check(loikb_solve(h_));
    solved();
  }
  // loik-loid-optimized.hpp:475-580
  void Solve(const DVec& q, const Mat6x6& H_ref, const Motion& v_ref, const std::vector<Index>& active_task_constraint_ids,
             const std::vector<Mat6x6>& Ais, const std::vector<Vec6>& bis, const DVec& lb, const DVec& ub)
  {
    Args a(*this, q, active_task_constraint_ids, Ais, bis, lb, ub);
    check(loikb_solve_full(h_, q.data(), H_ref.data(), v_ref.data(), a.ids.data(), (int)a.ids.size(), a.A.data(),
                           a.b.data(), lb.data(), ub.data(), a.nbound, a.flags));
    solved();
  }
  // loik-loid-optimized.hpp:596-695 (one Ai for the batch; bi per instance when bi.size() == batch)
  void Solve(const DVec& q, const Index c_id, const Mat6x6& Ai, const Vec6& bi)
  {
    int flags = LOIKB_A_SHARED | LOIKB_B_SHARED;
    check_q(q);
    if (batch_ > 1 && q.size() == static_cast<std::size_t>(model_.nq)) flags |= LOIKB_Q_SHARED;
    check(loikb_solve_tailored(h_, q.data(), (int)c_id, Ai.data(), bi.data(), flags));
    solved();
  }
  void Solve(const DVec& q, const Index c_id, const Mat6x6& Ai, const std::vector<Vec6>& bis)
  {
    DVec b(bis.size() * 6);
    for (std::size_t i = 0; i < bis.size(); ++i)
      for (int k = 0; k < 6; ++k) b[6 * i + k] = bis[i][k];
    int flags = LOIKB_A_SHARED;
    check_bis(bis);
    if (bis.size() == 1 && batch_ > 1) flags |= LOIKB_B_SHARED;
    check_q(q);
    if (batch_ > 1 && q.size() == static_cast<std::size_t>(model_.nq)) flags |= LOIKB_Q_SHARED;
    check(loikb_solve_tailored(h_, q.data(), (int)c_id, Ai.data(), b.data(), flags));
    solved();
  }

  // ---- IkProblemFormulationOptimized's editing methods (ik-id-description-optimized.hpp).  Upstream they sit behind the
  // protected problem_ member (loik-loid-optimized.hpp:765): a subclass reaches them; here they are public.  They act on the
  // problem the last SolveInit / Solve(q, H_ref, ...) set and stay in force until the next one.
  // UpdateReferences(H_refs, v_refs), :103-121: one weight and target per joint, index 0 = the universe
  void UpdateReferences(const std::vector<Mat6x6>& H_refs, const std::vector<Motion>& v_refs)
  {
    if (H_refs.size() != v_refs.size() || H_refs.size() != static_cast<std::size_t>(model_.njoints))
      throw std::runtime_error("[IkProblemFormulation::UpdateReferences]: input arguments 'H_refs', 'v_refs' have wrong size!!");
    DVec H(H_refs.size() * 36), v(v_refs.size() * 6);
    for (std::size_t i = 0; i < H_refs.size(); ++i) {
      for (int k = 0; k < 36; ++k) H[36 * i + k] = H_refs[i][k];
      for (int k = 0; k < 6; ++k) v[6 * i + k] = v_refs[i][k];
    }
    check(loikb_update_references(h_, H.data(), v.data(), (int)H_refs.size()));
  }
  // UpdateEqConstraint(c_id, Ai, bi) :178-218 and (c_id, bi) :224-238; bis: one per instance or one for the batch
  void UpdateEqConstraint(const Index c_id, const Mat6x6& Ai, const std::vector<Vec6>& bis)
  {
    const DVec b = flat(bis);
    check(loikb_update_eq_constraint(h_, (int)c_id, Ai.data(), b.data(), LOIKB_A_SHARED | b_flag(bis)));
  }
  void UpdateEqConstraint(const Index c_id, const std::vector<Vec6>& bis)
  {
    const DVec b = flat(bis);
    check(loikb_update_eq_constraint(h_, (int)c_id, nullptr, b.data(), b_flag(bis)));
  }
  // AddEqConstraint, :244-286 (needs a free slot: constructor argument eq_c_capacity)
  void AddEqConstraint(const Index c_id, const Mat6x6& Ai, const std::vector<Vec6>& bis)
  {
    const DVec b = flat(bis);
    check(loikb_add_eq_constraint(h_, (int)c_id, Ai.data(), b.data(), LOIKB_A_SHARED | b_flag(bis)));
    resize_constraints();
  }
  // RemoveEqConstraint, :292-319; false: nothing to remove (upstream prints a warning)
  bool RemoveEqConstraint(const Index c_id)
  {
    const int rc = loikb_remove_eq_constraint(h_, (int)c_id);
    if (rc < 0) check(rc);
    resize_constraints();
    return rc == LOIKB_OK;
  }
  std::vector<Index> active_task_constraint_ids() const
  {
    std::vector<int> ids(static_cast<std::size_t>(loikb_eq_c_capacity(h_)) + 1);
    const int n = loikb_active_constraint_ids(h_, ids.data(), (int)ids.size());
    return std::vector<Index>(ids.begin(), ids.begin() + n);
  }
  // solve on the edited set without rewriting a constraint (not upstream, whose tailored Solve always updates one)
  void Solve(const DVec& q)
  {
    int flags = 0;
    check_q(q);
    if (batch_ > 1 && q.size() == static_cast<std::size_t>(model_.nq)) flags |= LOIKB_Q_SHARED;
    check(loikb_solve_tailored(h_, q.data(), -1, nullptr, nullptr, flags));
    solved();
  }

  // ---- pass-level public methods (loik-loid-optimized.hpp:192-264; the reference's component-wise test calls them one by one,
  // tests/loik-loid.cpp:305-556).  They run on the plain one-instance-per-thread implementation behind loikb_pass (a debug
  // path; the Solve() overloads use the fused kernels); after each call the data object is refreshed like after a solve.
  void FwdPass1() { pass(LOIKB_PASS_FWD_PASS1); }
  void BwdPassOptimizedVisitor() { pass(LOIKB_PASS_BWD_PASS); }
  void FwdPass2OptimizedVisitor() { pass(LOIKB_PASS_FWD_PASS2); }
  void BoxProj() { pass(LOIKB_PASS_BOX_PROJ); }
  void DualUpdate() { pass(LOIKB_PASS_DUAL_UPDATE); }
  void ComputeResiduals() { pass(LOIKB_PASS_COMPUTE_RESIDUALS); }
  void CheckConvergence() { pass(LOIKB_PASS_CHECK_CONVERGENCE); }
  void CheckFeasibility() { pass(LOIKB_PASS_CHECK_FEASIBILITY); }
  void UpdateMu() { pass(LOIKB_PASS_UPDATE_MU); }
  // iter_ = i; ik_id_data_.UpdatePrev(); ik_id_data_.ResetInfNorms()  -- the head of a loop iteration (hpp:381-388)
  void BeginIteration() { pass(LOIKB_PASS_BEGIN_ITERATION); }

  // ---- outer loop on the device (not in the reference class: its callers -- a sampling planner, README.md:5 --
  //      integrate on the host and pass a new q to the tailored Solve every step; here q stays resident in HBM)
  // q <- q (+) dt * z, z = the answer of the last solve
  void Integrate(double dt) { check(loikb_integrate(h_, dt)); }
  // tailored Solve (loik-loid-optimized.hpp:596-695) on the resident q
  void Solve(const Index c_id, const Mat6x6& Ai, const std::vector<Vec6>& bis)
  {
    DVec b(bis.size() * 6);
    for (std::size_t i = 0; i < bis.size(); ++i)
      for (int k = 0; k < 6; ++k) b[6 * i + k] = bis[i][k];
    int flags = LOIKB_A_SHARED;
    check_bis(bis);
    if (bis.size() == 1 && batch_ > 1) flags |= LOIKB_B_SHARED;
    check(loikb_solve_tailored(h_, nullptr, (int)c_id, Ai.data(), b.data(), flags));
    solved();
  }
  // the resident configurations, [batch][nq]
  DVec q_resident() const
  {
    DVec q((std::size_t)batch_ * model_.nq);
    check(loikb_get(h_, LOIKB_F_Q, q.data(), 0));
    return q;
  }

  // ---- the logged lists of instance b (constructor argument logging = true; such a solver runs on the plain pass-by-pass
  //      implementation: "logging residuals, should be disabled for speed", loik-loid-optimized.hpp:408)
  LoikSolverInfo get_solver_info(int b = 0) const
  {
    LoikSolverInfo info;
    // (rows per instance as the library stored them: max_iter - 1 at the time of that solve, whatever set_max_iter did since)
    const std::size_t cap = static_cast<std::size_t>(std::max(loikb_solver_info_rows_cap(h_), 1));
    DVec buf(static_cast<std::size_t>(batch_) * cap);
    std::vector<int> rows(static_cast<std::size_t>(batch_));
    std::vector<double>* lists[] = {&info.primal_residual_task_list_, &info.primal_residual_slack_list_, &info.primal_residual_list_,
                                    &info.dual_residual_nu_list_, &info.dual_residual_v_list_, &info.dual_residual_list_,
                                    &info.mu_list_, &info.mu_eq_list_, &info.mu_ineq_list_};
    for (int l = 0; l < LOIKB_LOG_NLIST; ++l) {
      check(loikb_get_solver_info(h_, l, buf.data(), static_cast<int>(cap), rows.data()));
      lists[l]->assign(buf.begin() + b * cap, buf.begin() + b * cap + rows[b]);
    }
    const int n_iter = get_iter(b), n_tail = n_iter - rows[b];   // the tail solve extends iter_list_ only (hpp:286-290)
    for (int i = 1; i <= n_iter; ++i) info.iter_list_.push_back(i);
    for (int i = 1; i <= n_tail; ++i) info.tail_solve_iter_list_.push_back(i);
    return info;
  }

  // ---- which members of the data object every solve copies back (default: all six, as upstream's callers expect).
  // FETCH_NONE leaves the results on the device: read them with loikb_get(handle(), field, ptr, LOIKB_OUT_DEVICE) or call
  // fetch_now(mask) when needed -- a planner that only integrates on the device (Integrate) never needs the copy.
  enum : unsigned { FETCH_NONE = 0, FETCH_Z = 1, FETCH_NU = 2, FETCH_W = 4, FETCH_VIS = 8, FETCH_FIS = 16, FETCH_YIS = 32,
                    FETCH_ALL = 63 };
  void set_fetch(unsigned mask) { fetch_mask_ = mask; }
  unsigned get_fetch() const { return fetch_mask_; }
  void fetch_now(unsigned mask) { fetch(mask); }

  // task-solver-base.hpp:87-141 (instance index defaults to 0: the single-instance reading).  The per-instance scalars of
  // a solve are downloaded once, on the first getter call after it, and served from that copy: O(1) per call.
  int get_iter(int b = 0) const { return geti(LOIKB_F_ITER, b); }
  double get_primal_residual(int b = 0) const { return getd(LOIKB_F_PRIMAL_RESIDUAL, b); }
  double get_dual_residual(int b = 0) const { return getd(LOIKB_F_DUAL_RESIDUAL, b); }
  bool get_convergence_status(int b = 0) const { return geti(LOIKB_F_CONVERGED, b) != 0; }
  bool get_primal_infeasibility_status(int b = 0) const { return geti(LOIKB_F_PRIMAL_INFEASIBLE, b) != 0; }
  bool get_dual_infeasibility_status(int = 0) const { return false; }  // never set by the optimized solver upstream
  double get_mu(int b = 0) const { return getd(LOIKB_F_MU, b); }
  double get_rho() const { return rho_; }
  // set_tol_primal / set_tol_dual (task-solver-base.hpp:118-127) write tol_primal_ / tol_dual_, which CheckConvergence
  // recomputes at every iteration (hxx:544-552): as upstream, the value set is what the getter returns until the next solve
  double get_tol_primal(int b = 0) const { return tol_primal_set_ ? tol_primal_ : getd(LOIKB_F_TOL_PRIMAL, b); }
  double get_tol_dual(int b = 0) const { return tol_dual_set_ ? tol_dual_ : getd(LOIKB_F_TOL_DUAL, b); }
  void set_tol_primal(const double t) { tol_primal_ = t; tol_primal_set_ = true; }
  void set_tol_dual(const double t) { tol_dual_ = t; tol_dual_set_ = true; }
  double get_tol_primal_inf() const { return tol_primal_inf_; }
  double get_tol_dual_inf() const { return tol_dual_inf_; }
  void set_max_iter(const int max_iter) { check(loikb_set_max_iter(h_, max_iter)); max_iter_ = max_iter; }
  void set_rho(const double rho) { rho_ = rho; check(loikb_set_rho(h_, rho)); }
  void set_mu(const double mu) { check(loikb_set_mu(h_, mu)); }
  void set_tol_primal_inf(const double t) { tol_primal_inf_ = t; check(loikb_set_tol_primal_inf(h_, t)); }
  // dual infeasibility is never evaluated on this path: the tolerance is stored, as upstream stores it
  void set_tol_dual_inf(const double t) { tol_dual_inf_ = t; }
  // loik-loid-optimized.hpp:700-755
  // get_primal_residual_vec() / get_dual_residual_vec(), loik-loid-optimized.hpp:698-699: [6 nb + nv] of instance b
  DVec get_primal_residual_vec(int b = 0) const { return getvec(LOIKB_F_PRIMAL_RESIDUAL_VEC, b); }
  DVec get_dual_residual_vec(int b = 0) const { return getvec(LOIKB_F_DUAL_RESIDUAL_VEC, b); }
  double get_dual_residual_v(int b = 0) const { return getd(LOIKB_F_DUAL_RESIDUAL_V, b); }
  double get_dual_residual_nu(int b = 0) const { return getd(LOIKB_F_DUAL_RESIDUAL_NU, b); }
  double get_tol_tail_solve() const { return tol_tail_solve_; }
  void set_tol_tail_solve(const double tol) { tol_tail_solve_ = tol; check(loikb_set_tol_tail_solve(h_, tol)); }
  double get_delta_x_qp_inf_norm(int b = 0) const { return getd(LOIKB_F_DELTA_X_QP_INF_NORM, b); }
  double get_delta_z_qp_inf_norm(int b = 0) const { return getd(LOIKB_F_DELTA_Z_QP_INF_NORM, b); }
  double get_delta_y_qp_inf_norm(int b = 0) const { return getd(LOIKB_F_DELTA_Y_QP_INF_NORM, b); }
  double get_A_qp_T_delta_y_qp_inf_norm(int b = 0) const { return getd(LOIKB_F_A_QP_T_DELTA_Y_QP_INF_NORM, b); }
  double get_ub_qp_T_delta_y_qp_plus(int b = 0) const { return getd(LOIKB_F_UB_QP_T_DELTA_Y_QP_PLUS, b); }
  double get_lb_qp_T_delta_y_qp_minus(int b = 0) const { return getd(LOIKB_F_LB_QP_T_DELTA_Y_QP_MINUS, b); }
  bool get_primal_infeasibility_cond_1(int b = 0) const { return getd(LOIKB_F_PRIMAL_INFEASIBILITY_COND_1, b) != 0.0; }
  bool get_primal_infeasibility_cond_2(int b = 0) const { return getd(LOIKB_F_PRIMAL_INFEASIBILITY_COND_2, b) != 0.0; }

  loikb_stats stats() const
  {
    loikb_stats s{};
    loikb_get_stats(h_, &s);
    return s;
  }
  loikb_solver* handle() const { return h_; }

private:
  struct Args {
    std::vector<int> ids;
    DVec A, b;
    int flags = 0, nbound = 0;
    Args(const FirstOrderLoikOptimized& s, const DVec& q, const std::vector<Index>& cids, const std::vector<Mat6x6>& Ais,
         const std::vector<Vec6>& bis, const DVec& lb, const DVec& ub)
    {
      // same consistency checks, same messages as ik-id-description-optimized.hpp:132-151, :328-335
      if (lb.size() != ub.size())
        throw std::runtime_error("[IkProblemFormulation::UpdateIneqConstraints]: lower bound and upper bound have different dimensions!!!");
      const std::size_t nc = cids.size(), B = static_cast<std::size_t>(s.batch_);
      if (!(Ais.size() == nc || Ais.size() == nc * B) || !(bis.size() == nc || bis.size() == nc * B))
        throw std::runtime_error("[IkProblemFormulation::UpdateEqConstraints]: task_constraint_ids, Ais, and bis have different size !!!");
      for (Index c : cids) ids.push_back(static_cast<int>(c));
      A.resize(Ais.size() * 36);
      for (std::size_t i = 0; i < Ais.size(); ++i)
        for (int k = 0; k < 36; ++k) A[36 * i + k] = Ais[i][k];
      b.resize(bis.size() * 6);
      for (std::size_t i = 0; i < bis.size(); ++i)
        for (int k = 0; k < 6; ++k) b[6 * i + k] = bis[i][k];
      if (Ais.size() == nc) flags |= LOIKB_A_SHARED;
      if (bis.size() == nc && B > 1) flags |= LOIKB_B_SHARED;
      const std::size_t nv = static_cast<std::size_t>(s.model_.nv);
      if (lb.size() == nv * B && B > 1) nbound = (int)nv;
      else { nbound = (int)lb.size(); flags |= LOIKB_BOUNDS_SHARED; }
      const std::size_t nq = static_cast<std::size_t>(s.model_.nq);
      if (q.size() != nq && q.size() != nq * B)   // (the C-ABI takes bare pointers: sizes are checked here)
        throw std::runtime_error("loik_amd: q must hold model.nq values (one configuration for the batch) or batch * model.nq");
      if (B > 1 && q.size() == nq) flags |= LOIKB_Q_SHARED;
    }
  };

  void check_q(const DVec& q) const
  {
    const std::size_t nq = static_cast<std::size_t>(model_.nq);
    if (q.size() != nq && q.size() != nq * static_cast<std::size_t>(batch_))
      throw std::runtime_error("loik_amd: q must hold model.nq values (one configuration for the batch) or batch * model.nq");
  }
  static DVec flat(const std::vector<Vec6>& bis)
  {
    DVec b(bis.size() * 6);
    for (std::size_t i = 0; i < bis.size(); ++i)
      for (int k = 0; k < 6; ++k) b[6 * i + k] = bis[i][k];
    return b;
  }
  void check_bis(const std::vector<Vec6>& bis) const
  {
    if (bis.size() != 1 && bis.size() != static_cast<std::size_t>(batch_))
      throw std::runtime_error("loik_amd: bis must hold one target (for the whole batch) or one per instance");
  }
  int b_flag(const std::vector<Vec6>& bis) const { check_bis(bis); return (bis.size() == 1 && batch_ > 1) ? LOIKB_B_SHARED : 0; }
  // yis / Aty of the data object follow nc_eq_ (upstream never resizes them: its AddEqConstraint is deactivated)
  void resize_constraints()
  {
    nc_ = loikb_num_eq_c(h_);
    ik_id_data_.num_eq_c = nc_;
    ik_id_data_.yis.assign(static_cast<std::size_t>(batch_) * nc_ * 6, 0.0);
    ik_id_data_.Aty.buf_.assign(static_cast<std::size_t>(batch_) * nc_ * 6, 0.0);
    ++generation_;
  }

  void check(int rc) const
  {
    if (rc == LOIKB_OK) return;
    const char* msg = (rc <= LOIKB_ERR_ARG || rc == LOIKB_ERR_MODEL) ? loikb_last_error() : loikb_status_string(rc);
    throw std::runtime_error(msg && *msg ? msg : loikb_status_string(rc));
  }
  void pass(int id)
  {
    check(loikb_pass(h_, id));
    solved();
  }
  void solved()
  {
    ++generation_;  // lazy members and cached scalars of the previous solve are stale now
    tol_primal_set_ = tol_dual_set_ = false;
    fetch(fetch_mask_);
  }
  void fetch(unsigned mask)
  {
    // (one call for all of them -- loikb_get_results: one gather launch and one synchronisation for a small batch; the FETCH_* bits are LOIKB_RES_*)
    static_assert(FETCH_Z == LOIKB_RES_Z && FETCH_NU == LOIKB_RES_NU && FETCH_W == LOIKB_RES_W && FETCH_VIS == LOIKB_RES_VIS &&
                  FETCH_FIS == LOIKB_RES_FIS && FETCH_YIS == LOIKB_RES_YIS, "fetch mask out of step with loikb_get_results");
    if (nc_ <= 0) mask &= ~static_cast<unsigned>(FETCH_YIS);
    mask &= FETCH_ALL;
    // A small batch (the fused gather of loikb_get_results will serve it): the scalars behind get_iter(), get_convergence_status(), the
    // residuals ... ride along -- a scalar getter's own device call costs 0.025 ms, one problem's callers make two or three after every solve.
    // A large batch keeps them lazy (one download per field that is actually asked for).
    const std::size_t per = 3 * static_cast<std::size_t>(model_.nv) + 12 * static_cast<std::size_t>(model_.njoints - 1) +
                            6 * static_cast<std::size_t>(nc_ > 0 ? nc_ : 0) + LOIKB_RES_NSCALARS;
    const bool with_scalars = mask != 0 && sizeof(double) * per * static_cast<std::size_t>(batch_) <= LOIKB_RESULTS_FUSED_BYTES_DEFAULT;
    if (with_scalars) scal_.resize(static_cast<std::size_t>(batch_) * LOIKB_RES_NSCALARS);
    if (mask)
      check(loikb_get_results(h_, mask | (with_scalars ? LOIKB_RES_SCALARS : 0u), ik_id_data_.z.data(), ik_id_data_.nu.data(), ik_id_data_.w.data(),
                              ik_id_data_.vis.data(), ik_id_data_.fis.data(), ik_id_data_.yis.data(), with_scalars ? scal_.data() : nullptr));
    if (with_scalars) {
      const std::size_t B = static_cast<std::size_t>(batch_);
      for (int k = 0; k < LOIKB_RES_SCALAR_ITER; ++k) {   // the 30 scalar fields, LOIKB_F_PRIMAL_RESIDUAL + k
        Cached<DVec>& c = dcache_[LOIKB_F_PRIMAL_RESIDUAL + k];
        c.data.resize(B);
        for (std::size_t b = 0; b < B; ++b) c.data[b] = scal_[b * LOIKB_RES_NSCALARS + static_cast<std::size_t>(k)];
        c.gen = generation_;
      }
      auto fill_int = [&](int field, int col, int mask_bits) {
        Cached<std::vector<int>>& c = icache_[field];
        c.data.resize(B);
        for (std::size_t b = 0; b < B; ++b) {
          const int v = static_cast<int>(scal_[b * LOIKB_RES_NSCALARS + static_cast<std::size_t>(col)]);
          c.data[b] = mask_bits ? ((v & mask_bits) ? 1 : 0) : v;
        }
        c.gen = generation_;
      };
      fill_int(LOIKB_F_ITER, LOIKB_RES_SCALAR_ITER, 0);
      fill_int(LOIKB_F_STATUS, LOIKB_RES_SCALAR_STATUS, 0);
      fill_int(LOIKB_F_CONVERGED, LOIKB_RES_SCALAR_STATUS, 1);
      fill_int(LOIKB_F_PRIMAL_INFEASIBLE, LOIKB_RES_SCALAR_STATUS, 2);
      fill_int(LOIKB_F_MU_UPDATES, LOIKB_RES_SCALAR_MU_UPDATES, 0);
    }
  }
  // one download per field and solve, then O(1) per getter call
  template <typename V>
  struct Cached { V data; unsigned long long gen = ~0ull; };
  int geti(int field, int b) const
  {
    Cached<std::vector<int>>& c = icache_[field];
    if (c.gen != generation_) {
      c.data.resize(static_cast<std::size_t>(batch_));
      check(loikb_get(h_, field, c.data.data(), 0));
      c.gen = generation_;
    }
    return c.data[static_cast<std::size_t>(b)];
  }
  double getd(int field, int b) const
  {
    Cached<DVec>& c = dcache_[field];
    if (c.gen != generation_) {
      c.data.resize(static_cast<std::size_t>(batch_));
      check(loikb_get(h_, field, c.data.data(), 0));
      c.gen = generation_;
    }
    return c.data[static_cast<std::size_t>(b)];
  }
  DVec getvec(int field, int b) const
  {
    const std::size_t n = 6 * static_cast<std::size_t>(model_.njoints - 1) + static_cast<std::size_t>(model_.nv);
    Cached<DVec>& c = dcache_[field];
    if (c.gen != generation_) {
      c.data.resize(static_cast<std::size_t>(batch_) * n);
      check(loikb_get(h_, field, c.data.data(), 0));
      c.gen = generation_;
    }
    return DVec(c.data.begin() + b * n, c.data.begin() + (b + 1) * n);
  }
  friend class LazyField;
  void lazy_fetch(int field, double* dst) const { check(loikb_get(h_, field, dst, 0)); }

  Model model_;             // by value, as upstream (loik-loid-optimized.hpp:762)
  IkIdData& ik_id_data_;    // caller-owned, must outlive the solver (loik-loid-optimized.hpp:763)
  loikb_solver* h_ = nullptr;
  int batch_, nc_;
  int max_iter_ = 0;
  double rho_ = 0.0, tol_tail_solve_ = 0.0, tol_primal_inf_ = 0.0, tol_dual_inf_ = 0.0;
  double tol_primal_ = 0.0, tol_dual_ = 0.0;
  bool tol_primal_set_ = false, tol_dual_set_ = false;
  unsigned fetch_mask_ = FETCH_ALL;
  DVec scal_;   // [batch][LOIKB_RES_NSCALARS]: the scalar block of the last fused fetch
  unsigned long long generation_ = 1;
  mutable std::map<int, Cached<std::vector<int>>> icache_;
  mutable std::map<int, Cached<DVec>> dcache_;
};

inline const DVec& LazyField::get() const
{
  if (owner_ && generation_ && seen_ != *generation_) {
    owner_->lazy_fetch(field_, buf_.data());
    seen_ = *generation_;
  }
  return buf_;
}

}  // namespace loik_amd
