// C++ parity test of the drop-in wrapper (include/loik_amd/loik.hpp) against the CPU oracle (oracle/loik_ref.h),
// written the way the reference's own Boost tests are (tests/loik-loid.cpp): same fixture, same call sequences.
//   test_loik_solve_split (:261-302), test_1st_order_loik_optimized_correctness (:559-671, repeated Solve()),
//   test_1st_order_loik_tailored_timing (:1035-1078, iteration count stable over repeats), throw sites.
// Exit code 0 = all checks passed.  Needs a GPU.
#include "loik_amd/loik.hpp"
#include "../../oracle/loik_ref.h"

#include <cmath>
#include <cstdio>
#include <cstring>

using namespace loik_amd;

static int failures = 0;
#define CHECK(cond)                                                              \
  do {                                                                           \
    if (!(cond)) { ++failures; std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)

// check_eigen_dense_abs_or_rel_equal / check_scalar_abs_or_rel_equal of the reference (tests/loik-loid.cpp:39-83)
static bool close(double a, double b, double tol = 1e-9)
{
  const double d = std::fabs(a - b);
  return d < tol || (d / std::fabs(a) < tol && d / std::fabs(b) < tol);
}
static bool close(const double* a, const double* b, int n, double tol = 1e-9)
{
  for (int i = 0; i < n; ++i)
    if (!close(a[i], b[i], tol)) return false;
  return true;
}

struct Fixture {  // ProblemSetupFixture, tests/loik-loid.cpp:87-165
  int max_iter = 2;
  double tol_abs = 1e-3, tol_rel = 1e-3, tol_primal_inf = 1e-2, tol_dual_inf = 1e-2, tol_tail_solve = 1e-1, rho = 1e-5,
         mu = 1e-2, mu_equality_scale_factor = 1e4;
  ADMMPenaltyUpdateStrat mu_update_strat = DEFAULT;
  int num_eq_c = 1, eq_c_dim = 6;
  bool warm_start = false, verbose = false, logging = false;
  Model robot_model;
  DVec q;
  Mat6x6 H_ref = Identity6();
  Motion v_ref{};
  std::vector<Index> active_task_constraint_ids;
  std::vector<Mat6x6> Ais;
  std::vector<Vec6> bis;
  double bound_magnitude = 4.0;
  DVec lb, ub;
  explicit Fixture(const char* robot = "talos32") : robot_model(Model::Builtin(robot))
  {
    q.assign(robot_model.nq, 0.0);  // pinocchio::neutral (unit quaternion for a free-flyer)
    for (int i = 1; i < robot_model.njoints; ++i)
      if (robot_model.jtype[i] == LOIKB_J_FREEFLYER) q[robot_model.idx_q[i] + 6] = 1.0;
    active_task_constraint_ids.push_back(static_cast<Index>(robot_model.njoints - 1));
    Ais.push_back(Identity6());
    Vec6 bi{};
    bi[2] = 0.5;
    bis.push_back(bi);
    set_bound(4.0);
  }
  void set_bound(double b)
  {
    bound_magnitude = b;
    lb.assign(robot_model.nv, -b);
    ub.assign(robot_model.nv, b);
  }
};

struct Oracle {
  ref_solver* s = nullptr;
  explicit Oracle(const Fixture& f, int eq_c_capacity = 0)
  {
    const loikb_model_desc d = f.robot_model.desc();
    ref_model m{d.njoints, d.nq, d.nv, d.parents, d.jtype, d.axis, d.idx_q, d.idx_v, d.placement};
    ref_params p{f.max_iter, f.tol_abs, f.tol_rel, f.tol_primal_inf, f.tol_dual_inf, f.rho, f.mu, f.mu_equality_scale_factor,
                 (int)f.mu_update_strat, f.num_eq_c, f.eq_c_dim, (int)f.warm_start, f.tol_tail_solve, eq_c_capacity};
    ref_create(&m, &p, &s);
  }
  ~Oracle() { ref_destroy(s); }
  void Solve(const Fixture& f, const DVec& q, const Vec6& bi)
  {
    int id = (int)f.active_task_constraint_ids[0];
    ref_solve_full(s, q.data(), f.H_ref.data(), f.v_ref.data(), &id, 1, f.Ais[0].data(), bi.data(), f.lb.data(), f.ub.data(),
                   (int)f.lb.size());
  }
  const double* field(int which) const { int n; return ref_field(s, which, &n); }
};

static void compare(const Fixture& f, const IkIdDataOptimized& d, FirstOrderLoikOptimized& solver, Oracle& o)
{
  const int nv = f.robot_model.nv, nb = f.robot_model.njoints - 1;
  CHECK(close(d.z.data(), o.field(REF_F_Z), nv));
  CHECK(close(d.nu.data(), o.field(REF_F_NU), nv));
  {
    const DVec pr = solver.get_primal_residual_vec(), du = solver.get_dual_residual_vec();
    CHECK((int)pr.size() == 6 * (f.robot_model.njoints - 1) + nv && pr.size() == du.size());
    CHECK(close(pr.data(), o.field(REF_F_PRIMAL_RES_VEC), (int)pr.size()));
    CHECK(close(du.data(), o.field(REF_F_DUAL_RES_VEC), (int)du.size()));
  }
  CHECK(close(d.w.data(), o.field(REF_F_W), nv));
  CHECK(close(d.vis.data(), o.field(REF_F_VIS) + 6, 6 * nb));
  CHECK(close(d.fis.data(), o.field(REF_F_FIS) + 6, 6 * nb));
  CHECK(close(d.yis.data(), o.field(REF_F_YIS), 6));
  // members fetched on first access (upstream's tests read His / pis: tests/loik-loid.cpp:597-615)
  CHECK(close(d.Aty.data(), o.field(REF_F_ATY), 6));
  CHECK(close(d.liMi.data(), o.field(REF_F_LIMI) + 12, 12 * nb, 1e-14));
  for (int i = 1; i <= nb; ++i) {
    const Mat6x6 H = d.His_full(i);
    CHECK(close(H.data(), o.field(REF_F_HIS) + 36 * i, 36));
  }
  CHECK(solver.get_iter() == (int)ref_scalar(o.s, REF_S_ITER));
  CHECK(solver.get_convergence_status() == (ref_scalar(o.s, REF_S_CONVERGED) != 0));
  CHECK(solver.get_primal_infeasibility_status() == (ref_scalar(o.s, REF_S_PRIMAL_INFEASIBLE) != 0));
  CHECK(close(solver.get_primal_residual(), ref_scalar(o.s, REF_S_PRIMAL_RESIDUAL)));
  CHECK(close(solver.get_dual_residual(), ref_scalar(o.s, REF_S_DUAL_RESIDUAL)));
  CHECK(close(solver.get_mu(), ref_scalar(o.s, REF_S_MU), 1e-14));
  CHECK(close(solver.get_tol_primal(), ref_scalar(o.s, REF_S_TOL_PRIMAL)));
  CHECK(close(solver.get_tol_dual(), ref_scalar(o.s, REF_S_TOL_DUAL)));
}

#define MAKE_SOLVER(name, data, f)                                                                                       \
  FirstOrderLoikOptimized name{f.max_iter, f.tol_abs, f.tol_rel, f.tol_primal_inf, f.tol_dual_inf, f.rho, f.mu,          \
                               f.mu_equality_scale_factor, f.mu_update_strat, f.num_eq_c, f.eq_c_dim, f.robot_model, data, \
                               f.warm_start, f.tol_tail_solve, f.verbose, f.logging}

int main()
{
  if (loikb_device_count() < 1) { std::printf("no GPU\n"); return 2; }
  {  // test_loik_solve_split
    Fixture f; f.max_iter = 200; f.set_bound(5.0);
    IkIdDataOptimized d1(f.robot_model, f.num_eq_c), d2(f.robot_model, f.num_eq_c);
    MAKE_SOLVER(s1, d1, f); MAKE_SOLVER(s2, d2, f);
    s1.Solve(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    s2.SolveInit(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    s2.Solve();
    CHECK(d1.nu == d2.nu); CHECK(d1.z == d2.z); CHECK(d1.w == d2.w);
    CHECK(s1.get_iter() == s2.get_iter());
  }
  for (auto cfg : {std::pair<int, double>{8, 2.0}, {100, 2.0}, {200, 1.0}, {2, 1.0}}) {  // correctness + reset
    Fixture f; f.max_iter = cfg.first; f.set_bound(cfg.second);
    IkIdDataOptimized d(f.robot_model, f.num_eq_c);
    MAKE_SOLVER(solver, d, f);
    Oracle o(f);
    o.Solve(f, f.q, f.bis[0]);
    solver.SolveInit(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    for (int rep = 0; rep < 3; ++rep) {  // repeatedly call Solve() and check against ground truth (:592-669)
      solver.Solve();
      compare(f, d, solver, o);
    }
  }
  {  // a reachable target on the left wrist, tailored warm-started entry, iteration count stable over repeats
    Fixture f; f.max_iter = 300; f.tol_abs = 1e-6; f.tol_rel = 0.0; f.set_bound(0.5); f.warm_start = false;
    f.active_task_constraint_ids[0] = f.robot_model.getJointId("arm_left_7_joint");
    for (int k = 0; k < f.robot_model.nq; ++k) f.q[k] = 0.1 * std::sin(1.0 + k);
    f.bis[0] = Vec6{0.05, -0.03, 0.02, 0.01, 0.02, -0.04};
    IkIdDataOptimized d(f.robot_model, f.num_eq_c);
    MAKE_SOLVER(solver, d, f);
    Oracle o(f);
    o.Solve(f, f.q, f.bis[0]);
    solver.SolveInit(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    int it0 = -1;
    for (int rep = 0; rep < 20; ++rep) {
      solver.Solve(f.q, f.active_task_constraint_ids[0], f.Ais[0], f.bis[0]);
      if (it0 < 0) it0 = solver.get_iter();
      CHECK(solver.get_iter() == it0);
    }
    // (whether this hand-made target converges is the oracle's call: compare() checks the flags against it)
    compare(f, d, solver, o);
  }
  {  // outer loop on the device: Integrate(dt) + tailored Solve on the resident q  ==  integrating on the host and
     // passing the new q (what a caller of the reference does every planner step)
    Fixture f; f.max_iter = 300; f.tol_abs = 1e-6; f.tol_rel = 0.0; f.set_bound(0.5); f.warm_start = true;
    f.active_task_constraint_ids[0] = f.robot_model.getJointId("arm_left_7_joint");
    for (int k = 0; k < f.robot_model.nq; ++k) f.q[k] = 0.1 * std::sin(1.0 + k);
    f.bis[0] = Vec6{0.05, -0.03, 0.02, 0.01, 0.02, -0.04};
    IkIdDataOptimized dh(f.robot_model, f.num_eq_c), dd(f.robot_model, f.num_eq_c);
    MAKE_SOLVER(host_loop, dh, f);
    MAKE_SOLVER(dev_loop, dd, f);
    host_loop.SolveInit(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    dev_loop.SolveInit(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    DVec q = f.q;
    const double dt = 0.05;
    for (int step = 0; step < 4; ++step) {
      Vec6 b = f.bis[0];
      for (int k = 0; k < 6; ++k) b[k] *= (1.0 - 0.2 * step);  // the target velocity decays as the planner closes in
      host_loop.Solve(q, f.active_task_constraint_ids[0], f.Ais[0], b);
      if (step > 0) dev_loop.Integrate(dt);
      dev_loop.Solve(f.active_task_constraint_ids[0], f.Ais[0], std::vector<Vec6>{b});
      // (q + dt z may be contracted to an fma on one side and not the other: equal to rounding, not bit for bit)
      CHECK(close(dh.z.data(), dd.z.data(), f.robot_model.nv));
      CHECK(close(dh.w.data(), dd.w.data(), f.robot_model.nv));
      CHECK(host_loop.get_iter() == dev_loop.get_iter());
      const DVec qr = dev_loop.q_resident();
      CHECK(close(qr.data(), q.data(), f.robot_model.nq));
      for (int k = 0; k < f.robot_model.nq; ++k) q[k] += dt * dh.z[k];
    }
  }
  {  // test_1st_order_loik_optimized_correctness_component_wise (tests/loik-loid.cpp:305-556): the passes one by one
    Fixture f; f.max_iter = 200; f.set_bound(1.0);
    IkIdDataOptimized d(f.robot_model, f.num_eq_c);
    MAKE_SOLVER(solver, d, f);
    Oracle o(f);
    int id = (int)f.active_task_constraint_ids[0];
    ref_solve_init(o.s, f.q.data(), f.H_ref.data(), f.v_ref.data(), &id, 1, f.Ais[0].data(), f.bis[0].data(), f.lb.data(),
                   f.ub.data(), (int)f.lb.size());
    solver.SolveInit(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    const int nv = f.robot_model.nv, nb = f.robot_model.njoints - 1;
    solver.FwdPass1(); ref_fwd_pass1(o.s);
    for (int i = 1; i <= nb; ++i) CHECK(close(d.His_full(i).data(), o.field(REF_F_HIS) + 36 * i, 36, 1e-10));
    CHECK(close(d.pis.data(), o.field(REF_F_PIS) + 6, 6 * nb, 1e-10));
    solver.BwdPassOptimizedVisitor(); ref_bwd_pass(o.s);
    for (int i = 1; i <= nb; ++i) CHECK(close(d.His_full(i).data(), o.field(REF_F_HIS) + 36 * i, 36, 1e-10));
    CHECK(close(d.pis.data(), o.field(REF_F_PIS) + 6, 6 * nb, 1e-10));
    solver.FwdPass2OptimizedVisitor(); ref_fwd_pass2(o.s);
    CHECK(close(d.nu.data(), o.field(REF_F_NU), nv, 1e-10));
    CHECK(close(d.vis.data(), o.field(REF_F_VIS) + 6, 6 * nb, 1e-10));
    CHECK(close(d.fis.data(), o.field(REF_F_FIS) + 6, 6 * nb, 1e-10));
    solver.BoxProj(); ref_box_proj(o.s);
    CHECK(close(d.nu.data(), o.field(REF_F_NU), nv, 1e-10)); CHECK(close(d.w.data(), o.field(REF_F_W), nv, 1e-10));
    CHECK(close(d.z.data(), o.field(REF_F_Z), nv, 1e-10));
    solver.DualUpdate(); ref_dual_update(o.s);
    CHECK(close(d.w.data(), o.field(REF_F_W), nv, 1e-10)); CHECK(close(d.yis.data(), o.field(REF_F_YIS), 6, 1e-10));
    solver.ComputeResiduals(); ref_compute_residuals(o.s);
    CHECK(close(solver.get_primal_residual(), ref_scalar(o.s, REF_S_PRIMAL_RESIDUAL), 1e-10));
    CHECK(close(solver.get_dual_residual(), ref_scalar(o.s, REF_S_DUAL_RESIDUAL), 1e-10));
    solver.CheckConvergence(); ref_check_convergence(o.s);
    CHECK(close(solver.get_tol_primal(), ref_scalar(o.s, REF_S_TOL_PRIMAL), 1e-12));
    CHECK(close(solver.get_tol_dual(), ref_scalar(o.s, REF_S_TOL_DUAL), 1e-12));
    CHECK(solver.get_convergence_status() == (ref_scalar(o.s, REF_S_CONVERGED) != 0));
    solver.CheckFeasibility(); ref_check_feasibility(o.s);
    CHECK(close(solver.get_delta_y_qp_inf_norm(), ref_scalar(o.s, REF_S_DELTA_Y_QP_INF_NORM), 1e-10));
    CHECK(close(solver.get_A_qp_T_delta_y_qp_inf_norm(), ref_scalar(o.s, REF_S_A_QP_T_DELTA_Y_QP_INF_NORM), 1e-10));
    CHECK(solver.get_primal_infeasibility_status() == (ref_scalar(o.s, REF_S_PRIMAL_INFEASIBLE) != 0));
    solver.UpdateMu(); ref_update_mu(o.s);
    CHECK(close(solver.get_mu(), ref_scalar(o.s, REF_S_MU), 1e-14));
  }
  {  // results left on the device (set_fetch), fetched on demand; O(1) getters; tolerance setters of the base class
    Fixture f; f.max_iter = 100; f.set_bound(2.0);
    IkIdDataOptimized d(f.robot_model, f.num_eq_c), dref(f.robot_model, f.num_eq_c);
    MAKE_SOLVER(solver, d, f); MAKE_SOLVER(sref, dref, f);
    sref.Solve(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    solver.set_fetch(FirstOrderLoikOptimized::FETCH_Z);
    solver.Solve(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    CHECK(d.z == dref.z);
    bool untouched = true;
    for (double x : d.nu) untouched = untouched && x == 0.0;
    CHECK(untouched);  // nu was not copied
    solver.fetch_now(FirstOrderLoikOptimized::FETCH_NU | FirstOrderLoikOptimized::FETCH_VIS);
    CHECK(d.nu == dref.nu); CHECK(d.vis == dref.vis);
    CHECK(DVec(d.pis) == DVec(dref.pis));
    for (int rep = 0; rep < 1000; ++rep) CHECK(solver.get_iter() == sref.get_iter());  // served from one download
    CHECK(solver.get_tol_primal_inf() == f.tol_primal_inf && solver.get_tol_dual_inf() == f.tol_dual_inf);
    solver.set_tol_dual_inf(0.5); CHECK(solver.get_tol_dual_inf() == 0.5);
    const double tp = solver.get_tol_primal();
    solver.set_tol_primal(123.0); CHECK(solver.get_tol_primal() == 123.0);
    solver.Solve(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    CHECK(solver.get_tol_primal() == tp);  // CheckConvergence recomputed it, as upstream (hxx:544-546)
  }
  {  // throw sites carry the reference's messages
    Fixture f; f.max_iter = 10;
    IkIdDataOptimized d(f.robot_model, f.num_eq_c);
    bool thrown = false;
    try {
      FirstOrderLoikOptimized bad{f.max_iter, f.tol_abs, f.tol_rel, f.tol_primal_inf, f.tol_dual_inf, f.rho, f.mu,
                                  f.mu_equality_scale_factor, f.mu_update_strat, f.num_eq_c, 3, f.robot_model, d,
                                  f.warm_start, f.tol_tail_solve, f.verbose, f.logging};
    } catch (const std::runtime_error& e) {
      thrown = std::strstr(e.what(), "equality constraint dimension is not 6") != nullptr;
    }
    CHECK(thrown);
    MAKE_SOLVER(solver, d, f);
    thrown = false;
    DVec lb_bad(f.lb.begin(), f.lb.end() - 1), ub_bad(f.ub.begin(), f.ub.end() - 1);
    try {
      solver.Solve(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, lb_bad, ub_bad);
    } catch (const std::runtime_error& e) {
      thrown = std::strstr(e.what(), "inequality constraint dimension") != nullptr || std::strstr(e.what(), "lb/ub") != nullptr;
    }
    CHECK(thrown);
    solver.Solve(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    thrown = false;
    try {
      solver.Solve(f.q, 3, f.Ais[0], f.bis[0]);
    } catch (const std::runtime_error& e) {
      thrown = std::strstr(e.what(), "constraint doesn't yet exist at link 'c_id'") != nullptr;
    }
    CHECK(thrown);
  }
  {  // batch of 3 through the C++ interface: shared q/A/box, per-instance targets
    Fixture f; f.max_iter = 300; f.tol_abs = 1e-6; f.tol_rel = 0.0; f.set_bound(0.5);
    f.active_task_constraint_ids[0] = f.robot_model.getJointId("arm_left_7_joint");
    std::vector<Vec6> bis = {Vec6{0.05, -0.03, 0.02, 0.01, 0.02, -0.04}, Vec6{-0.02, 0.04, 0.01, 0.0, -0.03, 0.02},
                             Vec6{0.01, 0.01, -0.05, 0.02, 0.0, 0.01}};
    IkIdDataOptimized d(f.robot_model, f.num_eq_c, 3);
    MAKE_SOLVER(solver, d, f);
    solver.Solve(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, bis, f.lb, f.ub);
    for (int b = 0; b < 3; ++b) {
      Oracle o(f);
      o.Solve(f, f.q, bis[b]);
      CHECK(solver.get_iter(b) == (int)ref_scalar(o.s, REF_S_ITER));
      CHECK(close(d.z.data() + b * f.robot_model.nv, o.field(REF_F_Z), f.robot_model.nv));
    }
  }
  {  // editing the formulation between solves (ik-id-description-optimized.hpp:103-121, :178-319): per-link references,
     // AddEqConstraint / RemoveEqConstraint with one spare constraint slot
    Fixture f; f.max_iter = 300; f.tol_abs = 1e-6; f.tol_rel = 0.0; f.set_bound(0.5);
    const Index left = f.robot_model.getJointId("arm_left_7_joint"), right = f.robot_model.getJointId("arm_right_7_joint");
    f.active_task_constraint_ids[0] = left;
    f.bis[0] = Vec6{0.05, -0.03, 0.02, 0.01, 0.02, -0.04};
    IkIdDataOptimized d(f.robot_model, f.num_eq_c);
    FirstOrderLoikOptimized solver{f.max_iter, f.tol_abs, f.tol_rel, f.tol_primal_inf, f.tol_dual_inf, f.rho, f.mu,
                                   f.mu_equality_scale_factor, f.mu_update_strat, f.num_eq_c, f.eq_c_dim, f.robot_model, d,
                                   f.warm_start, f.tol_tail_solve, f.verbose, f.logging, 0, 0, /*eq_c_capacity=*/2};
    Oracle o(f, 2);
    const int nj = f.robot_model.njoints, nv = f.robot_model.nv;
    int id = (int)left;
    solver.SolveInit(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    ref_solve_init(o.s, f.q.data(), f.H_ref.data(), f.v_ref.data(), &id, 1, f.Ais[0].data(), f.bis[0].data(), f.lb.data(),
                   f.ub.data(), nv);
    std::vector<Mat6x6> H_refs(nj, Identity6());
    std::vector<Motion> v_refs(nj, Motion{});
    for (int i = 0; i < nj; ++i)
      for (int k = 0; k < 6; ++k) { H_refs[i][7 * k] = 0.5 + 0.1 * ((i + k) % 7); v_refs[i][k] = 0.02 * std::sin(1.0 + i + 3 * k); }
    H_refs[4][1] = H_refs[4][6] = 0.05;  // one full (symmetric) weight
    solver.UpdateReferences(H_refs, v_refs);
    ref_update_references(o.s, H_refs[0].data(), v_refs[0].data(), nj);  // (vectors of std::array are contiguous)
    solver.Solve();
    ref_solve(o.s);
    CHECK(solver.get_iter() == (int)ref_scalar(o.s, REF_S_ITER));
    CHECK(close(d.z.data(), o.field(REF_F_Z), nv, 1e-8));
    CHECK(close(d.vis.data(), o.field(REF_F_VIS) + 6, 6 * (nj - 1), 1e-8));
    {
      bool thrown = false;
      try { solver.UpdateReferences(std::vector<Mat6x6>(nj - 1, Identity6()), std::vector<Motion>(nj - 1, Motion{})); }
      catch (const std::runtime_error& e) { thrown = std::strstr(e.what(), "have wrong size") != nullptr; }
      CHECK(thrown);
    }
    // a second task on the other wrist
    const Vec6 b2{-0.02, 0.04, 0.01, 0.0, -0.03, 0.02};
    solver.AddEqConstraint(right, Identity6(), {b2});
    CHECK(ref_add_eq_constraint(o.s, (int)right, Identity6().data(), b2.data()) == REF_OK);
    CHECK((solver.active_task_constraint_ids() == std::vector<Index>{left, right}));
    solver.Solve(f.q);
    ref_solve_tailored(o.s, f.q.data(), -1, nullptr, nullptr);
    CHECK(solver.get_iter() == (int)ref_scalar(o.s, REF_S_ITER));
    CHECK(close(d.z.data(), o.field(REF_F_Z), nv, 1e-8));
    CHECK(d.yis.size() == 12 && close(d.yis.data(), o.field(REF_F_YIS), 12, 1e-6));
    CHECK(d.Aty.size() == 12 && close(d.Aty.data(), o.field(REF_F_ATY), 12, 1e-6));
    {
      bool thrown = false;  // no third slot
      try { solver.AddEqConstraint(3, Identity6(), {b2}); } catch (const std::runtime_error&) { thrown = true; }
      CHECK(thrown);
    }
    // drop the first one: the second moves down with its dual
    CHECK(solver.RemoveEqConstraint(left));
    CHECK(!solver.RemoveEqConstraint(left));
    CHECK(ref_remove_eq_constraint(o.s, (int)left) == REF_OK);
    CHECK((solver.active_task_constraint_ids() == std::vector<Index>{right}));
    solver.UpdateEqConstraint(right, std::vector<Vec6>{f.bis[0]});
    CHECK(ref_update_eq_constraint(o.s, (int)right, nullptr, f.bis[0].data()) == REF_OK);
    solver.Solve(f.q);
    ref_solve_tailored(o.s, f.q.data(), -1, nullptr, nullptr);
    CHECK(solver.get_iter() == (int)ref_scalar(o.s, REF_S_ITER));
    CHECK(close(d.z.data(), o.field(REF_F_Z), nv, 1e-8));
    CHECK(d.yis.size() == 6 && close(d.yis.data(), o.field(REF_F_YIS), 6, 1e-6));
    CHECK(close(solver.get_primal_residual(), ref_scalar(o.s, REF_S_PRIMAL_RESIDUAL), 1e-7));
  }
  {  // logging = true: LoikSolverInfo (loik-loid-optimized.hpp:406-420) of the last solve, against the oracle's lists
    Fixture f; f.max_iter = 200; f.tol_abs = 1e-6; f.tol_rel = 0.0; f.set_bound(0.5); f.logging = true;
    f.active_task_constraint_ids[0] = f.robot_model.getJointId("arm_left_7_joint");
    f.bis[0] = Vec6{0.05, -0.03, 0.02, 0.01, 0.02, -0.04};
    IkIdDataOptimized d(f.robot_model, f.num_eq_c);
    MAKE_SOLVER(solver, d, f);
    solver.Solve(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    Oracle o(f);
    o.Solve(f, f.q, f.bis[0]);
    const LoikSolverInfo info = solver.get_solver_info();
    int n = 0;
    const double* pr = ref_solver_info(o.s, 2, &n);
    CHECK(solver.get_iter() == (int)ref_scalar(o.s, REF_S_ITER));
    CHECK(info.Size() == solver.get_iter() && (int)info.primal_residual_list_.size() == n && (int)info.mu_list_.size() == n);
    CHECK(n > 0 && close(info.primal_residual_list_.data(), pr, n));
    CHECK(close(info.dual_residual_list_.data(), ref_solver_info(o.s, 5, nullptr), n));
    CHECK(close(info.mu_list_.data(), ref_solver_info(o.s, 6, nullptr), n, 1e-14));
    CHECK(close(d.z.data(), o.field(REF_F_Z), f.robot_model.nv));
  }
  {  // floating base: free-flyer root_joint + 32 revolute joints, nq = 39, nv = 38
    Fixture f("talos32_freeflyer"); f.max_iter = 300; f.tol_abs = 1e-6; f.tol_rel = 0.0; f.set_bound(0.5);
    CHECK(f.robot_model.nq == 39 && f.robot_model.nv == 38 && f.robot_model.njoints == 34);
    f.active_task_constraint_ids[0] = f.robot_model.getJointId("arm_left_7_joint");
    for (int k = 7; k < f.robot_model.nq; ++k) f.q[k] = 0.1 * std::sin(1.0 + k);
    f.bis[0] = Vec6{0.05, -0.03, 0.02, 0.01, 0.02, -0.04};
    IkIdDataOptimized d(f.robot_model, f.num_eq_c);
    MAKE_SOLVER(solver, d, f);
    solver.Solve(f.q, f.H_ref, f.v_ref, f.active_task_constraint_ids, f.Ais, f.bis, f.lb, f.ub);
    Oracle o(f);  // the oracle solves the true 6-DoF joint (6 x 6 Dinv); the device a chain of six 1-DoF joints
    o.Solve(f, f.q, f.bis[0]);
    CHECK(solver.get_convergence_status());
    CHECK(solver.get_iter() == (int)ref_scalar(o.s, REF_S_ITER));
    CHECK(close(d.z.data(), o.field(REF_F_Z), f.robot_model.nv, 1e-7));
    CHECK(close(d.vis.data(), o.field(REF_F_VIS) + 6, 6 * (f.robot_model.njoints - 1), 1e-7));
  }
  std::printf(failures ? "%d CHECKS FAILED\n" : "all wrapper checks passed\n", failures);
  return failures ? 1 : 0;
}
