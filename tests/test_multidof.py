"""Multi-DoF joints: free-flyer, spherical and translation joints.

The reference gets them from Pinocchio's joint variant (visitors /root/reference/include/loik/loik-loid-optimized.hxx:
21-23, :91-93; `calc_aba` with an nv x nv `Dinv`).  The device solves an all-1-DoF tree in which such a joint is a chain
of 1-DoF joints about the axes of one frame with massless intermediate links; eliminating the joint's coordinates one
at a time is the same block elimination, so every iterate is the same up to rounding.  The CPU tests here prove that
on the oracle (true nv x nv joints vs the chain), the GPU tests compare the product with the true multi-DoF oracle.
"""
import numpy as np
import pytest

import loik_amd
from helpers import (FIXTURE, J_FREEFLYER, J_SPHERICAL, J_TRANSLATION, assert_close, assert_end_to_end, expand_to_chains,
                     fetch_end_to_end,
                     random_tree_multidof)
from loik_amd import workloads
from oracle import ref


def one_problem(model, seed, link=None, bound=0.5):
    link = model.njoints - 1 if link is None else link
    wl = workloads.make_workload(model, 1, link, seed, bound=bound, snap_prob=0.2, nu_scale=0.4)
    return dict(q=wl["q"][0], H_ref=np.eye(6), v_ref=np.zeros(6), c_ids=wl["c_ids"], Ais=wl["Ais"], bis=wl["bis"][0],
                lb=wl["lb"], ub=wl["ub"])


# the two eliminations are the same arithmetic only up to rounding, and f = H v + p cancels digits (H ~ mu_eq ~ 1e2..1e4)
TOL = 1e-7

CASES = [dict(seed=3, nb=6, root_freeflyer=True, n_spherical=0, n_translation=0),
         dict(seed=5, nb=9, root_freeflyer=True, n_spherical=1, n_translation=1),
         dict(seed=8, nb=12, root_freeflyer=False, n_spherical=2, n_translation=1),
         dict(seed=11, nb=14, root_freeflyer=True, n_spherical=2, n_translation=2)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_chain_of_massless_links_reproduces_the_multidof_joint(case):
    model = random_tree_multidof(**case)
    assert model.nv > model.njoints - 1
    p = one_problem(model, case["seed"])
    prm = dict(FIXTURE, max_iter=40, tol_abs=1e-9, tol_rel=0.0)
    a = ref.RefSolver(model, **prm)
    m1, q1, link_of = expand_to_chains(model, p["q"])
    assert m1.nv == model.nv and m1.njoints - 1 == model.nv
    b = ref.RefSolver(m1, **prm)
    a.SolveInit(p["q"], p["H_ref"], p["v_ref"], p["c_ids"], p["Ais"], p["bis"], p["lb"], p["ub"])
    b.SolveInit(q1, p["H_ref"], p["v_ref"], link_of[p["c_ids"]], p["Ais"], p["bis"], p["lb"], p["ub"])
    compared = 0
    for it in range(1, 30):
        # once the residuals are rounding noise the mu rule (a ratio of the two) is decided by that noise
        if it > 1 and min(a.scalar("primal_residual"), a.scalar("dual_residual")) < 1e-7:
            break
        compared += 1
        for s in (a, b):
            s.IterationBody()
            s.CheckConvergence()
            if it > 1:
                s.CheckFeasibility()
            s.UpdateMu()
        for name in ("nu", "z", "w", "Stf_plus_w"):
            assert_close(getattr(a, name), getattr(b, name), TOL, "%s @%d" % (name, it))
        for name in ("vis", "fis", "g", "pis"):
            assert_close(getattr(a, name)[1:], getattr(b, name)[link_of[1:]], TOL, "%s @%d" % (name, it))
        assert_close(a.His[1:], b.His[link_of[1:]], TOL, "His @%d" % it)
        for name in ("primal_residual", "dual_residual", "mu", "delta_y_qp_inf_norm", "A_qp_T_delta_y_qp_inf_norm",
                     "delta_fis_inf_norm", "delta_vis_inf_norm", "delta_nu_inf_norm", "Href_v_inf_norm", "g_inf_norm",
                     "Stf_plus_w_inf_norm", "nu_inf_norm", "tol_primal", "tol_dual", "ub_qp_T_delta_y_qp_plus",
                     "lb_qp_T_delta_y_qp_minus"):
            assert_close(a.scalar(name), b.scalar(name), TOL, "%s @%d" % (name, it))
    assert compared >= 3


@pytest.mark.parametrize("case", CASES[:2], ids=lambda c: "seed%d" % c["seed"])
def test_multidof_solve_satisfies_the_task(case):
    """first principles: the converged answer moves the task link as asked and respects the box"""
    model = random_tree_multidof(**case)
    p = one_problem(model, case["seed"] + 100)
    s = ref.RefSolver(model, **dict(FIXTURE, max_iter=400, tol_abs=1e-8, tol_rel=0.0))
    s.Solve(p["q"], p["H_ref"], p["v_ref"], p["c_ids"], p["Ais"], p["bis"], p["lb"], p["ub"])
    assert s.get_convergence_status()
    z = s.z
    v = workloads.link_velocity(model, p["q"][None], z[None], int(p["c_ids"][0]))[0]
    assert np.max(np.abs(v - p["bis"][0])) < 1e-6
    assert np.all(z <= p["ub"] + 1e-9) and np.all(z >= p["lb"] - 1e-9)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_recursive_and_dense_oracles_agree_on_multidof_joints(case):
    """the reference's relational pin (optimized == plain at 1e-10, tests/loik-loid.cpp:305-556) with nv_i x nv_i
    joints: oracle/loik_ref.c (recursive, Cholesky Dinv) against oracle/dense.py (dense QP residuals, numpy inverse)"""
    from oracle import dense
    model = random_tree_multidof(**case)
    p = one_problem(model, case["seed"] + 50)
    prm = dict(FIXTURE, max_iter=8, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    opt, pl = ref.RefSolver(model, **prm), dense.DenseSolver(model, **prm)
    args = (p["q"], p["H_ref"], p["v_ref"], p["c_ids"], p["Ais"], p["bis"], p["lb"], p["ub"])
    opt.Solve(*args); pl.Solve(*args)
    assert opt.get_iter() == pl.get_iter() == 7
    for n in ("nu", "z", "w"):
        assert_close(getattr(opt, n), getattr(pl, n), 1e-9, n)
    assert_close(opt.vis[1:], pl.vis[1:], 1e-9, "vis")
    assert_close(opt.fis[1:], pl.fis[1:], 1e-8, "fis")
    assert_close(opt.His[1:], pl.His[1:], 1e-8, "His")
    assert_close(opt.scalar("primal_residual"), pl.primal_residual, 1e-9, "primal")
    assert_close(opt.scalar("dual_residual"), pl.dual_residual, 1e-8, "dual")
    assert_close(opt.dual_residual_vec, pl.dual_residual_vec, 1e-8, "dual vec")


@pytest.mark.parametrize("case", CASES[:3], ids=lambda c: "seed%d" % c["seed"])
def test_multidof_oracle_answer_is_the_qp_optimum(case):
    """pins the oracle's nv x nv joints themselves: the converged answer is the optimum of the reduced dense QP
    min_nu sum_i 1/2 |J_i nu|^2  s.t.  J_c nu = b, lb <= nu <= ub  found by scipy's SLSQP, with Jacobians built by an
    independent numpy kinematics (workloads.link_velocity)"""
    from scipy.optimize import minimize
    model = random_tree_multidof(**case)
    p = one_problem(model, case["seed"] + 300)
    s = ref.RefSolver(model, **dict(FIXTURE, max_iter=4000, tol_abs=1e-9, tol_rel=0.0, tol_primal_inf=1e-12))
    s.Solve(p["q"], p["H_ref"], p["v_ref"], p["c_ids"], p["Ais"], p["bis"], p["lb"], p["ub"])
    assert s.get_convergence_status(), s.get_iter()
    nu, z, link = s.nu, s.z, int(p["c_ids"][0])

    def jac(i):
        J = np.zeros((6, model.nv))
        for k in range(model.nv):
            e = np.zeros((1, model.nv)); e[0, k] = 1.0
            J[:, k] = workloads.link_velocity(model, p["q"][None], e, i)[0]
        return J
    Js = [jac(i) for i in range(1, model.njoints)]
    for i in range(1, model.njoints):
        assert np.max(np.abs(Js[i - 1] @ nu - s.vis[i])) < 1e-11
    bb = p["bis"][0]
    assert np.max(np.abs(s.vis[link] - bb)) < 1e-8 and np.max(np.abs(nu - z)) < 1e-8
    Hq = sum(J.T @ J for J in Js)
    Jc = Js[link - 1]
    # the task link may hang off a short chain (rank-deficient J_c, target still reachable): keep independent rows
    U, sv, _ = np.linalg.svd(Jc)
    Ur = U[:, sv > 1e-9 * sv[0]]
    Jc, bb = Ur.T @ Jc, Ur.T @ bb
    res = minimize(lambda x: 0.5 * x @ Hq @ x, np.zeros(model.nv), jac=lambda x: Hq @ x, method="SLSQP",
                   bounds=list(zip(p["lb"], p["ub"])),
                   constraints=[dict(type="eq", fun=lambda x: Jc @ x - bb, jac=lambda x: Jc)],
                   options=dict(ftol=1e-15, maxiter=800))
    assert res.success
    assert abs(0.5 * z @ Hq @ z - res.fun) < 1e-7 * max(1.0, abs(res.fun))
    assert np.max(np.abs(z - res.x)) < 5e-4


# =====================================================================================================================
# GPU: the product (chains on the device, invisible to the caller) against the TRUE multi-DoF oracle
# =====================================================================================================================
LINK_FIELDS = ["vis", "fis", "g"]
DOF_FIELDS = ["nu", "z", "w", "Stf_plus_w"]
GPU_SCALARS = ["primal_residual", "dual_residual", "primal_residual_task", "primal_residual_slack", "dual_residual_v",
               "dual_residual_nu", "mu", "delta_fis_inf_norm", "delta_yis_inf_norm", "delta_w_inf_norm",
               "delta_vis_inf_norm", "delta_nu_inf_norm", "Av_inf_norm", "nu_inf_norm", "Href_v_inf_norm", "g_inf_norm",
               "Stf_plus_w_inf_norm"]


def _batch(model, B, seed, link=None):
    link = model.njoints - 1 if link is None else link
    return workloads.make_workload(model, B, link, seed, bound=0.5, snap_prob=0.2, nu_scale=0.4)


def _gpu(model, wl, prm, **kw):
    s = loik_amd.BatchedLoik(model, wl["q"].shape[0], **prm, **kw)
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    return s


def _args(wl, b):
    return (wl["q"][b], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"][b], wl["lb"], wl["ub"])


def _compare_k_iterations(model, wl, k, tol, step=7, **kw):
    prm = dict(FIXTURE, max_iter=k + 1, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    s = _gpu(model, wl, prm, **kw)
    B = wl["q"].shape[0]
    got = {n: s.get(n) for n in LINK_FIELDS + DOF_FIELDS + GPU_SCALARS + ["yis", "Aty", "liMi", "pis"]}
    got["His"] = s.His_full()
    resvec = {n: s.get(n) for n in ("primal_residual_vec", "dual_residual_vec")}
    assert np.all(s.get("iter") == k)
    assert got["vis"].shape == (B, model.njoints - 1, 6) and got["z"].shape == (B, model.nv)
    for b in range(0, B, step):
        r = ref.RefSolver(model, **prm)
        r.Solve(*_args(wl, b))
        assert_close(got["liMi"][b], r.liMi[1:], 1e-13, "liMi")
        for n in LINK_FIELDS:
            assert_close(got[n][b], r.field(n)[1:], tol, "%s b%d k%d" % (n, b, k))
        for n in DOF_FIELDS + ["yis", "Aty"]:
            assert_close(got[n][b], r.field(n), tol, "%s b%d k%d" % (n, b, k))
        assert_close(got["His"][b], r.His[1:], tol, "His b%d" % b)
        if kw.get("flags", 0) & loik_amd.capi.OPT_NO_H_CACHE:
            assert_close(got["pis"][b], r.pis[1:], tol, "pis b%d" % b)
        for n in GPU_SCALARS:
            assert_close(got[n][b], r.scalar(n), tol, "%s b%d k%d" % (n, b, k))
        for n in ("primal_residual_vec", "dual_residual_vec"):  # public getters, loik-loid-optimized.hpp:698-699
            assert_close(resvec[n][b], r.field(n), tol, "%s b%d k%d" % (n, b, k))
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_gpu_random_trees_with_multidof_joints(case):
    model = random_tree_multidof(**case)
    wl = _batch(model, 80, case["seed"] + 200)
    for k in (1, 3, 6):
        _compare_k_iterations(model, wl, k, TOL)
    _compare_k_iterations(model, wl, 4, TOL, flags=loik_amd.capi.OPT_NO_H_CACHE)
    # end to end with the stopping logic, through the tail kernel as well
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"],
                          wl["ub"], nthreads=4, want_nu=True, **prm)
    for kw in (dict(), dict(tail_max_instances=-1)):
        s = _gpu(model, wl, prm, **kw)
        assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=1e-8, what="multidof seed %d" % case["seed"])
        s.close()


@pytest.mark.gpu
def test_gpu_floating_base_talos():
    """the floating-base humanoid of SURVEY.md 8(f) rank 2: free-flyer root_joint + 32 revolute joints"""
    model = loik_amd.builtin_model("talos32_freeflyer")
    assert (model.njoints, model.nq, model.nv) == (34, 39, 38)
    link = model.getJointId("arm_left_7_joint")
    wl = _batch(model, 200, 77, link)
    _compare_k_iterations(model, wl, 2, TOL, step=23)
    _compare_k_iterations(model, wl, 5, TOL, step=23)
    prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"],
                          wl["ub"], nthreads=4, **prm)
    s = _gpu(model, wl, prm)
    assert_end_to_end(fetch_end_to_end(s, nu=False), out, prm, same_frac=0.99, ztol=1e-8, what="floating-base talos")
    # the answer moves the wrist as asked: first principles, independent of both solvers
    ok = s.get("converged").astype(bool)
    assert ok.mean() > 0.5
    v = workloads.link_velocity(model, wl["q"], s.get("z"), link)
    assert np.max(np.abs(v - wl["bis"][:, 0])[ok]) < 1e-5
    s.close()


def _np_quat_mul(a, b):
    return np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1],
                     a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
                     a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3],
                     a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])


def _np_integrate(model, q, v):
    """pinocchio::integrate for the supported joints, written with scipy's rotation exponential"""
    from scipy.spatial.transform import Rotation
    out = q.copy()
    for i in range(1, model.njoints):
        t, iq, iv = int(model.jtype[i]), int(model.idx_q[i]), int(model.idx_v[i])
        if t == J_FREEFLYER:
            R0 = Rotation.from_quat(q[iq + 3:iq + 7])
            w, vl = v[iv + 3:iv + 6], v[iv:iv + 3]
            th = np.linalg.norm(w)
            K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            V = np.eye(3) + ((1 - np.cos(th)) / th ** 2) * K + ((th - np.sin(th)) / th ** 3) * K @ K
            out[iq:iq + 3] = q[iq:iq + 3] + R0.apply(V @ vl)
            qn = (R0 * Rotation.from_rotvec(w)).as_quat()
            out[iq + 3:iq + 7] = qn if qn @ q[iq + 3:iq + 7] >= 0 else -qn
        elif t == J_SPHERICAL:
            qn = (Rotation.from_quat(q[iq:iq + 4]) * Rotation.from_rotvec(v[iv:iv + 3])).as_quat()
            out[iq:iq + 4] = qn if qn @ q[iq:iq + 4] >= 0 else -qn
        elif t == J_TRANSLATION:
            out[iq:iq + 3] = q[iq:iq + 3] + v[iv:iv + 3]
        else:
            out[iq] = q[iq] + v[iv]
    return out


@pytest.mark.gpu
def test_gpu_integrate_on_the_configuration_manifold():
    """outer loop with quaternion joints: q <- q (+) dt z is the Lie-group update"""
    model = random_tree_multidof(seed=5, nb=9, root_freeflyer=True, n_spherical=1, n_translation=1)
    wl = _batch(model, 70, 123)
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0)
    s = _gpu(model, wl, prm)
    z = s.get("z")
    dt = 0.37
    s.integrate(dt)
    q1 = s.get("q")
    for b in range(0, 70, 3):
        want = _np_integrate(model, wl["q"][b], dt * z[b])
        assert np.max(np.abs(q1[b] - want)) < 1e-12, b
    # the quaternions stay unit, and a second solve from the resident configurations equals a solve from q1
    for i in range(1, model.njoints):
        t, iq = int(model.jtype[i]), int(model.idx_q[i])
        if t in (J_FREEFLYER, J_SPHERICAL):
            o = iq + (3 if t == J_FREEFLYER else 0)
            assert np.max(np.abs(np.linalg.norm(q1[:, o:o + 4], axis=1) - 1)) < 1e-12
    link = int(wl["c_ids"][0])
    s.Solve(None, link, wl["Ais"], wl["bis"])
    t2 = loik_amd.BatchedLoik(model, 70, **prm)
    t2.Solve(q1, wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    assert np.max(np.abs(s.get("liMi") - t2.get("liMi"))) < 1e-14
    s.close(); t2.close()


# ---- JointModelSphericalZYX (q-dependent motion subspace), JointModelPlanar, JointModelRUBX/Y/Z --------------------------------
NEW_CASES = [dict(seed=21, nb=8, root_freeflyer=False, n_spherical=0, n_translation=0, n_zyx=2, n_rub=2),
             dict(seed=22, nb=10, root_freeflyer=False, n_spherical=1, n_translation=0, n_zyx=1, n_planar=1, n_rub=1, root_planar=True),
             dict(seed=23, nb=13, root_freeflyer=True, n_spherical=0, n_translation=1, n_zyx=2, n_planar=1, n_rub=3),
             # JointModelRevoluteUnboundedUnaligned (LOIKB_J_RUBU): (cos, sin) about an arbitrary axis
             dict(seed=24, nb=11, root_freeflyer=False, n_spherical=1, n_translation=0, n_zyx=0, n_planar=0, n_rub=1, n_rubu=3)]


@pytest.mark.parametrize("case", NEW_CASES, ids=lambda c: "seed%d" % c["seed"])
def test_oracle_zyx_planar_unbounded_joints_solve_the_qp(case):
    """pins the oracle's new joint types by first principles: the converged answer is SLSQP's optimum of the reduced dense QP
    whose Jacobians come from an independent numpy kinematics (workloads.link_velocity: M(q) and S(q) of
    JointModelSphericalZYX::calc / JointModelPlanar::calc written out again)"""
    from scipy.optimize import minimize
    model = random_tree_multidof(**case)
    assert model.nq > model.nv >= model.njoints - 1
    p = one_problem(model, case["seed"] + 300)
    s = ref.RefSolver(model, **dict(FIXTURE, max_iter=4000, tol_abs=1e-9, tol_rel=0.0, tol_primal_inf=1e-12))
    s.Solve(p["q"], p["H_ref"], p["v_ref"], p["c_ids"], p["Ais"], p["bis"], p["lb"], p["ub"])
    assert s.get_convergence_status(), s.get_iter()
    nu, link = s.z, int(p["c_ids"][0])
    nv = model.nv
    eye = np.eye(nv)
    J = [np.stack([workloads.link_velocity(model, p["q"][None], eye[k][None], i)[0] for k in range(nv)], axis=1)
         for i in range(1, model.njoints)]
    # kinematic consistency of the oracle's sweep with the independent Jacobians (incl. S(q) of the ZYX joints)
    for i in range(1, model.njoints):
        assert_close(s.vis[i], J[i - 1] @ s.nu, 1e-9, "v_%d = J nu" % i)
    cost = lambda x: 0.5 * sum(float((Ji @ x) @ (Ji @ x)) for Ji in J)
    grad = lambda x: sum(Ji.T @ (Ji @ x) for Ji in J)
    Jc = J[link - 1]
    res = minimize(cost, np.zeros(nv), jac=grad, method="SLSQP", bounds=list(zip(p["lb"], p["ub"])),
                   constraints=[dict(type="eq", fun=lambda x: Jc @ x - p["bis"][0], jac=lambda x: Jc)],
                   options=dict(ftol=1e-14, maxiter=500))
    assert res.success
    assert abs(cost(nu) - res.fun) < 1e-6 * max(1.0, res.fun) and np.max(np.abs(Jc @ nu - p["bis"][0])) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("case", NEW_CASES, ids=lambda c: "seed%d" % c["seed"])
def test_gpu_zyx_planar_unbounded_joints(case):
    """the device's chains (RZ-RY-RX with their own angles; PX-PY-RZ of one frame) and (cos, sin) joints against the oracle's
    true joints: k iterations and end to end, every engine that applies"""
    model = random_tree_multidof(**case)
    wl = _batch(model, 90, case["seed"] + 200)
    for k in (1, 3, 6):
        _compare_k_iterations(model, wl, k, TOL)
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"],
                          wl["ub"], nthreads=4, want_nu=True, **prm)
    for kw in (dict(), dict(tail_max_instances=-1)):
        s = _gpu(model, wl, prm, **kw)
        assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=1e-8, what="new joints seed %d" % case["seed"])
        # liMi of the caller's joints (a ZYX joint's is the product over its chain)
        r = ref.RefSolver(model, **prm)
        r.Solve(*[wl[k][0] if k in ("q", "bis") else wl[k] for k in ("q", "H_ref", "v_ref", "c_ids", "Ais", "bis", "lb", "ub")])
        assert_close(s.get("liMi")[0], r.liMi[1:], 1e-13, "liMi")
        s.close()


@pytest.mark.gpu
def test_gpu_integrate_planar_and_unbounded_joints():
    """loikb_integrate on the (x, y, cos, sin) and (cos, sin) configuration spaces: SpecialEuclideanOperationTpl<2> /
    SpecialOrthogonalOperationTpl<2>::integrate written out in numpy"""
    model = random_tree_multidof(seed=31, nb=7, root_freeflyer=False, n_spherical=0, n_translation=0, n_zyx=1, n_planar=1, n_rub=2,
                                 root_planar=True)
    B = 50
    wl = _batch(model, B, 31)
    prm = dict(FIXTURE, max_iter=100, tol_abs=1e-6, tol_rel=0.0)
    s = _gpu(model, wl, prm)
    z, q0, dt = s.get("z"), wl["q"].copy(), 0.37
    s.integrate(dt)
    q1 = s.get("q")
    want = q0.copy()
    for i in range(1, model.njoints):
        t, iq, iv = int(model.jtype[i]), int(model.idx_q[i]), int(model.idx_v[i])
        if t == 13:
            vx, vy, w = (dt * z[:, iv + k] for k in range(3))
            c0, s0 = q0[:, iq + 2], q0[:, iq + 3]
            sw, cw = np.sin(w), np.cos(w)
            with np.errstate(divide="ignore", invalid="ignore"):
                tx = np.where(np.abs(w) > 1e-14, (sw * vx - (1 - cw) * vy) / w, vx)
                ty = np.where(np.abs(w) > 1e-14, ((1 - cw) * vx + sw * vy) / w, vy)
            want[:, iq] = q0[:, iq] + c0 * tx - s0 * ty
            want[:, iq + 1] = q0[:, iq + 1] + s0 * tx + c0 * ty
            c1, s1 = c0 * cw - s0 * sw, s0 * cw + c0 * sw
            n = 0.5 * (3 - (c1 * c1 + s1 * s1))
            want[:, iq + 2], want[:, iq + 3] = c1 * n, s1 * n
        elif t in (14, 15, 16):
            w = dt * z[:, iv]
            c0, s0 = q0[:, iq], q0[:, iq + 1]
            c1, s1 = c0 * np.cos(w) - s0 * np.sin(w), s0 * np.cos(w) + c0 * np.sin(w)
            n = 0.5 * (3 - (c1 * c1 + s1 * s1))
            want[:, iq], want[:, iq + 1] = c1 * n, s1 * n
        else:
            nvj = 3 if t == 12 else 1
            want[:, iq:iq + nvj] = q0[:, iq:iq + nvj] + dt * z[:, iv:iv + nvj]
    assert np.abs(z).max() > 1e-3
    assert_close(q1, want, 1e-13, "q after integrate")
    s.close()
