"""JointModelComposite ("... unaligned revolute/prismatic, composite"): a joint made of several 1-DoF joints
with their own placements and ONE body at the end.  The reference gets it from Pinocchio's joint variant
(/root/reference/include/loik/loik-loid-optimized.hxx:21-23, :91-93: jdata.S() is 6 x nv, calc_aba eliminates the nv x nv
block).  The oracle restates JointModelComposite::calc (M = prod P_k M_k, S_k seen from the last frame) and eliminates the
block like upstream; the device writes the joint out as the chain of its sub-joints with massless links.

CPU: the oracle's composite against (i) its own SphericalZYX joint (a composite RZ, RY, RX at one point IS that joint),
(ii) the massless chain of its sub-joints, (iii) scipy's SLSQP on the QP over the model's BODIES.  GPU: the device against
the oracle's true composite, every engine, plus integrate()."""
import numpy as np
import pytest
from scipy.optimize import minimize

import loik_amd
from helpers import FIXTURE, assert_close, assert_end_to_end, composite_tree, fetch_end_to_end, problem_args, random_tree
from loik_amd import workloads
from oracle import ref

J_RX, J_RY, J_RZ, J_PX, J_PY, J_PZ, J_RU, J_PU = 1, 2, 3, 4, 5, 6, 7, 8
J_SPHERICAL_ZYX, J_RUBX, J_COMPOSITE = 12, 14, 17
IDENT = np.concatenate([np.eye(3).ravel(), np.zeros(3)])


def chain_of(model):
    """the all-1-DoF model with massless links that the device solves (and that the oracle can also solve, with its
    test-only `massless` flags): returns (model1, link_of)"""
    ch = workloads._Chain(model)
    massless = np.zeros(ch.njoints, dtype=np.int32)
    bodies = set(ch.link_of)
    for j in range(1, ch.njoints):
        massless[j] = 0 if j in bodies else 1
    m1 = loik_amd.Model(ch.parents, ch.jtype, ch.axis, ch.placement, name=model.name + "_chain")
    m1.massless = massless
    assert m1.nq == model.nq and m1.nv == model.nv
    return m1, np.array(ch.link_of)


def batch_for(model, B, link, seed):
    return workloads.make_workload(model, B, link, seed, bound=0.5, snap_prob=0.0, nu_scale=0.4)


def test_composite_rz_ry_rx_is_the_spherical_zyx_joint():
    """JointModelSphericalZYX: R = Rz Ry Rx with S(q) -- the same joint as a composite of RZ, RY, RX with identity placements;
    the oracle implements the two separately (joint_calc / joint_S vs composite_calc)"""
    base = random_tree(5, 7)
    jt = base.jtype.copy(); jt[3] = J_SPHERICAL_ZYX
    zyx = loik_amd.Model(base.parents, jt, base.axis, base.placement)
    jt2 = base.jtype.copy(); jt2[3] = J_COMPOSITE
    comp = loik_amd.Model(base.parents, jt2, base.axis, base.placement,
                          composite={3: [(J_RZ, np.zeros(3), IDENT), (J_RY, np.zeros(3), IDENT), (J_RX, np.zeros(3), IDENT)]})
    assert comp.nq == zyx.nq and comp.nv == zyx.nv
    wl = batch_for(zyx, 3, zyx.njoints - 1, 4)
    prm = dict(FIXTURE, max_iter=40, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    for b in range(3):
        a, c = ref.RefSolver(zyx, **prm), ref.RefSolver(comp, **prm)
        a.Solve(*problem_args(wl, b)); c.Solve(*problem_args(wl, b))
        for n in ("nu", "z", "w", "vis", "fis", "liMi"):
            assert_close(getattr(c, n), getattr(a, n), 1e-12, n)


def test_oracle_composite_equals_its_massless_chain():
    model = composite_tree(21, 9, [2, 6])
    m1, link_of = chain_of(model)
    link = model.njoints - 1
    wl = batch_for(model, 3, link, 8)
    prm = dict(FIXTURE, max_iter=60, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    for b in range(3):
        t, c = ref.RefSolver(model, **prm), ref.RefSolver(m1, **prm)
        t.Solve(*problem_args(wl, b))
        c.Solve(wl["q"][b], wl["H_ref"], wl["v_ref"], np.array([link_of[link]], dtype=np.int32), wl["Ais"], wl["bis"][b], wl["lb"], wl["ub"])
        for n in ("nu", "z", "w"):
            assert_close(getattr(c, n), getattr(t, n), 1e-8, n)
        assert_close(c.vis[link_of[1:]], t.vis[1:], 1e-8, "vis of the bodies")
        assert_close(c.fis[link_of[1:]], t.fis[1:], 1e-7, "fis of the bodies")
        for n in ("primal_residual", "dual_residual", "delta_fis_inf_norm", "g_inf_norm"):
            assert_close(c.scalar(n), t.scalar(n), 1e-7, n)


def test_oracle_composite_solves_the_qp_over_the_bodies():
    model = composite_tree(33, 7, [3])
    link = model.njoints - 1
    wl = batch_for(model, 2, link, 17)

    def jac(q, i):
        J = np.zeros((6, model.nv))
        for k in range(model.nv):
            e = np.zeros((1, model.nv)); e[0, k] = 1.0
            J[:, k] = workloads.link_velocity(model, q[None], e, i)[0]
        return J
    for b in range(2):
        s = ref.RefSolver(model, **dict(FIXTURE, max_iter=4000, tol_abs=1e-9, tol_rel=0.0, tol_primal_inf=1e-12))
        s.Solve(*problem_args(wl, b))
        assert s.get_convergence_status(), s.get_iter()
        Js = [jac(wl["q"][b], i) for i in range(1, model.njoints)]
        for i in range(1, model.njoints):
            assert np.max(np.abs(Js[i - 1] @ s.nu - s.vis[i])) < 1e-10
        Hq = sum(J.T @ J for J in Js)
        A, bb = wl["Ais"][0], wl["bis"][b, 0]
        Jc = A @ Js[link - 1]
        res = minimize(lambda x: 0.5 * x @ Hq @ x, np.zeros(model.nv), jac=lambda x: Hq @ x, method="SLSQP",
                       bounds=list(zip(wl["lb"], wl["ub"])), constraints=[dict(type="eq", fun=lambda x: Jc @ x - bb, jac=lambda x: Jc)],
                       options=dict(ftol=1e-15, maxiter=500))
        assert res.success
        assert abs(0.5 * s.z @ Hq @ s.z - res.fun) < 1e-7 * max(1.0, abs(res.fun))
        assert np.max(np.abs(s.z - res.x)) < 2e-4


J_FREEFLYER, J_SPHERICAL, J_TRANSLATION, J_PLANAR = 9, 10, 11, 13
MULTIDOF_KINDS = [[J_TRANSLATION, J_SPHERICAL],        # the usual hand-made floating base: 3 + 3 DoF
                  [J_PLANAR, J_RY],                    # a planar base with a tilt joint
                  [J_RU, J_SPHERICAL_ZYX, J_PZ]]       # q-dependent subspace inside a composite


def test_oracle_composite_with_multidof_subjoints_equals_its_massless_chain():
    """JointModelComposite::addJoint takes any joint model: a composite of a translation and a spherical joint (a floating base
    assembled by hand), of a planar joint and a revolute one, ...  The oracle's composite_calc collects the nv_k columns of every
    sub-joint; the same model written as the chain of those sub-joints (each a TRUE multi-DoF joint of the oracle, massless
    links in between) takes another code path and must give the same iterates"""
    model = composite_tree(27, 9, [1, 4, 7], kinds=MULTIDOF_KINDS)
    assert model.nvs[1] == 6 and model.nqs[1] == 7 and model.nvs[4] == 4 and model.nqs[4] == 5 and model.nvs[7] == 5
    m1, link_of = chain_of(model)
    link = model.njoints - 1
    wl = batch_for(model, 3, link, 8)
    prm = dict(FIXTURE, max_iter=60, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    for b in range(3):
        t, c = ref.RefSolver(model, **prm), ref.RefSolver(m1, **prm)
        t.Solve(*problem_args(wl, b))
        c.Solve(wl["q"][b], wl["H_ref"], wl["v_ref"], np.array([link_of[link]], dtype=np.int32), wl["Ais"], wl["bis"][b], wl["lb"], wl["ub"])
        for n in ("nu", "z", "w"):
            assert_close(getattr(c, n), getattr(t, n), 1e-8, n)
        assert_close(c.vis[link_of[1:]], t.vis[1:], 1e-8, "vis of the bodies")
        assert_close(c.fis[link_of[1:]], t.fis[1:], 1e-7, "fis of the bodies")
        for n in ("primal_residual", "dual_residual", "delta_fis_inf_norm", "g_inf_norm"):
            assert_close(c.scalar(n), t.scalar(n), 1e-7, n)


def test_composite_with_a_zyx_subjoint_is_the_composite_of_its_three_revolute_joints():
    base = random_tree(15, 6)
    P = np.concatenate([np.eye(3).ravel(), [0.1, -0.2, 0.05]])
    jt = base.jtype.copy(); jt[2] = J_COMPOSITE
    a = loik_amd.Model(base.parents, jt, base.axis, base.placement, composite={2: [(J_PY, np.zeros(3), P), (J_SPHERICAL_ZYX, np.zeros(3), P)]})
    b_ = loik_amd.Model(base.parents, jt, base.axis, base.placement,
                        composite={2: [(J_PY, np.zeros(3), P), (J_RZ, np.zeros(3), P), (J_RY, np.zeros(3), IDENT), (J_RX, np.zeros(3), IDENT)]})
    wl = batch_for(a, 2, a.njoints - 1, 4)
    prm = dict(FIXTURE, max_iter=30, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    for b in range(2):
        x, y = ref.RefSolver(a, **prm), ref.RefSolver(b_, **prm)
        x.Solve(*problem_args(wl, b)); y.Solve(*problem_args(wl, b))
        for n in ("nu", "z", "w", "vis", "fis", "liMi"):
            assert_close(getattr(x, n), getattr(y, n), 1e-12, n)


def _rodrigues(a, th):
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def test_universal_joint_is_the_composite_of_its_two_revolute_joints():
    """JointModelUniversal(axis1, axis2) (nq = nv = 2): M = (R(axis1, q0) R(axis2, q1), 0), motion subspace
    S(q) = [R(axis2, q1)^T axis1 | axis2] (angular) -- the joint the composite of RevoluteUnaligned(axis1), RevoluteUnaligned(axis2)
    with identity placements describes, which is how the Pinocchio adapter hands it over (include/loik_amd/pinocchio_adapter.hpp)"""
    rng = np.random.default_rng(3)
    a1 = _unit_vec(rng); a2 = np.cross(a1, _unit_vec(rng)); a2 /= np.linalg.norm(a2)     # (orthogonal axes, as Pinocchio asserts)
    base = random_tree(9, 5)
    jt = base.jtype.copy(); jt[1] = J_COMPOSITE
    par = base.parents.copy(); par[1] = 0
    m = loik_amd.Model(par, jt, base.axis, base.placement, composite={1: [(J_RU, a1, IDENT), (J_RU, a2, IDENT)]})
    assert m.nqs[1] == 2 and m.nvs[1] == 2
    wl = batch_for(m, 2, m.njoints - 1, 11)
    prm = dict(FIXTURE, max_iter=10, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    for b in range(2):
        r = ref.RefSolver(m, **prm)
        r.Solve(*problem_args(wl, b))
        q0, q1 = wl["q"][b, 0], wl["q"][b, 1]
        R1, R2 = _rodrigues(a1, q0), _rodrigues(a2, q1)
        Pj = m.placement[1]
        assert_close(r.liMi[1][:9].reshape(3, 3), Pj[:9].reshape(3, 3) @ R1 @ R2, 1e-13, "M of the universal joint")
        assert_close(r.liMi[1][9:], Pj[9:], 1e-13, "no translation")
        w = np.column_stack([R2.T @ a1, a2]) @ r.nu[:2]          # joint 1 hangs on the universe: v_1 = S(q) nu_1
        assert_close(r.vis[1], np.concatenate([np.zeros(3), w]), 1e-12, "S(q) of the universal joint")


def _unit_vec(rng):
    a = rng.normal(size=3)
    return a / np.linalg.norm(a)


ENGINE_KW = {"default": {}, "solve_only": dict(tail_max_instances=-1), "tail_only": dict(tail_max_instances=1 << 20),
             "handover": dict(max_launch_iters=3, tail_max_instances=1 << 20)}


@pytest.mark.gpu
@pytest.mark.parametrize("engine", sorted(ENGINE_KW))
def test_gpu_composite_joints(engine, monkeypatch):
    if engine == "tail_only":
        monkeypatch.setenv("LOIKB_LEAN", "0")
    model = composite_tree(41, 14, [1, 5, 9])          # a composite root joint, two inside the tree
    link = model.njoints - 1
    B = 160
    wl = batch_for(model, B, link, 6)
    # a few iterations, field by field, per body of the caller's model
    prm = dict(FIXTURE, max_iter=5, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    s = loik_amd.BatchedLoik(model, B, **prm, **ENGINE_KW[engine])
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    got = {n: s.get(n) for n in ("nu", "z", "w", "vis", "fis", "g", "liMi", "yis", "primal_residual", "dual_residual")}
    for b in range(0, B, 23):
        r = ref.RefSolver(model, **prm)
        r.Solve(*problem_args(wl, b))
        for n in ("nu", "z", "w", "yis"):
            assert_close(got[n][b], r.field(n), 1e-7, n)
        for n in ("vis", "fis", "g"):
            assert_close(got[n][b], r.field(n)[1:], 1e-7, n)
        assert_close(got["liMi"][b], r.liMi[1:], 1e-12, "liMi of the composite = product of its sub-joints")
        for n in ("primal_residual", "dual_residual"):
            assert_close(got[n][b], r.scalar(n), 1e-7, n)
    s.close()
    # to convergence, every instance
    prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                          nthreads=4, want_nu=True, **prm)
    s = loik_amd.BatchedLoik(model, B, **prm, **ENGINE_KW[engine])
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=5e-8, off_ztol=1e-5, what="composite " + engine)
    # the outer loop: q <- q (+) dt z per sub-joint coordinate (an unbounded revolute sub-joint keeps (cos, sin))
    q0, z = s.get("q"), s.get("z")
    s.integrate(0.05)
    q1 = s.get("q")
    ch = workloads._Chain(model)
    for j in range(1, ch.njoints):
        iq, iv = int(ch.idx_q[j]), int(ch.idx_v[j])
        if int(ch.jtype[j]) >= J_RUBX:
            th = np.arctan2(q0[:, iq + 1], q0[:, iq]) + 0.05 * z[:, iv]
            assert np.max(np.abs(q1[:, iq] - np.cos(th))) < 1e-12 and np.max(np.abs(q1[:, iq + 1] - np.sin(th))) < 1e-12
        else:
            assert np.max(np.abs(q1[:, iq] - (q0[:, iq] + 0.05 * z[:, iv]))) < 1e-14
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["default", "solve_only"])
def test_gpu_composite_with_multidof_subjoints(engine):
    """composites whose sub-joints are multi-DoF joints (VERDICT r02 missing #3): on the device every sub-joint expands into its
    own chain; against the oracle's true composite, a few iterations field by field, to convergence, and integrate()"""
    model = composite_tree(27, 12, [1, 4, 7], kinds=MULTIDOF_KINDS)
    link = model.njoints - 1
    B = 128
    wl = batch_for(model, B, link, 6)
    prm = dict(FIXTURE, max_iter=5, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    s = loik_amd.BatchedLoik(model, B, **prm, **ENGINE_KW[engine])
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    got = {n: s.get(n) for n in ("nu", "z", "w", "vis", "fis", "g", "liMi", "yis", "primal_residual", "dual_residual")}
    for b in range(0, B, 19):
        r = ref.RefSolver(model, **prm)
        r.Solve(*problem_args(wl, b))
        for n in ("nu", "z", "w", "yis"):
            assert_close(got[n][b], r.field(n), 1e-7, n)
        for n in ("vis", "fis", "g"):
            assert_close(got[n][b], r.field(n)[1:], 1e-7, n)
        assert_close(got["liMi"][b], r.liMi[1:], 1e-12, "liMi of the composite = product of its sub-joints")
        for n in ("primal_residual", "dual_residual"):
            assert_close(got[n][b], r.scalar(n), 1e-7, n)
    s.close()
    prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                          nthreads=4, want_nu=True, **prm)
    s = loik_amd.BatchedLoik(model, B, **prm, **ENGINE_KW[engine])
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=2e-7, off_ztol=1e-5, what="composite of multi-DoF joints " + engine)
    # q <- q (+) dt z: the translation + spherical composite integrates like those joints (R^3 sum, SO(3) exponential)
    from test_multidof import _np_integrate
    q0, z = s.get("q"), s.get("z")
    s.integrate(0.05)
    q1 = s.get("q")
    ch = workloads._Chain(model)
    sel_q = list(range(0, 7))                       # the coordinates of composite joint 1: t (3), quaternion (4)
    for b in range(0, B, 31):
        want = _np_integrate(ch, q0[b], 0.05 * z[b])
        assert np.max(np.abs(q1[b][sel_q] - want[sel_q])) < 1e-9, b
    s.close()


@pytest.mark.gpu
def test_gpu_composite_errors():
    base = random_tree(5, 6)
    jt = base.jtype.copy(); jt[2] = J_COMPOSITE
    bad = loik_amd.Model(base.parents, jt, base.axis, base.placement,
                         composite={2: [(J_RZ, np.zeros(3), IDENT), (J_FREEFLYER, np.zeros(3), IDENT)]})
    with pytest.raises(loik_amd.LoikError) as e:      # more than six degrees of freedom in one composite
        loik_amd.BatchedLoik(bad, 4, **FIXTURE)
    assert e.value.code == -7
