"""Helical joints (VERDICT r02 "missing" #3): JointModelHelicalX / Y / Z / Unaligned -- a rotation by q about the axis together
with a translation of pitch * q along it, S = [pitch a; a].  The reference gets them from Pinocchio's joint variant
(/root/reference/include/loik/loik-loid-optimized.hxx:21-23, :91-93: `jdata.S()`, `calc_aba`).  Here they are 1-DoF joints of
every engine: the device stores (q, 0) for them (the translation needs the angle itself) and adds the linear term pitch * a
next to the revolute joint's angular one wherever S appears.

CPU: the oracle's helical joint against first principles (finite differences of the placement; the QP optimum over the bodies).
GPU: every engine against the oracle, integrate()."""
import numpy as np
import pytest
from scipy.optimize import minimize

import loik_amd
from helpers import FIXTURE, assert_close, assert_end_to_end, fetch_end_to_end, helical_tree, problem_args
from loik_amd import workloads
from oracle import ref


def _placements(model, q):
    """world placements of every joint frame, from scratch (Rodrigues + translation along the axis)"""
    out = [(np.eye(3), np.zeros(3))]
    for i in range(1, model.njoints):
        jt, a = int(model.jtype[i]), np.asarray(model.axis[i], dtype=float)
        P = model.placement[i]; Rp, tp = P[:9].reshape(3, 3), P[9:]
        qi = q[int(model.idx_q[i])]
        if jt in (1, 2, 3, 19, 20, 21):
            a = np.eye(3)[(jt - 1) % 3 if jt < 4 else jt - 19]
        elif jt in (4, 5, 6):
            a = np.eye(3)[jt - 4]
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        rot = jt in (1, 2, 3, 7, 19, 20, 21, 22)
        Rj = np.eye(3) + np.sin(qi) * K + (1 - np.cos(qi)) * K @ K if rot else np.eye(3)
        tj = (model.pitch[i] * qi * a) if jt >= 19 else (np.zeros(3) if rot else qi * a)
        Rw, tw = out[int(model.parents[i])]
        out.append((Rw @ Rp @ Rj, tw + Rw @ (tp + Rp @ tj)))
    return out


def test_helical_joint_kinematics_from_first_principles():
    """liMi of the oracle = placement * (Rot(a, q), pitch q a); the link velocities J nu of workloads.link_velocity (S = [pitch a; a])
    = the finite-difference velocity of the frames under q -> q + eps nu"""
    model = helical_tree(7, 8, 3)
    assert np.count_nonzero(model.pitch) == 3 and model.nq == model.nv == 8
    rng = np.random.default_rng(0)
    q = rng.uniform(-1, 1, size=model.nq); nu = rng.normal(size=model.nv)
    wl = workloads.make_workload(model, 1, model.njoints - 1, 3, bound=0.5, snap_prob=0.0, nu_scale=0.4)
    r = ref.RefSolver(model, **dict(FIXTURE, max_iter=3))
    r.Solve(q, wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"][0], wl["lb"], wl["ub"])
    W = _placements(model, q)
    for i in range(1, model.njoints):
        Rw, tw = W[i]; Rpw, tpw = W[int(model.parents[i])]
        assert_close(r.liMi[i][:9].reshape(3, 3), Rpw.T @ Rw, 1e-13, "liMi rotation")
        assert_close(r.liMi[i][9:], Rpw.T @ (tw - tpw), 1e-13, "liMi translation (pitch q a for a helical joint)")
    eps = 1e-6
    Wp, Wm = _placements(model, q + eps * nu), _placements(model, q - eps * nu)
    for i in range(1, model.njoints):
        v = workloads.link_velocity(model, q[None], nu[None], i)[0]
        Rw, tw = W[i]
        lin = Rw.T @ ((Wp[i][1] - Wm[i][1]) / (2 * eps))
        Om = Rw.T @ ((Wp[i][0] - Wm[i][0]) / (2 * eps))     # R^T Rdot = [omega]x
        ang = np.array([Om[2, 1], Om[0, 2], Om[1, 0]])
        assert_close(v, np.concatenate([lin, ang]), 1e-8, "link velocity of joint %d" % i)


def test_oracle_helical_solves_the_qp_over_the_bodies():
    model = helical_tree(11, 9, 4, branch_prob=0.12)
    depth = [0] * model.njoints
    for i in range(1, model.njoints):
        depth[i] = depth[int(model.parents[i])] + 1
    link = int(np.argmax(depth))                       # (a chain of >= 6 joints under the task: a 6-D target is reachable)
    assert depth[link] >= 6 and np.count_nonzero(model.pitch) == 4
    wl = workloads.make_workload(model, 2, link, 17, bound=0.5, snap_prob=0.0, nu_scale=0.4)

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
            assert np.max(np.abs(Js[i - 1] @ s.nu - s.vis[i])) < 1e-10       # the solver's link velocities ARE J nu
        Hq = sum(J.T @ J for J in Js)
        Jc, bb = wl["Ais"][0] @ Js[link - 1], wl["bis"][b, 0]
        res = minimize(lambda x: 0.5 * x @ Hq @ x, np.zeros(model.nv), jac=lambda x: Hq @ x, method="SLSQP",
                       bounds=list(zip(wl["lb"], wl["ub"])), constraints=[dict(type="eq", fun=lambda x: Jc @ x - bb, jac=lambda x: Jc)],
                       options=dict(ftol=1e-15, maxiter=500))
        assert res.success
        assert abs(0.5 * s.z @ Hq @ s.z - res.fun) < 1e-7 * max(1.0, abs(res.fun)) and np.max(np.abs(s.z - res.x)) < 2e-4


ENGINE_KW = {"default": ({}, {}), "solve_only": ({}, dict(tail_max_instances=-1)), "tail_only": ({"LOIKB_LEAN": "0"}, dict(tail_max_instances=1 << 20)),
             "handover": ({}, dict(max_launch_iters=3, tail_max_instances=1 << 20)), "lean": ({"LOIKB_FLAT": "0"}, {}),
             "flat_one_lane": ({"LOIKB_FLAT_SPLIT": "0"}, {})}


@pytest.mark.gpu
@pytest.mark.parametrize("nb", [9, 22, 40])
@pytest.mark.parametrize("engine", sorted(ENGINE_KW))
def test_gpu_helical_joints(engine, nb, monkeypatch):
    """9 joints: k_solve / k_tail; 22: k_flat2 (k_flat with LOIKB_FLAT_SPLIT=0), k_lean; 40: k_lean / k_solve + k_tail"""
    env, kw = ENGINE_KW[engine]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    model = helical_tree(100 + nb, nb, 4, branch_prob=0.5 if nb > 16 else 0.35)
    link = model.njoints - 1
    B = 150
    wl = workloads.make_workload(model, B, link, 6, bound=0.5, snap_prob=0.0, nu_scale=0.4)
    prm = dict(FIXTURE, max_iter=5, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    s = loik_amd.BatchedLoik(model, B, **prm, **kw)
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    got = {n: s.get(n) for n in ("nu", "z", "w", "vis", "fis", "g", "liMi", "yis", "Stf_plus_w", "primal_residual", "dual_residual")}
    his = s.His_full()
    for b in range(0, B, 29):
        r = ref.RefSolver(model, **prm)
        r.Solve(*problem_args(wl, b))
        assert_close(got["liMi"][b], r.liMi[1:], 1e-12, "liMi")
        for n in ("nu", "z", "w", "yis", "Stf_plus_w"):
            assert_close(got[n][b], r.field(n), 1e-9, n)
        for n in ("vis", "fis", "g"):
            assert_close(got[n][b], r.field(n)[1:], 1e-9, n)
        assert_close(his[b], r.His[1:], 1e-8, "His")
        for n in ("primal_residual", "dual_residual"):
            assert_close(got[n][b], r.scalar(n), 1e-9, n)
    s.close()
    prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                          nthreads=4, want_nu=True, **prm)
    s = loik_amd.BatchedLoik(model, B, **prm, **kw)
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    st = s.stats()
    if engine == "default" and nb == 22:
        assert st["flat_split_launches"] >= 1, (st, s.plan())     # (k_flat2; the 40-joint tree is too deep for the flat engine: k_lean)
    assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=1e-9, what="helical %s nb %d" % (engine, nb))
    q0, z = s.get("q"), s.get("z")
    s.integrate(0.05)
    assert np.max(np.abs(s.get("q") - (q0 + 0.05 * z))) < 1e-14      # (a helical joint's configuration is its angle: R^1)
    s.close()


HELICAL_SUBS = [[20, 4], [22, 10], [2, 19, 6]]   # (HY, PX) | (HU, spherical) | (RY, HX, PZ)


def test_oracle_composite_with_helical_subjoints_equals_its_chain():
    """a helical joint as a sub-joint of a JointModelComposite (comp_pitch): the oracle's composite (its S columns seen from the last
    frame) against the same model written as the chain of its sub-joints with massless links"""
    from helpers import composite_tree
    from test_composite import chain_of
    model = composite_tree(61, 9, [1, 4, 7], kinds=HELICAL_SUBS)
    assert np.count_nonzero(model.comp_pitch) == 3
    m1, link_of = chain_of(model)
    m1.pitch = workloads._Chain(model).pitch
    link = model.njoints - 1
    wl = workloads.make_workload(model, 3, link, 8, bound=0.5, snap_prob=0.0, nu_scale=0.4)
    prm = dict(FIXTURE, max_iter=60, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    for b in range(3):
        t, c = ref.RefSolver(model, **prm), ref.RefSolver(m1, **prm)
        t.Solve(*problem_args(wl, b))
        c.Solve(wl["q"][b], wl["H_ref"], wl["v_ref"], np.array([link_of[link]], dtype=np.int32), wl["Ais"], wl["bis"][b], wl["lb"], wl["ub"])
        for n in ("nu", "z", "w"):
            assert_close(getattr(c, n), getattr(t, n), 1e-8, n)
        assert_close(c.vis[link_of[1:]], t.vis[1:], 1e-8, "vis of the bodies")
        assert_close(c.fis[link_of[1:]], t.fis[1:], 1e-7, "fis of the bodies")


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["default", "solve_only"])
def test_gpu_composite_with_helical_subjoints(engine):
    from helpers import composite_tree
    model = composite_tree(61, 12, [1, 4, 7], kinds=HELICAL_SUBS)
    link = model.njoints - 1
    B = 96
    wl = workloads.make_workload(model, B, link, 6, bound=0.5, snap_prob=0.0, nu_scale=0.4)
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                          nthreads=4, want_nu=True, **prm)
    s = loik_amd.BatchedLoik(model, B, **prm, **ENGINE_KW[engine][1])
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=1e-8, off_ztol=1e-5, what="composite with helical sub-joints " + engine)
    for b in range(0, B, 31):
        r = ref.RefSolver(model, **prm)
        r.Solve(*problem_args(wl, b))
        assert_close(s.get("liMi")[b], r.liMi[1:], 1e-12, "liMi of the composite")
    s.close()


@pytest.mark.gpu
def test_gpu_helical_pass_level_and_errors():
    model = helical_tree(5, 10, 3)
    B = 40
    wl = workloads.make_workload(model, B, model.njoints - 1, 2, bound=0.5, snap_prob=0.0, nu_scale=0.4)
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0)
    s = loik_amd.BatchedLoik(model, B, logging=True, **prm)      # B < 64: the plain pass-by-pass implementation
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    for b in range(0, B, 13):
        r = ref.RefSolver(model, **prm)
        r.Solve(*problem_args(wl, b))
        assert s.get("iter")[b] == r.get_iter()
        assert_close(s.get("z")[b], r.z, 1e-9, "z (pass-level path)")
    s.close()
    bad = loik_amd.Model(model.parents, model.jtype, model.axis, model.placement)   # helical joints without the pitch array
    with pytest.raises(loik_amd.LoikError) as e:
        loik_amd.BatchedLoik(bad, 4, **FIXTURE)
    assert e.value.code == -7
