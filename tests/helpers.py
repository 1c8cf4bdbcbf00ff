"""shared test helpers: the reference's comparison predicates, its fixture problem, random kinematic trees"""
import os
import numpy as np

import loik_amd
from loik_amd import workloads


def scalar_abs_or_rel_equal(a, b, tol=1e-10):
    """check_scalar_abs_or_rel_equal, /root/reference/tests/loik-loid.cpp:39-57"""
    a, b = float(a), float(b)
    d = abs(a - b)
    c_abs = d < tol
    c_rel = (abs(a) > 0 and abs(b) > 0 and d / abs(a) < tol and d / abs(b) < tol)
    return c_abs or c_rel


def dense_abs_or_rel_equal(a, b, tol=1e-10):
    """check_eigen_dense_abs_or_rel_equal, /root/reference/tests/loik-loid.cpp:60-83
    (Eigen isApprox: ||a-b||_F <= 1e-12 * min(||a||_F, ||b||_F), OR inf-norm of the difference < tol)"""
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    c1 = np.linalg.norm(a - b) <= 1e-12 * min(np.linalg.norm(a), np.linalg.norm(b))
    c2 = np.max(np.abs(a - b)) < tol if a.size else True
    return bool(c1 or c2)


def assert_close(a, b, tol=1e-10, what=""):
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    d = np.abs(a - b)
    den = np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-300)
    bad = np.minimum(d, d / den) >= tol
    assert not bad.any(), "%s: abs-or-rel mismatch %.3e (tol %.1e)" % (what, float(np.minimum(d, d / den).max()), tol)


def fixture_problem(model, bound=4.0, link=None):
    """ProblemSetupFixture, /root/reference/tests/loik-loid.cpp:87-165: q = neutral, H_ref = I, v_ref = 0, one
    constraint on the LAST joint with A = I, b = (0,0,.5,0,0,0), box = +-bound"""
    link = model.njoints - 1 if link is None else link
    return dict(q=np.zeros(model.nq), H_ref=np.eye(6), v_ref=np.zeros(6), c_ids=np.array([link], dtype=np.int32),
                Ais=np.eye(6).reshape(1, 6, 6), bis=np.array([[0, 0, 0.5, 0, 0, 0.0]]),
                lb=-bound * np.ones(model.nv), ub=bound * np.ones(model.nv))


def problem_args(p, b=None):
    """(q,H_ref,v_ref,ids,Ais,bis,lb,ub) of one instance `b` of a batched workload, or of a single problem"""
    if b is None:
        return (p["q"], p["H_ref"], p["v_ref"], p["c_ids"], p["Ais"], p["bis"], p["lb"], p["ub"])
    Ais = p["Ais"] if np.asarray(p["Ais"]).ndim == 3 else p["Ais"][b]
    lb = p["lb"] if np.asarray(p["lb"]).ndim == 1 else p["lb"][b]
    ub = p["ub"] if np.asarray(p["ub"]).ndim == 1 else p["ub"][b]
    return (p["q"][b], p["H_ref"], p["v_ref"], p["c_ids"], Ais, p["bis"][b], lb, ub)


FIXTURE = dict(tol_abs=1e-3, tol_rel=1e-3, tol_primal_inf=1e-2, tol_dual_inf=1e-2, tol_tail_solve=1e-1, rho=1e-5,
               mu=1e-2, mu_equality_scale_factor=1e4, mu_update_strat=0, num_eq_c=1, eq_c_dim=6, warm_start=False)


def random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def random_tree(seed, nb, branch_prob=0.35, all_types=True):
    """random depth-first-numbered kinematic tree (like pinocchio::buildModels::humanoidRandom in spirit,
    /root/reference/tests/loik-loid-data.cpp:24-31): mixed revolute / prismatic / unaligned joints, random
    placements, several branch points and several root children -> exercises the LDS branch stack."""
    rng = np.random.default_rng(seed)
    parents = [0]
    # build by DFS: keep a stack of "open" ancestors; next joint attaches to the top or pops
    path = [0]
    for i in range(1, nb + 1):
        while len(path) > 1 and rng.random() < branch_prob:
            path.pop()
        parents.append(path[-1])
        path.append(i)
    types = [0]
    axis = [np.zeros(3)]
    placement = [np.concatenate([np.eye(3).ravel(), np.zeros(3)])]
    for i in range(1, nb + 1):
        t = int(rng.integers(1, 9)) if all_types else int(rng.integers(1, 4))
        a = np.zeros(3)
        if t in (7, 8):
            a = rng.normal(size=3)
            a /= np.linalg.norm(a)
        else:
            a[(t - 1) % 3] = 1.0
        types.append(t)
        axis.append(a)
        placement.append(np.concatenate([random_rotation(rng).ravel(), rng.uniform(-0.4, 0.4, size=3)]))
    return loik_amd.Model(parents, types, np.array(axis), np.array(placement), q_lo=-np.ones(nb), q_hi=np.ones(nb),
                          name="random_tree_%d_%d" % (seed, nb))


def helical_tree(seed, nb, n_helical, branch_prob=0.35):
    """random_tree(seed, nb) with `n_helical` of its revolute joints turned into helical ones (JointModelHelicalX / Y / Z /
    Unaligned: the same axis + a pitch in +-[0.05, 0.3] m per radian)"""
    m = random_tree(seed, nb, branch_prob=branch_prob)
    rng = np.random.default_rng(seed + 4321)
    jt = m.jtype.copy()
    pitch = np.zeros(m.njoints)
    rev = [i for i in range(1, m.njoints) if int(jt[i]) in (1, 2, 3, 7)]
    for i in rng.choice(rev, size=min(n_helical, len(rev)), replace=False):
        jt[i] = 22 if int(jt[i]) == 7 else 19 + (int(jt[i]) - 1)
        pitch[i] = float(rng.choice([-1.0, 1.0]) * rng.uniform(0.05, 0.3))
    return loik_amd.Model(m.parents, jt, m.axis, m.placement, q_lo=m.q_lo, q_hi=m.q_hi, pitch=pitch, name="helical_tree_%d_%d" % (seed, nb))


def feasible_batch(model, batch, link, seed, bound=0.5, nu_scale=0.4, per_instance_A=False, per_instance_bounds=False):
    wl = workloads.make_workload(model, batch, link, seed, bound=bound, snap_prob=0.0, nu_scale=nu_scale)
    rng = np.random.default_rng(seed + 1)
    if per_instance_A:
        A = np.eye(6)[None, None] + 0.3 * rng.normal(size=(batch, 1, 6, 6))
        v = workloads.link_velocity(model, wl["q"], wl["nu_star"], link)
        wl["Ais"] = A
        wl["bis"] = np.einsum("bcij,bj->bci", A, v)
    if per_instance_bounds:
        wl["lb"] = -bound * (1 + 0.2 * rng.random((batch, model.nv)))
        wl["ub"] = bound * (1 + 0.2 * rng.random((batch, model.nv)))
    return wl


# ---- multi-DoF joints (free-flyer / spherical / translation) ------------------------------------------------------
J_FREEFLYER, J_SPHERICAL, J_TRANSLATION = 9, 10, 11
# virtual 1-DoF joint types of the chain that stands for a multi-DoF joint (S = the columns of I6 it selects)
CHAIN_TYPES = {J_FREEFLYER: [4, 5, 6, 1, 2, 3], J_SPHERICAL: [1, 2, 3], J_TRANSLATION: [4, 5, 6]}


J_SPHERICAL_ZYX, J_PLANAR, J_RUBX, J_RUBY, J_RUBZ = 12, 13, 14, 15, 16
J_RUBU = 18   # JointModelRevoluteUnboundedUnaligned


def random_tree_multidof(seed, nb, root_freeflyer=True, n_spherical=1, n_translation=1, branch_prob=0.3, n_zyx=0, n_planar=0,
                         n_rub=0, root_planar=False, n_rubu=0):
    """random_tree() with some joints replaced by multi-DoF ones (optionally a free-flyer root joint: the
    floating-base case of SURVEY.md 8(f) rank 2)"""
    m = random_tree(seed, nb, branch_prob=branch_prob)
    rng = np.random.default_rng(seed + 77)
    jt = m.jtype.copy()
    if root_freeflyer:
        jt[1] = J_FREEFLYER
    cand = [i for i in range(2 if (root_freeflyer or root_planar) else 1, nb + 1)]
    rng.shuffle(cand)
    for i in cand[:n_spherical]:
        jt[i] = J_SPHERICAL
    for i in cand[n_spherical:n_spherical + n_translation]:
        jt[i] = J_TRANSLATION
    k = n_spherical + n_translation
    for i in cand[k:k + n_zyx]:
        jt[i] = J_SPHERICAL_ZYX
    for i in cand[k + n_zyx:k + n_zyx + n_planar]:
        jt[i] = J_PLANAR
    for n_, i in enumerate(cand[k + n_zyx + n_planar:k + n_zyx + n_planar + n_rub]):
        jt[i] = J_RUBX + n_ % 3
    axis = np.array(m.axis, dtype=float).copy()
    for i in cand[k + n_zyx + n_planar + n_rub:k + n_zyx + n_planar + n_rub + n_rubu]:
        jt[i] = J_RUBU
        a = rng.normal(size=3)
        axis[i] = a / np.linalg.norm(a)   # an axis aligned with nothing
    if root_planar:
        jt[1] = J_PLANAR
    return loik_amd.Model(m.parents, jt, axis, m.placement, name="random_multidof_%d_%d" % (seed, nb))


def expand_to_chains(model, q):
    """The all-1-DoF model the device solves instead of `model` (single configuration q): every multi-DoF joint becomes
    a chain of 1-DoF joints about the axes of ONE frame (identity placements in between) whose intermediate links are
    massless; M(q) of the joint is folded into the placement of the first chain joint.  Returns (model1, q1, link_of)
    with link_of[i] = the chain link that carries body i."""
    from loik_amd import workloads as W
    parents, types, axis, placement, massless, link_of = [0], [0], [np.zeros(3)], [model.placement[0]], [0], [0]
    for i in range(1, model.njoints):
        t = int(model.jtype[i])
        par = link_of[int(model.parents[i])]
        if t not in CHAIN_TYPES:
            parents.append(par); types.append(t); axis.append(model.axis[i]); placement.append(model.placement[i])
            massless.append(0)
            link_of.append(len(parents) - 1)
            continue
        iq = int(model.idx_q[i])
        P = model.placement[i]
        Rp, tp = P[:9].reshape(3, 3), P[9:]
        if t == J_FREEFLYER:
            Rj, tj = W.quat_rot(q[None, iq + 3:iq + 7])[0], q[iq:iq + 3]
        elif t == J_SPHERICAL:
            Rj, tj = W.quat_rot(q[None, iq:iq + 4])[0], np.zeros(3)
        else:
            Rj, tj = np.eye(3), q[iq:iq + 3]
        first = np.concatenate([(Rp @ Rj).ravel(), tp + Rp @ tj])
        ident = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
        chain = CHAIN_TYPES[t]
        for k, ct in enumerate(chain):
            parents.append(par if k == 0 else len(parents) - 1)
            types.append(ct)
            a = np.zeros(3); a[(ct - 1) % 3] = 1.0
            axis.append(a)
            placement.append(first if k == 0 else ident)
            massless.append(0 if k == len(chain) - 1 else 1)
        link_of.append(len(parents) - 1)
    m1 = loik_amd.Model(parents, types, np.array(axis), np.array(placement), name=model.name + "_chains")
    m1.massless = np.array(massless, dtype=np.int32)
    # configuration of the chain model: the 1-DoF joints keep their q, the chain joints sit at 0
    q1 = np.zeros(m1.nq)
    for i in range(1, model.njoints):
        if int(model.jtype[i]) not in CHAIN_TYPES:
            q1[int(m1.idx_q[link_of[i]])] = q[int(model.idx_q[i])]
    return m1, q1, np.array(link_of)


def multi_task_batch(model, batch, links, seed, bound=0.5, nu_scale=0.4, per_instance_A=False):
    """several simultaneous task constraints (num_eq_c = len(links) > 1, the ctor argument of the reference,
    loik-loid-optimized.hpp:129-134): b_c = A_c J_c(q) nu_star for one common nu_star -> jointly feasible"""
    wl = workloads.make_workload(model, batch, links[0], seed, bound=bound, snap_prob=0.0, nu_scale=nu_scale)
    rng = np.random.default_rng(seed + 5)
    nc = len(links)
    if per_instance_A:
        A = np.eye(6)[None, None] + 0.3 * rng.normal(size=(batch, nc, 6, 6))
    else:
        A = np.eye(6)[None] + 0.3 * rng.normal(size=(nc, 6, 6))
    b = np.empty((batch, nc, 6))
    for c, link in enumerate(links):
        v = workloads.link_velocity(model, wl["q"], wl["nu_star"], link)
        b[:, c] = np.einsum("bij,bj->bi", A[:, c], v) if per_instance_A else v @ A[c].T
    wl["c_ids"] = np.array(links, dtype=np.int32)
    wl["Ais"], wl["bis"] = A, b
    return wl


# what assert_end_to_end has seen in this test session: instances compared, instances off the oracle's iteration count, and
# off-count instances whose flags differed and were let through because the oracle's stopping comparison was within 1e-6
# (relative) of the tolerance or one side ran to max_iter (tests/test_zz_tally.py asserts the share)
TALLY = dict(compared=0, off_count=0, flags_exempted=0)


def assert_end_to_end(got, out, prm, same_frac=0.99, ztol=1e-9, off_ztol=1e-6, off_iter=None, what="", res_tol=(1e-9, 1e-6),
                      inf_ztol=None, off_scale=None):
    """End-to-end comparison of a batch with the oracle's `solve_batch` output -- every instance is checked, none dropped.

    got: dict with iter, converged, primal_infeasible, z (optionally nu, primal_residual, dual_residual) of the device;
    out: oracle.ref.solve_batch(...).  Instances with the oracle's iteration count (at least `same_frac` of the batch)
    must agree in every flag and to `ztol` in z / nu.  The rest -- a comparison of the stopping logic (residual < tol,
    primal > 10 dual, the certificate's <=) fell on the other side of a rounding error, so the instance stopped at a
    neighbouring iteration or took a different mu for a while -- is NOT skipped: both solvers must have stopped the
    same way (same flags, unless the oracle's own residual sits within rounding of the tolerance), within
    `off_iter` iterations of each other when given, and their answers must coincide to the solver tolerance (`off_ztol`).
    inf_ztol: budget in z for identical-iteration instances that BOTH solvers flagged primal infeasible (default: ztol) -- what such an
    instance returns is the iterate its tail solve stopped at, not a solution; off_scale: per-instance factor on off_ztol (the fuzz: 1 / mu
    where mu < 1 -- the dual residual bounds mu |z_k - z_k-1|, so neighbouring iterates of a converged instance are tol / mu apart).
    Returns the mask of identical-iteration instances."""
    it = np.asarray(got["iter"]); it_o = np.asarray(out["iters"])
    conv = np.asarray(got["converged"]).astype(bool); inf = np.asarray(got["primal_infeasible"]).astype(bool)
    same = it == it_o
    if os.environ.get("LOIKB_TEST_MARGINS"):   # (a record of how much of each call's tolerance is used: scripts/r06/test_margins.sh)
        dz_ = np.abs(np.asarray(got["z"]) - out["z"]).reshape(it.size, -1).max(axis=1)
        with open(os.environ["LOIKB_TEST_MARGINS"], "a") as f:
            f.write("%-60s n %6d  same %.4f (asked %.2f)  max|dz| same %.2e (ztol %.0e)  off %d max|dz| off %.2e (off_ztol %.0e)\n" % (
                what, it.size, same.mean(), same_frac, dz_[same].max() if same.any() else 0.0, ztol, int((~same).sum()),
                dz_[~same].max() if (~same).any() else 0.0, off_ztol))
    assert same.mean() >= same_frac, (what, "iteration counts differ", it[~same][:20], it_o[~same][:20])
    assert np.array_equal(conv[same], out["converged"][same]), what
    assert np.array_equal(inf[same], out["primal_infeasible"][same]), what
    dz = np.abs(np.asarray(got["z"]) - out["z"]).reshape(it.size, -1).max(axis=1)
    zt = np.full(it.size, float(ztol))
    if inf_ztol is not None:
        zt[inf & np.asarray(out["primal_infeasible"]).astype(bool)] = max(float(inf_ztol), float(ztol))
    assert np.all(dz[same] < zt[same]), (what, "z", dz[same].max())
    if "nu" in got and "nu" in out:
        dn = np.abs(np.asarray(got["nu"]) - out["nu"]).reshape(it.size, -1).max(axis=1)
        assert np.all(dn[same] < zt[same]), (what, "nu", dn[same].max())
    for name in ("primal_residual", "dual_residual"):
        if name in got and name in out:
            a, b = np.asarray(got[name])[same], out[name][same]
            assert np.all(np.abs(a - b) <= res_tol[0] + res_tol[1] * np.abs(b)), (what, name, float(np.abs(a - b).max()))
    off = np.flatnonzero(~same)
    tol = prm["tol_abs"]
    TALLY["compared"] += int(it.size)
    TALLY["off_count"] += int(off.size)
    for b in off:
        # how close the oracle's stopping comparison was: relative distance of its residuals from the tolerance
        near_tol = min(abs(out["primal_residual"][b] - tol), abs(out["dual_residual"][b] - tol)) <= 1e-6 * tol
        hit_max = it[b] >= prm["max_iter"] - 1 or it_o[b] >= prm["max_iter"] - 1
        assert dz[b] <= off_ztol * (1.0 if off_scale is None else float(off_scale[b])), (what, "instance %d: iterations %d vs %d, |dz| = %.3e" % (b, it[b], it_o[b], dz[b]))
        if not (near_tol or hit_max):
            assert conv[b] == out["converged"][b] and inf[b] == out["primal_infeasible"][b], (what, b, it[b], it_o[b])
        elif conv[b] != out["converged"][b] or inf[b] != out["primal_infeasible"][b]:
            TALLY["flags_exempted"] += 1  # (the exemption was actually used: tests/test_zz_tally.py bounds how often)
        if off_iter is not None and not hit_max:
            assert abs(int(it[b]) - int(it_o[b])) <= off_iter, (what, b, it[b], it_o[b])
    return same


def fetch_end_to_end(s, idx=None, nu=True, residuals=False):
    """the device side of assert_end_to_end (optionally a subset `idx` of the batch)"""
    sel = (lambda a: a) if idx is None else (lambda a: a[idx])
    got = dict(iter=sel(s.get("iter")), converged=sel(s.get("converged")), primal_infeasible=sel(s.get("primal_infeasible")),
               z=sel(s.get("z")))
    if nu:
        got["nu"] = sel(s.get("nu"))
    if residuals:
        got["primal_residual"] = sel(s.get("primal_residual"))
        got["dual_residual"] = sel(s.get("dual_residual"))
    return got


def renumber_breadth_first(model):
    """the same tree with its joints numbered level by level (parents[i] < i still holds, subtrees are no longer contiguous):
    what a model assembled with pinocchio's addJoint in arbitrary order may look like"""
    nj = model.njoints
    depth = [0] * nj
    for i in range(1, nj):
        depth[i] = depth[int(model.parents[i])] + 1
    order = sorted(range(nj), key=lambda i: (depth[i], i))      # new index -> old index
    new_of = {old: new for new, old in enumerate(order)}
    parents = [new_of[int(model.parents[o])] if o else 0 for o in order]
    m = loik_amd.Model(parents, model.jtype[order], model.axis[order], model.placement[order],
                       names=[model.names[o] for o in order], q_lo=None if model.q_lo is None else model.q_lo[[o - 1 for o in order[1:]]],
                       q_hi=None if model.q_hi is None else model.q_hi[[o - 1 for o in order[1:]]], name=model.name + "_bfs")
    return m, np.array(order)


J_COMPOSITE_ = 17


def _unit(rng):
    a = rng.normal(size=3)
    return a / np.linalg.norm(a)


def composite_tree(seed, nb, which, kinds=None):
    """random_tree(seed, nb) with the joints `which` replaced by composites of 2..4 random 1-DoF sub-joints (aligned,
    unaligned, unbounded revolute) with random internal placements"""
    m = random_tree(seed, nb, branch_prob=0.3)
    rng = np.random.default_rng(seed + 1234)
    jt = m.jtype.copy()
    comp = {}
    for n_, i in enumerate(which):
        jt[i] = J_COMPOSITE_
        subs = []
        types = kinds[n_] if kinds else [int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, J_RUBY, J_RUBU]))
                                         for _ in range(int(rng.integers(2, 5)))]
        for t in types:
            a = _unit(rng) if t in (7, 8, J_RUBU, 22) else np.zeros(3)
            P = np.concatenate([random_rotation(rng).ravel(), rng.uniform(-0.3, 0.3, size=3)])
            if t in (19, 20, 21, 22):   # helical sub-joint: (type, axis, placement, pitch)
                subs.append((t, a, P, float(rng.choice([-1.0, 1.0]) * rng.uniform(0.05, 0.3))))
            else:
                subs.append((t, a, P))
        comp[i] = subs
    return loik_amd.Model(m.parents, jt, m.axis, m.placement, composite=comp, name="composite_tree_%d_%d" % (seed, nb))
