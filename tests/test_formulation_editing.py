"""IkProblemFormulationOptimized's editing methods: per-link `UpdateReferences`
(/root/reference/include/loik/ik-id-description-optimized.hpp:103-121), `UpdateEqConstraint` (:178-238),
`AddEqConstraint` / `RemoveEqConstraint` (:244-319, "deactivated for now" upstream).

CPU part: the recursive oracle against the dense restatement and against scipy's SLSQP on the per-link cost; the edit
functions against a fresh solver built with the edited constraint set.  GPU part: the C-ABI's loikb_update_references /
loikb_add_eq_constraint / loikb_remove_eq_constraint / loikb_update_eq_constraint against the oracle, every engine that
takes them."""
import numpy as np
import pytest
from scipy.optimize import minimize

import loik_amd
from helpers import FIXTURE, assert_close, feasible_batch, multi_task_batch, problem_args, random_tree, random_tree_multidof
from oracle import dense, ref


def per_link_references(model, seed, kind="full"):
    """one weight and one target per link (index 0 = the universe: carried, never read)"""
    rng = np.random.default_rng(seed)
    nj = model.njoints
    H = np.zeros((nj, 6, 6)); v = 0.3 * rng.normal(size=(nj, 6))
    for i in range(nj):
        if kind == "diag":
            H[i] = np.diag(rng.uniform(0.2, 3.0, size=6))
        else:
            M = rng.normal(size=(6, 6))
            H[i] = M @ M.T / 6 + rng.uniform(0.1, 1.0) * np.eye(6)
    if kind == "some_zero":   # a link without a cost of its own
        H[2] = 0.0; v[2] = 0.0
    return H, v


def jacobian(model, q, link):
    from loik_amd import workloads
    J = np.zeros((6, model.nv))
    for k in range(model.nv):
        e = np.zeros((1, model.nv)); e[0, k] = 1.0
        J[:, k] = workloads.link_velocity(model, q[None], e, link)[0]
    return J


@pytest.mark.parametrize("kind", ["full", "diag", "some_zero"])
def test_oracles_agree_on_per_link_references(kind):
    model = random_tree(31, 9)
    wl = feasible_batch(model, 2, model.njoints - 1, 8)
    H, v = per_link_references(model, 5, kind)
    prm = dict(FIXTURE, max_iter=12, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    for b in range(2):
        opt, pl = ref.RefSolver(model, **prm), dense.DenseSolver(model, **prm)
        opt.SolveInit(*problem_args(wl, b)); opt.UpdateReferences(H, v); opt.Solve()
        pl.SolveInit(*problem_args(wl, b)); pl.UpdateReferences(H, v); pl.Solve()
        for n in ("nu", "z", "w"):
            assert_close(getattr(opt, n), getattr(pl, n), 1e-9, n)
        assert_close(opt.vis[1:], pl.vis[1:], 1e-9, "vis")
        assert_close(opt.fis[1:], pl.fis[1:], 1e-8, "fis")
        assert_close(opt.scalar("dual_residual"), pl.dual_residual, 1e-8, "dual")
        assert_close(opt.scalar("primal_residual"), pl.primal_residual, 1e-9, "primal")
        assert_close(opt.dual_residual_vec, pl.dual_residual_vec, 1e-8, "dual residual vector")


def test_per_link_references_solve_the_weighted_qp():
    """min_nu sum_i 1/2 |J_i nu - v_ref_i|^2_{H_i}  s.t.  A J_c nu = b, lb <= nu <= ub, by an unrelated method"""
    model = random_tree(4, 9)
    link = model.njoints - 1
    wl = feasible_batch(model, 2, link, 17, bound=0.5, nu_scale=0.5)
    H, v = per_link_references(model, 11)
    for b in range(2):
        s = ref.RefSolver(model, **dict(FIXTURE, max_iter=4000, tol_abs=1e-9, tol_rel=0.0, tol_primal_inf=1e-12))
        s.SolveInit(*problem_args(wl, b)); s.UpdateReferences(H, v); s.Solve()
        assert s.get_convergence_status(), s.get_iter()
        Js = [jacobian(model, wl["q"][b], i) for i in range(1, model.njoints)]
        A, bb = wl["Ais"][0], wl["bis"][b, 0]
        Jc = A @ Js[link - 1]

        def cost(x):
            return sum(0.5 * (J @ x - v[i + 1]) @ H[i + 1] @ (J @ x - v[i + 1]) for i, J in enumerate(Js))

        def grad(x):
            return sum(J.T @ H[i + 1] @ (J @ x - v[i + 1]) for i, J in enumerate(Js))
        res = minimize(cost, np.zeros(model.nv), jac=grad, method="SLSQP", bounds=list(zip(wl["lb"], wl["ub"])),
                       constraints=[dict(type="eq", fun=lambda x: Jc @ x - bb, jac=lambda x: Jc)],
                       options=dict(ftol=1e-15, maxiter=500))
        assert res.success
        assert abs(cost(s.z) - res.fun) < 1e-7 * max(1.0, abs(res.fun))
        assert np.max(np.abs(s.z - res.x)) < 2e-4


def test_update_references_quirks():
    """wrong size -> error (hpp:105-107); Hv_inf_norm_ only grows in UpdateReferences (hpp:114-116) and is the universe
    entry's norm after UpdateReference (hpp:94)"""
    model = random_tree(3, 6)
    wl = feasible_batch(model, 1, model.njoints - 1, 2)
    s = ref.RefSolver(model, **FIXTURE)
    wl["H_ref"], wl["v_ref"] = 2.0 * np.eye(6), np.full(6, 0.25)
    s.SolveInit(*problem_args(wl, 0))
    assert s.scalar("Hv_inf_norm") == 0.5
    H, v = per_link_references(model, 1, "diag")
    with pytest.raises(RuntimeError):
        s.UpdateReferences(H[1:], v[1:])
    s.UpdateReferences(0.01 * H, v)          # all |H_i v_i| < 0.5: the norm stays
    assert s.scalar("Hv_inf_norm") == 0.5
    s.UpdateReferences(100.0 * H, v)
    want = max(np.max(np.abs(100.0 * H[i] @ v[i])) for i in range(model.njoints))
    assert_close(s.scalar("Hv_inf_norm"), want, 1e-12, "Hv_inf_norm")


def _constraint_set(model, links, seed, B=1):
    wl = multi_task_batch(model, B, links, seed, bound=0.5, nu_scale=0.4)
    return wl


def test_add_and_remove_equal_a_fresh_solver():
    """cold start: a solver whose set was edited to {a, c} answers like one constructed with {a, c}"""
    model = random_tree(8, 12, branch_prob=0.4)
    a, b_, c = 5, 9, 12
    full = _constraint_set(model, [a, b_, c], 3)
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-8, tol_rel=0.0)
    A = {l: full["Ais"][k] for k, l in enumerate((a, b_, c))}
    bv = {l: full["bis"][0, k] for k, l in enumerate((a, b_, c))}
    q, lb, ub = full["q"][0], full["lb"], full["ub"]

    def fresh(links):
        s = ref.RefSolver(model, **dict(prm, num_eq_c=len(links)))
        s.Solve(q, full["H_ref"], full["v_ref"], np.array(links, dtype=np.int32), np.array([A[l] for l in links]).reshape(-1, 6, 6),
                np.array([bv[l] for l in links]).reshape(-1, 6), lb, ub)
        return s

    s = ref.RefSolver(model, **dict(prm, num_eq_c=2, eq_c_capacity=3))
    s.Solve(q, full["H_ref"], full["v_ref"], np.array([a, b_], dtype=np.int32), np.array([A[a], A[b_]]), np.array([bv[a], bv[b_]]), lb, ub)
    want = fresh([a, b_])
    assert s.get_iter() == want.get_iter() and np.array_equal(s.z, want.z)
    # add c (third slot), solve without a constraint update
    s.AddEqConstraint(c, A[c], bv[c])
    assert s.active_task_constraint_ids() == [a, b_, c]
    s.Solve(q, -1, None, None)
    want = fresh([a, b_, c])
    assert s.get_iter() == want.get_iter() and np.array_equal(s.z, want.z) and np.array_equal(s.yis, want.yis)
    # a fourth one does not fit
    with pytest.raises(RuntimeError):
        s.AddEqConstraint(3, A[c], bv[c])
    # remove the middle one: c moves down
    assert s.RemoveEqConstraint(b_)
    assert s.active_task_constraint_ids() == [a, c]
    assert not s.RemoveEqConstraint(b_)          # nothing to remove: upstream warns and returns
    s.Solve(q, -1, None, None)
    want = fresh([a, c])
    assert s.get_iter() == want.get_iter() and np.array_equal(s.z, want.z) and np.array_equal(s.yis, want.yis)
    assert s.scalar("bis_inf_norm") == want.scalar("bis_inf_norm")
    # AddEqConstraint on a link that has one = UpdateEqConstraint (hpp:250-253); the (c_id, bi) overload keeps A (hpp:224-238)
    s.AddEqConstraint(c, A[b_], bv[b_])
    s.UpdateEqConstraint(a, bv[c])
    s.Solve(q, -1, None, None)
    w2 = ref.RefSolver(model, **dict(prm, num_eq_c=2))
    w2.Solve(q, full["H_ref"], full["v_ref"], np.array([a, c], dtype=np.int32), np.array([A[a], A[b_]]), np.array([bv[c], bv[b_]]), lb, ub)
    assert s.get_iter() == w2.get_iter() and np.array_equal(s.z, w2.z)
    # removing everything leaves the box-constrained reference tracking problem
    s.RemoveEqConstraint(a); s.RemoveEqConstraint(c)
    s.Solve(q, -1, None, None)
    w0 = ref.RefSolver(model, **dict(prm, num_eq_c=0))
    w0.Solve(q, full["H_ref"], full["v_ref"], np.zeros(0, dtype=np.int32), np.zeros((0, 6, 6)), np.zeros((0, 6)), lb, ub)
    assert s.get_iter() == w0.get_iter() and np.array_equal(s.z, w0.z)
