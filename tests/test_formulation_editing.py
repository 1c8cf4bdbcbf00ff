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


# ---------------------------------------------------------------------------------------------------------------------------
# GPU: the C-ABI's editing entry points against the oracle's
# ---------------------------------------------------------------------------------------------------------------------------
def oracle_batch(model, prm, B, run):
    """run(solver, pick) on one oracle solver per instance (pick(x) = instance b's slice of a per-instance array) and collect
    what assert_end_to_end compares"""
    out = dict(iters=np.zeros(B, dtype=np.int64), converged=np.zeros(B, dtype=bool), primal_infeasible=np.zeros(B, dtype=bool),
               z=np.zeros((B, model.nv)), nu=np.zeros((B, model.nv)), primal_residual=np.zeros(B), dual_residual=np.zeros(B))
    solvers = []
    for b in range(B):
        s = ref.RefSolver(model, **prm)
        run(s, lambda x, b=b: x[b])
        out["iters"][b] = s.get_iter(); out["converged"][b] = s.get_convergence_status()
        out["primal_infeasible"][b] = s.get_primal_infeasibility_status()
        out["z"][b] = s.z; out["nu"][b] = s.nu
        out["primal_residual"][b] = s.scalar("primal_residual"); out["dual_residual"][b] = s.scalar("dual_residual")
        solvers.append(s)
    return out, solvers


ENGINE_KW = {"default": {}, "solve_only": dict(tail_max_instances=-1), "handover": dict(max_launch_iters=3, tail_max_instances=1 << 20),
             "lean": {}}   # ("lean": LOIKB_FLAT=0 -- the per-link instantiations of k_hslots + k_lean, the engine of robots outside the flat engine's domain)
REF_FIELDS = ["nu", "z", "w", "vis", "fis", "g", "yis", "Aty", "Stf_plus_w", "dual_residual_vec", "primal_residual_vec"]
REF_SCALARS = ["primal_residual", "dual_residual", "dual_residual_v", "dual_residual_nu", "mu", "Href_v_inf_norm", "g_inf_norm",
               "delta_fis_inf_norm", "tol_dual", "tol_primal"]


def _model_for(which, request):
    if which == "talos":
        model = request.getfixturevalue("talos"); link = model.getJointId("arm_left_7_joint")
    elif which == "tree":
        model = random_tree(13, 23); link = model.njoints - 1
    else:
        model = random_tree_multidof(seed=5, nb=9, root_freeflyer=True, n_spherical=1, n_translation=1)
        link = model.njoints - 1
    return model, link


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["full", "diag"])
@pytest.mark.parametrize("which", ["talos", "tree", "multidof"])
def test_gpu_per_link_references_iteration_by_iteration(which, kind, request):
    """the state after 1, 3, 6 iterations, field by field (tol_rel > 0: tol_dual depends on |H_i v_i| and on Hv_inf_norm_)"""
    from loik_amd import workloads
    model, link = _model_for(which, request)
    tol = 1e-7 if which == "multidof" else 1e-9
    B = 48
    wl = workloads.make_workload(model, B, link, 44, bound=0.5, snap_prob=0.0, nu_scale=0.4)
    wl["H_ref"], wl["v_ref"] = 2.0 * np.eye(6), np.full(6, 0.1)      # what SolveInit broadcasts first (Hv_inf_norm_ = 0.2)
    H, v = per_link_references(model, 7, kind)
    for k in (1, 3, 6):
        prm = dict(FIXTURE, max_iter=k + 1, tol_abs=0.0, tol_rel=1e-30, tol_primal_inf=0.0)
        s = loik_amd.BatchedLoik(model, B, **prm)
        s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        s.UpdateReferences(H, v)
        s.Solve()
        got = {n: s.get(n) for n in REF_FIELDS + REF_SCALARS}
        got["His"] = s.His_full()
        for b in range(0, B, 7):
            r = ref.RefSolver(model, **prm)
            r.SolveInit(*problem_args(wl, b)); r.UpdateReferences(H, v); r.Solve()
            for n in REF_FIELDS:
                want = r.field(n)
                if n in ("vis", "fis", "g"):
                    want = want[1:]
                assert_close(got[n][b], want, tol, "%s b%d k%d" % (n, b, k))
            assert_close(got["His"][b], r.His[1:], tol, "His")
            for n in REF_SCALARS:
                assert_close(got[n][b], r.scalar(n), tol, "%s b%d k%d" % (n, b, k))
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("engine", sorted(ENGINE_KW))
@pytest.mark.parametrize("which", ["talos", "tree", "multidof"])
def test_gpu_per_link_references_end_to_end(which, engine, request, monkeypatch):
    from helpers import assert_end_to_end, fetch_end_to_end
    from loik_amd import workloads
    if engine == "lean":
        monkeypatch.setenv("LOIKB_FLAT", "0")
    model, link = _model_for(which, request)
    B = 200
    wl = workloads.make_workload(model, B, link, 45, bound=0.5, snap_prob=0.0, nu_scale=0.4)
    H, v = per_link_references(model, 9, "full")
    prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)

    def run_oracle(s, pick):
        s.SolveInit(wl["q"][pick(np.arange(B))], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], pick(wl["bis"]), wl["lb"], wl["ub"])
        s.UpdateReferences(H, v)
        s.Solve()
    out, _ = oracle_batch(model, prm, B, run_oracle)
    s = loik_amd.BatchedLoik(model, B, **prm, **ENGINE_KW[engine])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    s.UpdateReferences(H, v)
    if which == "talos" and engine == "default":
        assert "per-link references in force" in s.plan()
    s.Solve()
    loose = which == "multidof"
    assert_end_to_end(fetch_end_to_end(s, residuals=not loose), out, prm, same_frac=0.99, ztol=1e-9 if loose else 2e-10,
                      off_ztol=1e-5, what="%s %s" % (which, engine))
    st = s.stats()
    if which == "talos" and engine == "lean":
        assert st["lean_launches"] >= 1 and st["flat_launches"] == 0 and st["tail_instances"] == B and "k_lean" in s.plan(), (st, s.plan())
    if which == "talos" and engine == "default":
        # ordinary API use stays on the fast engine: k_flat2's per-link instantiation reads the links' table (HM = 3)
        assert st["flat_launches"] >= 1 and st["tail_instances"] == B, (st, s.plan())
        assert "the flat engine reads the links' table" in s.plan()
    # the next SolveInit broadcasts one pair again (hpp:355) and the lean engine is back
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    out0 = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                           nthreads=4, want_nu=True, **prm)
    assert_end_to_end(fetch_end_to_end(s, residuals=not loose), out0, prm, same_frac=0.99, ztol=1e-9 if loose else 2e-10,
                      off_ztol=1e-5, what="%s %s broadcast again" % (which, engine))
    assert "per-link" not in s.plan()
    s.close()


@pytest.mark.gpu
def test_gpu_update_references_errors_and_quirks(talos):
    wl = feasible_batch(talos, 4, talos.njoints - 1, 3)
    s = loik_amd.BatchedLoik(talos, 4, **dict(FIXTURE, max_iter=5))
    H, v = per_link_references(talos, 1, "diag")
    with pytest.raises(loik_amd.LoikError) as e:      # before SolveInit
        s.UpdateReferences(H, v)
    assert e.value.code == -24
    wl["H_ref"], wl["v_ref"] = 2.0 * np.eye(6), np.full(6, 0.25)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    with pytest.raises(loik_amd.LoikError) as e:      # one entry per joint incl. the universe (hpp:105-107)
        s.UpdateReferences(H[1:], v[1:])
    assert e.value.code == -8
    Hbad = H.copy(); Hbad[3, 0, 1] = 0.3
    with pytest.raises(loik_amd.LoikError) as e:
        s.UpdateReferences(Hbad, v)
    assert e.value.code == -23
    # Hv_inf_norm_ only grows in UpdateReferences (hpp:114-116): visible through tol_dual with tol_rel > 0
    r = ref.RefSolver(talos, **dict(FIXTURE, max_iter=5))
    r.SolveInit(*problem_args(wl, 0))
    for scale in (0.01, 100.0):
        s.UpdateReferences(scale * H, v); r.UpdateReferences(scale * H, v)
        s.Solve(); r.Solve()
        assert_close(s.get("tol_dual")[0], r.scalar("tol_dual"), 1e-12, "tol_dual")
    s.close()


def _edit_sequence(model, full, links, B, shared_A, warm):
    """the same editing session on any solver with the reference's method names; pick(x) = this solver's slice of a
    per-instance array.  Starts with {a, b}, capacity 3."""
    a, b_, c = links
    A = {l: (full["Ais"][k] if shared_A else full["Ais"][:, k]) for k, l in enumerate(links)}
    bv = {l: full["bis"][:, k] for k, l in enumerate(links)}
    pa = (lambda x, pick: x) if shared_A else (lambda x, pick: pick(x))
    checkpoints = []

    def run(s, pick, snap):
        def A2(ls):
            return np.stack([pa(A[l], pick) for l in ls], axis=-3)
        def b2(ls):
            return np.stack([pick(bv[l]) for l in ls], axis=-2)
        q = pick(full["q"])
        s.Solve(q, full["H_ref"], full["v_ref"], np.array([a, b_], dtype=np.int32), A2([a, b_]), b2([a, b_]), full["lb"], full["ub"])
        snap("init {a,b}")
        s.AddEqConstraint(c, pa(A[c], pick), pick(bv[c]))
        assert s.active_task_constraint_ids() == [a, b_, c]
        s.Solve(q, -1, None, None); snap("add c")
        assert s.RemoveEqConstraint(b_) and not s.RemoveEqConstraint(b_)
        assert s.active_task_constraint_ids() == [a, c]
        s.Solve(q, -1, None, None); snap("remove b")
        s.AddEqConstraint(c, pa(A[b_], pick), pick(bv[b_]))           # present: UpdateEqConstraint (hpp:250-253)
        s.UpdateEqConstraint(a, pick(bv[c]))                          # (c_id, bi) overload keeps A (hpp:224-238)
        s.Solve(q, a, pa(A[a], pick), pick(bv[a])); snap("update, tailored")
        s.AddEqConstraint(b_, pa(A[b_], pick), pick(bv[b_]))          # the freed slot is taken again: order [a, c, b]
        assert s.active_task_constraint_ids() == [a, c, b_]
        s.Solve(q, -1, None, None); snap("add b back")
        s.RemoveEqConstraint(a); s.RemoveEqConstraint(c); s.RemoveEqConstraint(b_)
        s.Solve(q, -1, None, None); snap("none left")
    return run


@pytest.mark.gpu
@pytest.mark.parametrize("warm", [False, True])
@pytest.mark.parametrize("shared_A", [True, False])
@pytest.mark.parametrize("which", ["talos", "tree"])
def test_gpu_constraint_editing_session(which, shared_A, warm, request):
    """Add / Remove / Update between tailored solves, compared with the oracle's editing functions after every solve (all
    instances).  talos runs in k_lean (null slots in its constraint blocks), tree in k_solve + k_tail."""
    from helpers import assert_end_to_end, fetch_end_to_end
    if which == "talos":
        model = request.getfixturevalue("talos")
        links = [model.getJointId("arm_left_7_joint"), model.getJointId("arm_right_7_joint"), model.getJointId("leg_left_6_joint")]
    else:
        model = random_tree(8, 12, branch_prob=0.4); links = [5, 9, 12]
    B = 96
    full = multi_task_batch(model, B, links, 3, bound=0.5, nu_scale=0.4, per_instance_A=not shared_A)
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0, num_eq_c=2, eq_c_capacity=3, warm_start=warm)
    run = _edit_sequence(model, full, links, B, shared_A, warm)
    # oracle: one solver per instance, snapshots after every solve
    snaps_o = {}
    for b in range(B):
        r = ref.RefSolver(model, **prm)

        def snap(name, r=r, b=b):
            d = snaps_o.setdefault(name, dict(iters=np.zeros(B, dtype=np.int64), converged=np.zeros(B, dtype=bool),
                                              primal_infeasible=np.zeros(B, dtype=bool), z=np.zeros((B, model.nv)),
                                              nu=np.zeros((B, model.nv)), primal_residual=np.zeros(B), dual_residual=np.zeros(B),
                                              yis=[None] * B))
            d["iters"][b] = r.get_iter(); d["converged"][b] = r.get_convergence_status()
            d["primal_infeasible"][b] = r.get_primal_infeasibility_status(); d["z"][b] = r.z; d["nu"][b] = r.nu
            d["primal_residual"][b] = r.scalar("primal_residual"); d["dual_residual"][b] = r.scalar("dual_residual")
            d["yis"][b] = r.yis.copy()
        run(r, lambda x, b=b: x[b], snap)
    s = loik_amd.BatchedLoik(model, B, **prm)
    order = []

    def snap_g(name):
        order.append(name)
        out = snaps_o[name]
        same = assert_end_to_end(fetch_end_to_end(s, residuals=True), out, prm, same_frac=0.99, ztol=2e-9, off_ztol=1e-5,
                                 what="%s: %s" % (which, name), res_tol=(1e-7, 1e-5))
        y = s.get("yis")
        want = np.array([out["yis"][b] for b in range(B)]).reshape(B, -1, 6)
        assert y.shape == want.shape, (name, y.shape, want.shape)
        if want.size:
            scale = 1.0 + np.abs(want).max()
            assert np.abs(y - want)[same].max() < 1e-6 * scale, (name, np.abs(y - want)[same].max())
    run(s, lambda x: x, snap_g)
    assert len(order) == 6
    if which == "talos" and shared_A:
        assert s.stats()["lean_launches"] > 0, s.plan()
    s.close()


@pytest.mark.gpu
def test_gpu_constraint_editing_errors(talos):
    links = [talos.getJointId("arm_left_7_joint"), talos.getJointId("arm_right_7_joint")]
    wl = multi_task_batch(talos, 4, links, 3)
    s = loik_amd.BatchedLoik(talos, 4, **dict(FIXTURE, max_iter=5, num_eq_c=2))          # capacity = num_eq_c
    with pytest.raises(loik_amd.LoikError) as e:
        s.AddEqConstraint(3, wl["Ais"][0], wl["bis"][:, 0])
    assert e.value.code == -24                                                           # before SolveInit
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    with pytest.raises(loik_amd.LoikError) as e:
        s.AddEqConstraint(3, wl["Ais"][0], wl["bis"][:, 0])
    assert e.value.code == -2                                                            # no free slot
    with pytest.raises(loik_amd.LoikError) as e:
        s.UpdateEqConstraint(3, wl["bis"][:, 0])
    assert e.value.code == -4                                                            # hpp:184-186
    assert s.RemoveEqConstraint(links[0])
    with pytest.raises(loik_amd.LoikError) as e:                                         # SolveInit checks against nc_eq_ (hpp:143)
        s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    assert e.value.code == -2
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"][1:], wl["Ais"][1:], wl["bis"][:, 1:], wl["lb"], wl["ub"])
    with pytest.raises(loik_amd.LoikError) as e:
        s.AddEqConstraint(talos.njoints, wl["Ais"][0], wl["bis"][:, 0])                  # link id out of range
    assert e.value.code == -20
    s.close()
