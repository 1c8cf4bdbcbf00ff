"""LoikSolverInfo (SURVEY 8(a) a15 / section 5 "Metrics / logging"): the per-iteration lists the reference fills when it is
constructed with logging = true (/root/reference/include/loik/loik-loid-optimized.hpp:406-420, lists declared :47-127 and
task-solver-base.hpp:25-52).  Upstream keeps the struct protected without accessor; here the oracle and the C-ABI expose it."""
import numpy as np
import pytest

import loik_amd
from helpers import FIXTURE, assert_close, feasible_batch, fixture_problem, problem_args, random_tree, random_tree_multidof
from oracle import ref

LISTS = ["primal_residual_task_list", "primal_residual_slack_list", "primal_residual_list", "dual_residual_nu_list",
         "dual_residual_v_list", "dual_residual_list", "mu_list", "mu_eq_list", "mu_ineq_list"]


def test_oracle_lists_follow_the_main_loop(talos):
    wl = feasible_batch(talos, 3, talos.getJointId("arm_left_7_joint"), 5)
    prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
    for b in range(3):
        s = ref.RefSolver(talos, **prm)
        s.Solve(*problem_args(wl, b))
        n = s.get_iter() - int(s.scalar("tail_solve_iter"))
        lists = [s.solver_info(k) for k in range(9)]
        assert all(len(x) == n for x in lists)
        assert lists[6][0] == prm["mu"] and np.all(lists[7] == prm["mu_equality_scale_factor"] * lists[6]) and np.all(lists[8] == lists[6])
        if s.scalar("tail_solve_iter") == 0:     # the last entries are the solver's final residuals
            assert lists[2][-1] == s.scalar("primal_residual") and lists[5][-1] == s.scalar("dual_residual")
        assert np.all(lists[2] == np.maximum(lists[0], lists[1])) and np.all(lists[5] == np.maximum(lists[3], lists[4]))
        # a second solve starts new lists (upstream's Solve() never clears them: not replicated)
        s.Solve()
        assert len(s.solver_info(0)) == n


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["talos", "tree", "floating_base"])
def test_gpu_solver_info_matches_oracle(which, request):
    if which == "talos":
        model = request.getfixturevalue("talos"); link = model.getJointId("arm_left_7_joint")
    elif which == "floating_base":   # a multi-DoF joint: logged solves of any model the solver accepts (loik-loid-optimized.hpp:406-420)
        model = loik_amd.builtin_model("talos32_freeflyer"); link = model.getJointId("arm_left_7_joint")
    else:
        model = random_tree(13, 23); link = model.njoints - 1
    B = 48
    if which == "floating_base":
        from loik_amd import workloads
        wl = workloads.make_workload(model, B, link, 5, bound=0.5, snap_prob=0.2, nu_scale=0.4)
    else:
        wl = feasible_batch(model, B, link, 5)
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0)
    s = loik_amd.BatchedLoik(model, B, logging=True, **prm)
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    info = s.solver_info()
    it, z, tail = s.get("iter"), s.get("z"), s.get("tail_solve_iter")
    for b in range(B):
        r = ref.RefSolver(model, **prm)
        r.Solve(*problem_args(wl, b))
        assert it[b] == r.get_iter() and tail[b] == int(r.scalar("tail_solve_iter")), (b, it[b], r.get_iter())
        n = len(r.solver_info(0))
        assert info["rows"][b] == n
        for k, name in enumerate(LISTS):
            assert_close(info[name][b, :n], r.solver_info(k), 1e-9, "%s b%d" % (name, b))
            assert np.all(info[name][b, n:] == 0.0)
        assert_close(z[b], r.z, 1e-9, "z")
    # a repeated Solve() and a tailored solve refill the lists from the start
    s.Solve()
    assert np.array_equal(s.solver_info()["rows"], info["rows"])
    s.Solve(wl["q"], link, wl["Ais"], 0.5 * wl["bis"][:, 0])
    r = ref.RefSolver(model, **prm)
    r.Solve(*problem_args(wl, 0)); r.Solve(); r.Solve(wl["q"][0], link, wl["Ais"][0], 0.5 * wl["bis"][0, 0])
    i2 = s.solver_info()
    assert i2["rows"][0] == len(r.solver_info(2))
    assert_close(i2["primal_residual_list"][0, :i2["rows"][0]], r.solver_info(2), 1e-9, "tailored")
    # the logged solve runs the same problem as the engines: same answers from a handle without logging
    s2 = loik_amd.BatchedLoik(model, B, **prm)
    s2.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    same = s.get("iter") == s2.get("iter")
    assert same.mean() >= 0.97 and np.abs(s.get("z") - s2.get("z"))[same].max() < 1e-9
    s.close(); s2.close()


@pytest.mark.gpu
def test_gpu_solver_info_from_the_flat_engine(talos):
    """logging = 1 keeps a solve on the fast engine: its LOG build writes the nine lists itself (one more fold per iteration); the same
    lists, rows and results as the oracle's, instance by instance -- incl. instances that end in the infeasibility tail solve,
    whose iterations append nothing (hpp:271-319)"""
    from loik_amd import workloads
    B = 256
    wl = workloads.talos_c3(B, seed=77)
    prm = dict(wl["params"], max_iter=300)
    s = loik_amd.BatchedLoik(talos, B, logging=True, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    assert "the flat engine writes the SolverInfo lists" in s.plan(), s.plan()
    s.Solve()
    st = s.stats()
    assert st["flat_launches"] >= 1 and st["tail_instances"] == B and st["lean_escaped"] == 0, st
    info = s.solver_info()
    it, z, tail = s.get("iter"), s.get("z"), s.get("tail_solve_iter")
    assert (s.get("primal_infeasible") > 0).sum() >= 5 and (tail > 0).sum() >= 1, "the sample should contain tail solves"
    for b in range(B):
        r = ref.RefSolver(talos, **prm)
        r.Solve(*problem_args(wl, b))
        assert it[b] == r.get_iter() and tail[b] == int(r.scalar("tail_solve_iter")), (b, it[b], r.get_iter())
        n = len(r.solver_info(0))
        assert info["rows"][b] == n == it[b] - tail[b]
        for k, name in enumerate(LISTS):
            assert_close(info[name][b, :n], r.solver_info(k), 1e-9, "%s b%d" % (name, b))
            assert np.all(info[name][b, n:] == 0.0)
        assert_close(z[b], r.z, 1e-9, "z")
    # the lists restart with every solve; a handle without logging gives the same answers (the same kernel without the lists)
    s.Solve()
    assert np.array_equal(s.solver_info()["rows"], info["rows"])
    s2 = loik_amd.BatchedLoik(talos, B, **prm)
    s2.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    assert np.array_equal(s.get("iter"), s2.get("iter")) and np.abs(s.get("z") - s2.get("z")).max() < 1e-9
    s.close(); s2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("robot,weight", [("talos32", "diagonal"), ("talos32", "general"), ("talos32", "per_link"), ("talos44", "scalar"),
                                          ("talos44", "general")])
def test_gpu_solver_info_from_the_flat_engines_with_any_reference_weight(robot, weight):
    """a logged solve whose reference weight is not h I -- diagonal, general, per link -- and a logged solve of a
    33..64-joint robot stay on the on-chip engines (the LOG builds of k_flat2 / k_flat1) instead of the pass-by-pass
    implementation: the nine lists, their row counts and the results equal the oracle's, instance by instance"""
    from loik_amd import workloads
    model = loik_amd.builtin_model(robot)
    B = 96
    wl = workloads.talos_wholebody(B, seed=31, model=model) if robot == "talos44" else workloads.talos_c3(B, seed=31)
    prm = dict(wl["params"], max_iter=250)
    Href = np.diag([0.4, 1.5, 0.7, 3.0, 0.2, 2.2])
    if weight == "general":
        Q = np.linalg.qr(np.random.default_rng(4).normal(size=(6, 6)))[0]
        Href = Q @ Href @ Q.T
        Href = 0.5 * (Href + Href.T)
    if weight == "scalar":
        Href = wl["H_ref"]
    vref = np.array([0.02, -0.01, 0.03, 0.05, -0.04, 0.01])
    s = loik_amd.BatchedLoik(model, B, logging=True, **prm)
    s.SolveInit(wl["q"], Href, vref, wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    H_list = v_list = None
    if weight == "per_link":
        rng = np.random.default_rng(8)
        H_list = np.stack([np.diag(rng.uniform(0.2, 3.0, size=6)) for _ in range(model.njoints)])
        v_list = rng.uniform(-0.05, 0.05, size=(model.njoints, 6))
        s.UpdateReferences(H_list, v_list)
    assert "the flat engine writes the SolverInfo lists" in s.plan(), s.plan()
    s.Solve()
    st = s.stats()
    assert st["flat_launches"] >= 1 and st["tail_instances"] == B and st["lean_escaped"] == 0, (s.plan(), st)
    info = s.solver_info()
    assert info["truncated_instances"] == 0
    it, z, tail = s.get("iter"), s.get("z"), s.get("tail_solve_iter")
    for b in range(B):
        r = ref.RefSolver(model, **prm)
        r.SolveInit(wl["q"][b], Href, vref, wl["c_ids"], wl["Ais"], wl["bis"][b], wl["lb"], wl["ub"])
        if weight == "per_link":
            r.UpdateReferences(H_list, v_list)
        r.Solve()
        assert it[b] == r.get_iter() and tail[b] == int(r.scalar("tail_solve_iter")), (b, it[b], r.get_iter())
        n = len(r.solver_info(0))
        assert info["rows"][b] == n == it[b] - tail[b]
        for k, name in enumerate(LISTS):
            assert_close(info[name][b, :n], r.solver_info(k), 1e-9, "%s b%d" % (name, b))
            assert np.all(info[name][b, n:] == 0.0)
        assert_close(z[b], r.z, 1e-9, "z")
    s.close()


@pytest.mark.gpu
def test_gpu_solver_info_is_complete_when_mu_leaves_the_precomputed_decades(talos, monkeypatch):
    """an instance whose mu leaves the flat engine's precomputed decades is finished by k_tail, which writes no
    SolverInfo lists.  A cold logged solve in which that happens is repeated on the pass-by-pass implementation: complete lists,
    equal to the oracle's; a warm-started one cannot be repeated and says how many instances' lists are short"""
    from loik_amd import workloads
    monkeypatch.setenv("LOIKB_LEAN_KLO", "0")
    monkeypatch.setenv("LOIKB_LEAN_DECADES", "2")      # decades 0..1 only: a quarter of the headline's instances visit decade 2
    monkeypatch.setenv("LOIKB_LEAN_ADAPT", "0")
    B = 128
    wl = workloads.talos_c3(B, seed=78)
    prm = dict(wl["params"], max_iter=200)
    s = loik_amd.BatchedLoik(talos, B, logging=True, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    s.Solve()
    info = s.solver_info()
    assert info["truncated_instances"] == 0
    it, tail = s.get("iter"), s.get("tail_solve_iter")
    assert (np.round(np.log10(s.get("mu") / prm["mu"])) >= 2).sum() >= 5, "the sample should contain instances beyond decade 1"
    for b in range(B):
        r = ref.RefSolver(talos, **prm)
        r.Solve(*problem_args(wl, b))
        n = len(r.solver_info(0))
        assert it[b] == r.get_iter() and info["rows"][b] == n == it[b] - tail[b], (b, it[b], r.get_iter(), info["rows"][b], n)
        for k, name in enumerate(LISTS):
            assert_close(info[name][b, :n], r.solver_info(k), 1e-9, "%s b%d" % (name, b))
    s.close()
    # the same through the entries that begin with Reset(false) + FwdPassInit -- Solve(q, H_ref, ...) and the
    # tailored Solve(q, c_id, Ai, bi) of a cold-start handle, the latter as the SECOND solve of the handle (the first one left duals
    # behind): the repeat starts from yis = Aty = 0 like the first attempt (optimized.hxx:270-278), not from the first attempt's duals
    wl2 = workloads.talos_c3(B, seed=79)
    for entry in ("full", "tailored"):
        s = loik_amd.BatchedLoik(talos, B, logging=True, **prm)
        if entry == "full":
            s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
            cur = wl
        else:
            s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
            s.Solve(wl2["q"], int(wl["c_ids"][0]), wl2["Ais"][0], wl2["bis"][:, 0])
            cur = wl2
        info = s.solver_info()
        assert info["truncated_instances"] == 0, entry
        assert (np.round(np.log10(s.get("mu") / prm["mu"])) >= 2).sum() >= 5, "the sample should contain instances beyond decade 1 (%s)" % entry
        it, tail = s.get("iter"), s.get("tail_solve_iter")
        for b in range(0, B, 2):
            r = ref.RefSolver(talos, **prm)
            r.Solve(*problem_args(wl, b))
            if entry == "tailored":
                r.Solve(wl2["q"][b], int(wl["c_ids"][0]), wl2["Ais"][0], wl2["bis"][b, 0])
            n = len(r.solver_info(0))
            assert it[b] == r.get_iter() and info["rows"][b] == n == it[b] - tail[b], (entry, b, it[b], r.get_iter(), info["rows"][b], n)
            for k, name in enumerate(LISTS):
                assert_close(info[name][b, :n], r.solver_info(k), 1e-9, "%s %s b%d" % (entry, name, b))
        s.close()
    # warm start: the flat engine's lists of the escaped instances end early, and the handle says so
    s = loik_amd.BatchedLoik(talos, B, logging=True, **dict(prm, warm_start=True))
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    s.Solve(wl["q"], int(wl["c_ids"][0]), wl["Ais"][0], wl["bis"][:, 0])
    info = s.solver_info()
    st = s.stats()
    assert st["lean_escaped"] > 0 and info["truncated_instances"] == st["lean_escaped"], (st["lean_escaped"], info["truncated_instances"])
    s.close()


@pytest.mark.gpu
def test_gpu_solver_info_infeasible_fixture_and_errors(talos):
    """the reference fixture's unreachable head target: the lists stop at the iteration that raises the certificate, the tail
    solve only counts (hpp:286-290)"""
    p = fixture_problem(talos, bound=4.0)
    prm = dict(FIXTURE, max_iter=200)
    s = loik_amd.BatchedLoik(talos, 1, logging=True, **prm)
    s.Solve(*problem_args(p))
    r = ref.RefSolver(talos, **prm)
    r.Solve(*problem_args(p))
    assert s.get("primal_infeasible")[0] == 1 and r.get_primal_infeasibility_status()
    info = s.solver_info()
    assert info["rows"][0] == len(r.solver_info(0)) == r.get_iter() - int(r.scalar("tail_solve_iter"))
    assert s.get("tail_solve_iter")[0] == int(r.scalar("tail_solve_iter")) and s.get("iter")[0] == r.get_iter()
    assert_close(info["mu_list"][0, :info["rows"][0]], r.solver_info(6), 1e-12, "mu_list")
    s.close()
    s = loik_amd.BatchedLoik(talos, 1, **prm)            # no logging: no lists
    s.Solve(*problem_args(p))
    with pytest.raises(loik_amd.LoikError) as e:
        s.solver_info()
    assert e.value.code == -24
    s.close()
    m = random_tree_multidof(seed=5, nb=9, root_freeflyer=True, n_spherical=1, n_translation=1)
    wl = feasible_batch(m, 2, m.njoints - 1, 3)
    s = loik_amd.BatchedLoik(m, 2, logging=True, **prm)   # multi-DoF joints (free-flyer, spherical, translation): logged like any model
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    for b in range(2):
        r = ref.RefSolver(m, **prm)
        r.Solve(*problem_args(wl, b))
        n = len(r.solver_info(0))
        assert s.solver_info()["rows"][b] == n and s.get("iter")[b] == r.get_iter()
        assert_close(s.solver_info()["dual_residual_list"][b, :n], r.solver_info(5), 1e-8, "dual_residual_list, multi-DoF tree")
    s.close()


@pytest.mark.gpu
def test_gpu_logged_solve_leaves_the_solver_state_in_the_tiles(talos):
    """a handle with logging = 1 solves on the pass-by-pass implementation; what follows -- a warm-started tailored Solve
    (Reset(true), loik-loid-data-optimized.hxx:114-127), loikb_integrate, the getters of the engines' own fields -- continues
    from that solve's result: the same sequence with logging off gives the same numbers.  Also: the lists' capacity is the
    library's (set_max_iter between the solve and the read changes nothing)."""
    link = talos.getJointId("arm_left_7_joint")
    B = 40
    wl = feasible_batch(talos, B, link, 31, nu_scale=0.5)
    wl2 = feasible_batch(talos, B, link, 32, nu_scale=0.5)
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0, warm_start=True)
    res = {}
    for logging in (False, True):
        s = loik_amd.BatchedLoik(talos, B, logging=logging, **prm)
        s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        first = {n: s.get(n) for n in ["z", "iter", "status", "mu", "primal_residual", "mu_updates"] if not (logging and n == "mu_updates")}
        if logging:
            info = s.solver_info()
            s.set_max_iter(20)                      # (the stored lists keep the capacity of the solve that filled them)
            info2 = s.solver_info()
            assert info["primal_residual_list"].shape == info2["primal_residual_list"].shape == (B, 299)
            assert np.array_equal(info["primal_residual_list"], info2["primal_residual_list"])
            s.set_max_iter(300)
        s.integrate(0.05)                            # q <- q (+) z dt on the device: reads z from the tiles
        q1 = s.get("q")
        # warm-started tailored solve on the integrated configuration with a new target for the constraint
        s.Solve(None, int(wl["c_ids"][0]), wl["Ais"][0], wl2["bis"][:, 0])
        second = {n: s.get(n) for n in ["z", "iter", "status", "w"]}
        res[logging] = (first, q1, second)
        s.close()
    (f0, q0, s0), (f1, q1, s1) = res[False], res[True]
    assert np.array_equal(f0["iter"], f1["iter"]) and np.array_equal(f0["status"], f1["status"])
    assert np.abs(f0["z"] - f1["z"]).max() < 1e-9 and np.allclose(f0["mu"], f1["mu"], rtol=1e-12)
    assert np.abs(q0 - q1).max() < 1e-10, "integrate after a logged solve used a stale z"
    same = s0["iter"] == s1["iter"]
    assert same.mean() >= 0.95, "the warm start after a logged solve did not continue from its duals"
    assert np.abs(s0["z"] - s1["z"])[same].max() < 1e-8 and np.abs(s0["w"] - s1["w"])[same].max() < 1e-6
