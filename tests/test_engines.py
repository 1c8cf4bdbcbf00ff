"""The same solves through every execution path of the library: the lean tail kernel (default for fp64 batches: two
wavefronts per SIMD, decade slots of H precomputed), the one-wavefront-per-SIMD tail kernel (LOIKB_LEAN=0), the solve
kernel alone (tail_max_instances < 0), the hybrid of solve kernel and tail kernel, and the lean kernel with too few
precomputed decades (instances "escape" and are finished by the other tail kernel).  All must agree with the oracle."""
import numpy as np
import pytest

import loik_amd
from helpers import FIXTURE, assert_close, assert_end_to_end, feasible_batch, fetch_end_to_end, problem_args, random_tree
from oracle import ref

pytestmark = pytest.mark.gpu

ENGINES = {
    "flat": (dict(), dict()),
    "flat_one_lane": (dict(LOIKB_FLAT_SPLIT="0"), dict()),
    # k_flat2's time slicing forced, one wavefront per CU so that instances wait: requeues at every 5th iteration
    "flat_sliced": (dict(LOIKB_FLAT_SLICE="5", LOIKB_LEAN_WG_PER_CU="1"), dict()),
    "lean": (dict(LOIKB_FLAT="0"), dict()),
    "tail": (dict(LOIKB_LEAN="0"), dict(tail_max_instances=1 << 20)),
    "solve": (dict(), dict(tail_max_instances=-1)),
    "hybrid": (dict(LOIKB_LEAN="0"), dict(tail_max_instances=120, max_launch_iters=2)),
    "hybrid_flat": (dict(), dict(tail_max_instances=120, max_launch_iters=2)),
    "hybrid_lean": (dict(LOIKB_FLAT="0"), dict(tail_max_instances=120, max_launch_iters=2)),
    "flat_escapes": (dict(LOIKB_LEAN_KLO="0", LOIKB_LEAN_DECADES="2"), dict()),
    # round 5: the decades the table lacks are built by the instance's own wavefront (flat_build_slot), also across time slices
    "flat_builds": (dict(LOIKB_LEAN_KLO="0", LOIKB_LEAN_DECADES="1", LOIKB_LEAN_ADAPT="0", LOIKB_FLAT_BUILD="1"), dict()),
    "flat_builds_sliced": (dict(LOIKB_LEAN_KLO="1", LOIKB_LEAN_DECADES="1", LOIKB_LEAN_ADAPT="0", LOIKB_FLAT_BUILD="1", LOIKB_FLAT_SLICE="5",
                                LOIKB_LEAN_WG_PER_CU="1"), dict()),
    "lean_escapes": (dict(LOIKB_FLAT="0", LOIKB_LEAN_KLO="0", LOIKB_LEAN_DECADES="2"), dict()),
}
FIELDS = ["nu", "z", "w", "vis", "fis", "g", "yis", "Aty", "Stf_plus_w", "primal_residual_vec", "dual_residual_vec"]
SCALARS = ["primal_residual", "dual_residual", "primal_residual_task", "primal_residual_slack", "dual_residual_v",
           "dual_residual_nu", "mu", "delta_fis_inf_norm", "delta_yis_inf_norm", "delta_w_inf_norm", "delta_vis_inf_norm",
           "delta_nu_inf_norm", "Av_inf_norm", "nu_inf_norm", "Href_v_inf_norm", "g_inf_norm", "Stf_plus_w_inf_norm",
           "tol_primal", "tol_dual"]


def _solver(model, B, prm, engine, monkeypatch):
    env, kw = ENGINES[engine]
    for k in ("LOIKB_LEAN", "LOIKB_FLAT", "LOIKB_FLAT_SPLIT", "LOIKB_FLAT_SLICE", "LOIKB_LEAN_WG_PER_CU", "LOIKB_LEAN_KLO", "LOIKB_LEAN_DECADES",
              "LOIKB_LEAN_ADAPT", "LOIKB_FLAT_BUILD"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    return loik_amd.BatchedLoik(model, B, **prm, **kw)


@pytest.mark.parametrize("engine", list(ENGINES))
@pytest.mark.parametrize("which", ["talos", "tree"])
def test_every_engine_matches_the_oracle(which, engine, request, monkeypatch):
    model = random_tree(6, 21) if which == "tree" else request.getfixturevalue("talos")
    link = model.njoints - 1 if which == "tree" else model.getJointId("arm_left_7_joint")
    B = 200
    wl = feasible_batch(model, B, link, 91, nu_scale=0.5)
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    for k in (1, 2, 5, 12):
        prm = dict(FIXTURE, max_iter=k + 1, tol_abs=0.0, tol_rel=1e-30, tol_primal_inf=0.0)
        s = _solver(model, B, prm, engine, monkeypatch)
        s.Solve(*args)
        got = {n: s.get(n) for n in FIELDS + SCALARS}
        got["His"] = s.His_full()
        assert np.all(s.get("iter") == k)
        for b in range(0, B, 23):
            r = ref.RefSolver(model, **prm)
            r.Solve(*problem_args(wl, b))
            for n in FIELDS:
                want = r.field(n)
                if n in ("vis", "fis", "g"):
                    want = want[1:]
                assert_close(got[n][b], want, 1e-9, "%s b%d k%d %s" % (n, b, k, engine))
            assert_close(got["His"][b], r.His[1:], 1e-9, "His")
            for n in SCALARS:
                assert_close(got[n][b], r.scalar(n), 1e-9, "%s b%d k%d %s" % (n, b, k, engine))
        s.close()
    prm = dict(FIXTURE, max_iter=500, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, *args[:4], wl["Ais"], wl["bis"], wl["lb"], wl["ub"], nthreads=4, want_nu=True, **prm)
    s = _solver(model, B, prm, engine, monkeypatch)
    s.Solve(*args)
    st = s.stats()
    if engine in ("lean", "flat"):
        assert st["lean_launches"] >= 1 and st["lean_escaped"] == 0 and st["tail_instances"] == B
        # (the random tree's joints are numbered depth-first too: both run the engine they ask for)
        assert st["flat_launches"] == (1 if engine == "flat" else 0), (s.plan(), st)
    if engine == "flat_one_lane":
        assert st["flat_launches"] >= 1 and st["flat_split_launches"] == 0 and st["tail_instances"] == B, (s.plan(), st)
    if engine == "flat_sliced" and which == "talos":
        assert st["flat_split_launches"] >= 1, (s.plan(), st)
    if engine in ("tail", "hybrid"):
        assert st["lean_launches"] == 0 and st["tail_instances"] > 0
    if engine == "solve":
        assert st["tail_instances"] == 0
    if engine in ("hybrid_lean", "hybrid_flat"):
        assert st["lean_launches"] >= 1 and 0 < st["tail_instances"] < B
        assert (st["flat_launches"] >= 1) == (engine == "hybrid_flat")
    if engine in ("flat_builds", "flat_builds_sliced") and which == "talos":
        assert st["flat_split_launches"] >= 1 and st["flat_built"] > 0 and st["lean_escaped"] == 0 and st["tail_instances"] == B, (s.plan(), st)
    if engine in ("lean_escapes", "flat_escapes"):
        assert st["lean_launches"] >= 1 and st["lean_escaped"] > 0, st
        assert (st["flat_launches"] >= 1) == (engine == "flat_escapes")
    assert_end_to_end(fetch_end_to_end(s, residuals=True), out, prm, ztol=1e-9, what=engine)
    s.close()


def test_lean_rounds_with_iteration_quanta(talos, monkeypatch):
    """LOIKB_LEAN_QUANTA: the lean kernel in rounds of at most q iterations per instance (decade slots are indexed by the
    instance's slot, the lists shrink from round to round) -- same answers as one launch"""
    link = talos.getJointId("arm_left_7_joint")
    B = 300
    wl = feasible_batch(talos, B, link, 17, nu_scale=0.5)
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    prm = dict(FIXTURE, max_iter=600, tol_abs=1e-6, tol_rel=0.0)
    monkeypatch.setenv("LOIKB_FLAT", "0")  # (a mechanism of k_lean)
    monkeypatch.delenv("LOIKB_LEAN_QUANTA", raising=False)
    a = loik_amd.BatchedLoik(talos, B, **prm)
    a.Solve(*args)
    monkeypatch.setenv("LOIKB_LEAN_QUANTA", "5,11,40")
    b = loik_amd.BatchedLoik(talos, B, **prm)
    b.Solve(*args)
    assert b.stats()["lean_launches"] >= 3 and a.stats()["lean_launches"] == 1
    for name in ["iter", "converged", "primal_infeasible", "mu"]:
        assert np.array_equal(a.get(name), b.get(name)), name
    for name in ["z", "nu", "w", "vis", "fis", "primal_residual", "dual_residual", "delta_vis_inf_norm", "g_inf_norm"]:
        assert np.max(np.abs(a.get(name) - b.get(name))) < 1e-10, name
    a.close(); b.close()


def test_lean_time_slicing_changes_nothing(talos, monkeypatch):
    """LOIKB_LEAN_SLICE=q: round-robin time slicing inside the lean launch (k_lean<.., SLICED = true>): an instance that used
    q iterations while others wait for a slot is written back and re-queued -- it migrates between lane groups, i.e. between
    CUs of different XCDs, within one launch (coherent record accesses).  Per-instance results are bit-identical."""
    link = talos.getJointId("arm_left_7_joint")
    B = 12000  # more instances than resident lane groups (4096): the queue is never empty at the first slice boundaries
    wl = feasible_batch(talos, B, link, 23, nu_scale=0.5)
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
    monkeypatch.setenv("LOIKB_FLAT", "0")  # (a mechanism of k_lean)
    monkeypatch.delenv("LOIKB_LEAN_SLICE", raising=False)
    a = loik_amd.BatchedLoik(talos, B, **prm)
    a.Solve(*args)
    assert a.stats()["lean_requeues"] == 0
    for q in ("7", "40"):
        monkeypatch.setenv("LOIKB_LEAN_SLICE", q)
        b = loik_amd.BatchedLoik(talos, B, **prm)
        for _ in range(2):
            b.Solve(*args)
            st = b.stats()
            assert st["lean_requeues"] > 0 and st["lean_escaped"] == 0
            assert st["instance_iterations"] == a.stats()["instance_iterations"]
            for name in ["iter", "converged", "primal_infeasible", "mu", "z", "nu", "w", "vis", "fis", "g", "yis", "primal_residual",
                         "dual_residual", "delta_vis_inf_norm", "g_inf_norm", "mu_updates"]:
                assert np.array_equal(a.get(name), b.get(name)), (q, name)
        b.close()
    a.close()


@pytest.mark.parametrize("robot", ["talos32", "talos44", "talos32_lean"])
def test_longest_first_order_changes_nothing_but_the_schedule(robot, monkeypatch):
    """the flat engine takes a handle's second and later solves longest first (counting sort of the previous solve's iteration
    counts, k_order_*): the schedule changes, no number does -- every member bit-identical to the first solve's (arrival order,
    time-sliced) and to a handle with LOIKB_FLAT_ORDER=0; a different problem on the same handle (stale order) as well"""
    from loik_amd import workloads
    for v in ("LOIKB_FLAT_ORDER", "LOIKB_FLAT_SLICE", "LOIKB_LEAN_WG_PER_CU"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("LOIKB_LEAN_WG_PER_CU", "1")     # (few resident wavefronts: the queue matters at a test-sized batch)
    monkeypatch.setenv("LOIKB_FLAT_ORDER_HOLDOFF", "0")  # (no fall-back to arrival order on a timing comparison: the test wants the ordered launches)
    lean = robot == "talos32_lean"     # (k_lean takes its list in the same order: LOIKB_FLAT=0)
    if lean:
        monkeypatch.setenv("LOIKB_FLAT", "0")
    B = 6000
    wl = workloads.talos_wholebody(B, seed=5) if robot == "talos44" else workloads.talos_c3(B, seed=5)
    wl2 = workloads.talos_wholebody(B, seed=6) if robot == "talos44" else workloads.talos_c3(B, seed=6)
    names = ["iter", "converged", "primal_infeasible", "mu", "z", "nu", "w", "vis", "fis", "g", "yis", "primal_residual", "dual_residual",
             "mu_updates"]

    def run(s, w):
        s.SolveInit(w["q"], w["H_ref"], w["v_ref"], w["c_ids"], w["Ais"], w["bis"], w["lb"], w["ub"])
        s.Solve()
        return {n: s.get(n) for n in names}, s.stats()
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    first, st1 = run(s, wl)
    assert st1["flat_launches"] == (0 if lean else 1) and st1["lean_launches"] == 1 and st1["flat_ordered"] == 0, st1
    s.Solve()
    st2 = s.stats()
    assert st2["flat_ordered"] == 1 and st2["lean_requeues"] == 0 and st2["tail_instances"] == B, st2
    for n in names:
        assert np.array_equal(s.get(n), first[n]), n
    other, st3 = run(s, wl2)                           # another batch on the handle: the counts were taken on other inputs,
    assert st3["flat_ordered"] == 0, st3               # the stale order is NOT used (VERDICT r03 #2b) ...
    s.Solve()
    assert s.stats()["flat_ordered"] == 1              # ... and this batch's own counts are, from its second solve on
    for n in names:
        assert np.array_equal(s.get(n), other[n]), n
    # a caller whose consecutive problems resemble each other may ask for the previous solve's order across a change of inputs
    # (LOIKB_OPT_ORDER_FROM_PREVIOUS): only a schedule -- same bits
    t = loik_amd.BatchedLoik(wl["model"], B, flags=loik_amd.capi.OPT_ORDER_FROM_PREVIOUS, **wl["params"])
    run(t, wl)
    stale, st4 = run(t, wl2)
    assert st4["flat_ordered"] == 1, st4
    # UpdateEqConstraint / a tailored solve / integrate / a setter between two solves invalidate the order as SolveInit does
    s.UpdateEqConstraint(int(wl2["c_ids"][0]), wl2["Ais"][0] if wl2["Ais"].ndim == 3 else wl2["Ais"], wl2["bis"][:, 0])
    s.Solve()
    assert s.stats()["flat_ordered"] == 0
    s.Solve()
    assert s.stats()["flat_ordered"] == 1
    s.set_tol(1e-5, 0.0)
    s.Solve()
    assert s.stats()["flat_ordered"] == 0
    monkeypatch.setenv("LOIKB_FLAT_ORDER", "0")
    p = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    plain, _ = run(p, wl2)
    p.Solve()
    assert p.stats()["flat_ordered"] == 0
    for n in names:
        assert np.array_equal(other[n], plain[n]), n
        assert np.array_equal(stale[n], plain[n]), n
    s.close(); p.close(); t.close()


def test_zero_state_launch_fetches_less_and_computes_the_same(monkeypatch):
    """straight after a cold reset the flat engine does not fetch vis, fis, g, w, z (zeros in every record, MODE_ZERO_STATE): same
    bits as with the fetch (LOIKB_FLAT_ZERO_STATE=0); Solve() again keeps nu / Stf_plus_w (ResetRecursion) and a warm-started
    tailored solve fetches everything -- both against the handle that always fetches"""
    from loik_amd import workloads
    for v in ("LOIKB_FLAT_ORDER", "LOIKB_FLAT_SLICE", "LOIKB_FLAT_ZERO_STATE"):
        monkeypatch.delenv(v, raising=False)
    B = 1500
    wl = workloads.talos_c3(B, seed=9)
    link = int(wl["c_ids"][0])
    names = ["iter", "converged", "mu", "z", "nu", "w", "vis", "fis", "g", "yis", "Stf_plus_w", "primal_residual", "dual_residual"]
    a = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    monkeypatch.setenv("LOIKB_FLAT_ZERO_STATE", "0")
    b = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    for s in (a, b):
        s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    for step in ("Solve()", "Solve() again", "warm-started tailored Solve"):
        for s in (a, b):
            if step.startswith("warm"):
                s.set_warm_start(True)
                s.Solve(None, link, wl["Ais"], 0.9 * wl["bis"][:, 0])
            else:
                s.Solve()
            assert s.stats()["flat_split_launches"] == 1
        for n in names:
            assert np.array_equal(a.get(n), b.get(n)), (step, n)
    a.close(); b.close()


def test_engine_plan_is_made_in_one_place(talos, panda7, monkeypatch):
    """loikb_plan_string: the dispatch (nb, nc, A shared?, children, options) -> engines, re-made at SolveInit when the sharing
    mode of A is known (round 1 fixed the chunk count at create with the default mode)"""
    for v in ("LOIKB_LEAN", "LOIKB_FLAT", "LOIKB_CHUNKS", "LOIKB_LEAN_SLICE"):
        monkeypatch.delenv(v, raising=False)
    s = loik_amd.BatchedLoik(talos, 256, **FIXTURE)
    assert "k_flat2" in s.plan() and "1 chunk" in s.plan() and "any reference cost" in s.plan(), s.plan()
    s.close()
    s = loik_amd.BatchedLoik(talos, 256, logging=True, **FIXTURE)   # (logging handles: the same engine, its LOG build -- round 4)
    assert "k_flat2" in s.plan() and "any reference cost" in s.plan() and "logging = 1" in s.plan(), s.plan()
    s.close()
    monkeypatch.setenv("LOIKB_FLAT", "0")
    s = loik_amd.BatchedLoik(talos, 40000, **dict(FIXTURE, num_eq_c=2))
    assert "k_lean" in s.plan() and "1 chunk" in s.plan() and "8 wavefronts per CU" in s.plan() and "LOIKB_FLAT=0" in s.plan()
    # per-instance A with two constraints: larger constraint blocks in LDS -> the plan is re-made at SolveInit: seven
    # single-wavefront workgroups per CU instead of two 4-wavefront ones
    link = [talos.getJointId("arm_left_7_joint"), talos.getJointId("arm_right_7_joint")]
    from helpers import multi_task_batch
    wl = multi_task_batch(talos, 40000, link, 3, per_instance_A=True)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    assert "7 wavefronts per CU in workgroups of 1" in s.plan(), s.plan()
    s.Solve()
    st = s.stats()
    assert st["chunks"] == 1 and st["lean_launches"] == 1 and st["tail_instances"] == 40000
    s.close()
    s = loik_amd.BatchedLoik(talos, 64, mu_update_strat=1, **{k: v for k, v in FIXTURE.items() if k != "mu_update_strat"})
    assert "OSQP" in s.plan()
    s.close()
    monkeypatch.setenv("LOIKB_LEAN", "0")
    s = loik_amd.BatchedLoik(panda7, 64, **FIXTURE)
    assert "LOIKB_LEAN=0" in s.plan() and "1 chunk" in s.plan()
    s.close()
    s = loik_amd.BatchedLoik(talos, 40000, **FIXTURE)  # without the lean kernel a large batch is solved as two chunks
    assert "no k_lean" in s.plan() and "2 chunk" in s.plan(), s.plan()
    s.close()


@pytest.mark.parametrize("engine", ["lean", "tail", "solve"])
def test_joints_not_numbered_depth_first(talos, engine, monkeypatch):
    """Pinocchio only guarantees parents[i] < i; a model assembled with addJoint need not be depth-first (round 1 refused
    such trees).  Talos renumbered level by level and a random tree: every engine against the oracle on the SAME numbering,
    and the answer is the depth-first model's answer, permuted."""
    from helpers import renumber_breadth_first
    for base in (talos, random_tree(9, 28, branch_prob=0.45, all_types=False)):
        bfs, order = renumber_breadth_first(base)
        assert not np.array_equal(bfs.parents, base.parents)
        link_old = base.getJointId("arm_left_7_joint") if base is talos else base.njoints - 1
        link = int(np.flatnonzero(order == link_old)[0])
        B = 200
        wl0 = feasible_batch(base, B, link_old, 61, nu_scale=0.5)
        # the same problems in the new numbering (1-DoF joints: q / bounds / nu are permuted with the joints)
        perm = order[1:] - 1
        wl = dict(wl0, q=wl0["q"][:, perm], c_ids=np.array([link], dtype=np.int32), lb=wl0["lb"][perm], ub=wl0["ub"][perm])
        args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
        out = ref.solve_batch(bfs, *args, nthreads=4, want_nu=True, **prm)
        out0 = ref.solve_batch(base, wl0["q"], wl0["H_ref"], wl0["v_ref"], wl0["c_ids"], wl0["Ais"], wl0["bis"], wl0["lb"], wl0["ub"],
                               nthreads=4, **prm)
        same = out["iters"] == out0["iters"]
        assert same.mean() > 0.97 and np.abs(out["z"] - out0["z"][:, perm])[same].max() < 1e-9  # (oracle: numbering-invariant)
        s = _solver(bfs, B, prm, engine, monkeypatch)
        s.Solve(*args)
        assert_end_to_end(fetch_end_to_end(s, residuals=True), out, prm, ztol=1e-9, what="bfs numbering, " + engine)
        for k in (2,):
            prk = dict(prm, max_iter=k + 1, tol_abs=0.0, tol_primal_inf=0.0)
            sk = _solver(bfs, B, prk, engine, monkeypatch)
            sk.Solve(*args)
            vis, fis, His = sk.get("vis"), sk.get("fis"), sk.His_full()
            for b in range(0, B, 41):
                r = ref.RefSolver(bfs, **prk)
                r.Solve(*problem_args(wl, b))
                assert_close(vis[b], r.vis[1:], 1e-9, "vis"); assert_close(fis[b], r.fis[1:], 1e-9, "fis")
                assert_close(His[b], r.His[1:], 1e-9, "His")
            sk.close()
        s.close()


@pytest.mark.gpu
def test_more_joints_than_lanes_of_a_wavefront():
    """a robot with more than 64 joints (a humanoid with hands): the one-joint-per-lane engines do not apply, k_solve runs the
    whole solve; every instance against the oracle"""
    from helpers import assert_end_to_end, fetch_end_to_end, multi_task_batch
    nb = 80
    model = random_tree(3, nb, branch_prob=0.3)
    B = 200
    wl = multi_task_batch(model, B, [nb // 3, nb], 7, bound=0.5, nu_scale=0.3)
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0, num_eq_c=2)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                          nthreads=8, want_nu=True, **prm)
    s = loik_amd.BatchedLoik(model, B, **prm)
    assert "more joints than lanes" in s.plan()
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=1e-9, off_ztol=1e-5, what="80 joints")
    st = s.stats()
    assert st["tail_instances"] == 0 and st["lean_launches"] == 0
    s.close()


@pytest.mark.gpu
def test_two_handles_on_their_own_streams_run_concurrently_and_agree(talos):
    """LOIKB_OPT_OWN_STREAM: two handles, two host threads, one device -- the launches interleave on the GPU; each handle's
    results are those of the same solve run alone"""
    import threading
    from loik_amd import capi
    link = talos.getJointId("arm_left_7_joint")
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0)
    wls = [feasible_batch(talos, 4096, link, 11 + k) for k in range(2)]
    alone = []
    for wl in wls:
        s = loik_amd.BatchedLoik(talos, 4096, **prm)
        s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        alone.append((s.get("z"), s.get("iter"), s.get("converged")))
        s.close()
    solvers = [loik_amd.BatchedLoik(talos, 4096, flags=capi.OPT_OWN_STREAM, **prm) for _ in wls]
    got, errs = [None, None], []

    def work(k):
        try:
            s, wl = solvers[k], wls[k]
            for _ in range(3):
                s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
            got[k] = (s.get("z"), s.get("iter"), s.get("converged"))
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for k in range(2):
        assert np.array_equal(got[k][1], alone[k][1]) and np.array_equal(got[k][2], alone[k][2])
        assert np.array_equal(got[k][0], alone[k][0])      # bit-identical: instances never interact
    for s in solvers:
        s.close()


@pytest.mark.gpu
def test_decade_table_follows_the_handles_history(talos):
    """after the first solve of a handle the lean engine builds decade slots only around the decades its instances were seen in;
    a later solve that needs more escapes to k_tail for those instances (correct answers) and gets the full table back"""
    link = talos.getJointId("arm_left_7_joint")
    easy = feasible_batch(talos, 2048, link, 3, bound=2.0, nu_scale=0.1)     # converges in a few iterations at mu0
    hard = feasible_batch(talos, 2048, link, 4, bound=0.5, nu_scale=0.4)     # mu moves over several decades
    prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
    s = loik_amd.BatchedLoik(talos, 2048, **prm)
    for wl in (easy, easy, hard, hard, easy):
        args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        s.Solve(*args)
        out = ref.solve_batch(talos, *args, nthreads=8, want_nu=True, **prm)
        assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=1e-9, off_ztol=1e-5, what="history")
        assert s.stats()["lean_launches"] > 0
    s.close()


def _fuzz():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_engines", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                              "scripts", "fuzz_engines.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fuzz_slice_every_engine(monkeypatch):
    """a bounded slice of scripts/fuzz_engines.py inside the suite: random trees (1-DoF / multi-DoF / composite joints, depth- and
    breadth-first numbering), 0..4 constraints, shared / per-instance data, reference costs, tolerances, penalty rules, every engine
    configuration -- against the oracle, no instance dropped.  (The long runs live in profiles/r03_*_fuzz_summary.txt.)"""
    for k in ("LOIKB_LEAN", "LOIKB_FLAT", "LOIKB_FLAT_SPLIT", "LOIKB_FLAT_SLICE", "LOIKB_LEAN_KLO", "LOIKB_LEAN_DECADES", "LOIKB_LEAN_SLICE"):
        monkeypatch.delenv(k, raising=False)   # (the fuzzer sets and clears them itself; restored after the test)
    out = _fuzz().fuzz(40, 31337, verbose=False, max_batch=700)
    assert out["cases"] + out["refused"] == 40 and out["instances"] > 5000, out
    assert out["mismatches"] == 0, out
    # (a case whose only differences are instances that max_iter stopped unconverged is named, not waved through)
    for c in out["unconverged_cases"]:
        print("unconverged-only:", c)
    assert out["unconverged_only"] == 0, out["unconverged_cases"]


def test_fuzz_slice_flat_engine(monkeypatch):
    """the same, drawn inside the flat engine's domain (> 16 joints numbered depth-first, H_ref = h I with or without a target,
    DEFAULT penalty rule; default plan, hand-over from k_solve, forced escapes, two stages)"""
    for k in ("LOIKB_LEAN", "LOIKB_FLAT", "LOIKB_FLAT_SPLIT", "LOIKB_FLAT_SLICE", "LOIKB_LEAN_KLO", "LOIKB_LEAN_DECADES", "LOIKB_LEAN_SLICE"):
        monkeypatch.delenv(k, raising=False)
    out = _fuzz().fuzz(30, 4242, verbose=False, max_batch=700, flat_bias=1.0)
    for c in out["unconverged_cases"]:
        print("unconverged-only:", c)
    assert out["mismatches"] == 0 and out["unconverged_only"] == 0, (out["unconverged_cases"], out)
    assert out["flat_cases"] >= 15, out   # (batches below 64 instances and trees the schedule refuses run elsewhere)


def test_flat_engine_builds_the_decades_its_table_lacks(talos, monkeypatch):
    """Round 5: with LOIKB_FLAT_BUILD=1 an instance whose mu leaves the decades k_fslots built for the launch builds the
    missing slot in-wave (flat_build_slot: k_fslots' two passes for one mu, the same operations in the same order) and carries on -- no
    hand-over to k_tail.  The columns are k_fslots' bit for bit, so whatever the table holds the results are THE SAME BITS: the full
    table, a table of one decade (everything else built in-wave), and one decade with time slices (parked instances rebuild after a
    switch) are compared on 3000 headline instances."""
    from loik_amd import workloads
    B = 3000
    wl = workloads.talos_c3(B, seed=321)
    prm = dict(wl["params"], max_iter=400)
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    names = ("LOIKB_LEAN_KLO", "LOIKB_LEAN_DECADES", "LOIKB_LEAN_ADAPT", "LOIKB_FLAT_BUILD", "LOIKB_FLAT_SLICE", "LOIKB_LEAN_WG_PER_CU", "LOIKB_FLAT_WINDOW")
    # (LOIKB_FLAT_WINDOW=lo,n: the table keeps its whole range and k_fslots populates decades lo .. lo + n - 1 of it; what an instance
    #  builds in-wave goes into the table too -- its later visits of the decade load it --, also from a wavefront of another XCD)
    cases = (("table", dict()),
             ("window_of_one", dict(LOIKB_FLAT_BUILD="1", LOIKB_FLAT_WINDOW="0,1")),
             ("window_of_one_sliced", dict(LOIKB_FLAT_BUILD="1", LOIKB_FLAT_WINDOW="1,1", LOIKB_FLAT_SLICE="7", LOIKB_LEAN_WG_PER_CU="1")),
             ("one_decade", dict(LOIKB_LEAN_KLO="0", LOIKB_LEAN_DECADES="1", LOIKB_LEAN_ADAPT="0", LOIKB_FLAT_BUILD="1")),
             ("no_decade_of_use", dict(LOIKB_LEAN_KLO="-2", LOIKB_LEAN_DECADES="1", LOIKB_LEAN_ADAPT="0", LOIKB_FLAT_BUILD="1")),
             ("one_decade_sliced", dict(LOIKB_LEAN_KLO="1", LOIKB_LEAN_DECADES="1", LOIKB_LEAN_ADAPT="0", LOIKB_FLAT_BUILD="1", LOIKB_FLAT_SLICE="7",
                                        LOIKB_LEAN_WG_PER_CU="1")))
    res = {}
    for name, env in cases:
        for k in names:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = loik_amd.BatchedLoik(talos, B, **prm)
        s.Solve(*args)
        st = s.stats()
        assert st["flat_split_launches"] >= 1 and st["tail_instances"] == B and st["lean_escaped"] == 0, (name, st)
        assert (st["flat_built"] > B // 2) == (name != "table"), (name, st["flat_built"])
        if name != "table":
            assert "in-wave" in s.plan(), s.plan()
        if name.startswith("window"):   # (every (instance, decade) pair is built at most once: the table remembers)
            assert st["flat_built"] < 4 * B, (name, st["flat_built"])
        res[name] = {k: s.get(k) for k in ("iter", "converged", "primal_infeasible", "z", "nu", "mu", "yis", "fis", "vis", "w")}
        s.close()
    for name, _ in cases[1:]:
        for k in res["table"]:
            assert np.array_equal(res["table"][k], res[name][k]), (name, k, np.abs(res["table"][k] - res[name][k]).max())
    out = ref.solve_batch(talos, *args, nthreads=8, want_nu=True, **prm)
    assert (res["one_decade"]["iter"] == out["iters"]).mean() > 0.99


def test_flat_lazy_table_window_from_the_handles_history(talos, monkeypatch):
    """LOIKB_FLAT_BUILD=2: a time-sliced launch (32 768+ instances in arrival order) of a handle with a history lets k_fslots build only the
    decades 97 % of the previous solve's instances ended within; whoever goes further builds its slot in-wave, once, into the table.
    Same bits as the full table; fewer decades built (the slot kernel's share of the launch shrinks)."""
    from loik_amd import workloads
    B = 32768
    wa, wb = workloads.talos_c3(B, seed=11), workloads.talos_c3(B, seed=12)
    args = lambda w: (w["q"], w["H_ref"], w["v_ref"], w["c_ids"], w["Ais"], w["bis"], w["lb"], w["ub"])
    res, slots = {}, {}
    for mode in ("0", "2"):
        monkeypatch.setenv("LOIKB_FLAT_BUILD", mode)
        s = loik_amd.BatchedLoik(talos, B, **wa["params"])
        s.Solve(*args(wa))                       # the history
        s.Solve(*args(wb))                       # a new batch: arrival order, sliced
        st = s.stats()
        assert st["flat_split_launches"] == 1 and st["flat_ordered"] == 0 and st["lean_requeues"] > 0 and st["lean_escaped"] == 0, st
        assert (st["flat_built"] > 0) == (mode == "2"), st
        res[mode] = {k: s.get(k) for k in ("iter", "converged", "primal_infeasible", "z", "nu", "mu", "yis")}
        slots[mode] = st["hslots_ms"]
        s.close()
    for k in res["0"]:
        assert np.array_equal(res["0"][k], res["2"][k]), k
    assert slots["2"] < 0.8 * slots["0"], slots


def test_flat_lazy_table_default_window_of_a_first_solve(talos, monkeypatch):
    """The default (LOIKB_FLAT_BUILD unset = 2): a handle's FIRST time-sliced solve of 49 152+ instances lets k_fslots build the five decades from
    mu0's upwards only; whoever leaves them builds its slot in-wave.  Same bits as the full table (LOIKB_FLAT_BUILD=0), a shorter slot kernel."""
    from loik_amd import workloads
    B = 65536
    wl = workloads.talos_c3(B, seed=5)
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    res, slots = {}, {}
    for mode in ("0", None):
        monkeypatch.delenv("LOIKB_FLAT_BUILD", raising=False)
        monkeypatch.delenv("LOIKB_FLAT_WINDOW", raising=False)
        if mode is not None:
            monkeypatch.setenv("LOIKB_FLAT_BUILD", mode)
        s = loik_amd.BatchedLoik(talos, B, **wl["params"])
        s.Solve(*args)
        st = s.stats()
        assert st["flat_split_launches"] == 1 and st["flat_ordered"] == 0 and st["lean_requeues"] > 0 and st["lean_escaped"] == 0, st
        if mode == "0":
            assert st["flat_built"] == 0, st
        res[mode] = {k: s.get(k) for k in ("iter", "converged", "primal_infeasible", "z", "nu", "mu", "yis", "fis")}
        slots[mode] = st["hslots_ms"]
        s.close()
    for k in res["0"]:
        assert np.array_equal(res["0"][k], res[None][k]), k
    assert slots[None] < 0.8 * slots["0"], slots


def test_flat_time_slicing_changes_nothing(talos, monkeypatch):
    """k_flat2's round-robin time slicing (LOIKB_FLAT_SLICE; on by default for arrival-order launches of 32 768..262 144
    instances, slices of 288 then 96 iterations): an instance whose slice is used up while others wait is parked and resumed later,
    possibly by a wavefront of another XCD (agent-scope accesses of the park record).  Same arithmetic, so bit-identical results; forced here with a
    slice of 5 iterations and one wavefront per CU, 1500 instances: thousands of requeues."""
    from loik_amd import workloads
    B = 1500
    wl = workloads.talos_c3(B, seed=123)
    prm = dict(wl["params"], max_iter=400)
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    res = {}
    for name, env in (("plain", dict(LOIKB_FLAT_SLICE="0")), ("sliced", dict(LOIKB_FLAT_SLICE="5", LOIKB_LEAN_WG_PER_CU="1"))):
        for k in ("LOIKB_FLAT_SLICE", "LOIKB_LEAN_WG_PER_CU"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = loik_amd.BatchedLoik(talos, B, **prm)
        s.Solve(*args)
        st = s.stats()
        assert st["flat_split_launches"] >= 1 and st["tail_instances"] == B
        assert (st["lean_requeues"] > 1000) == (name == "sliced"), st
        res[name] = {k: s.get(k) for k in ("iter", "converged", "primal_infeasible", "z", "nu", "mu", "yis", "fis", "vis")}
        assert st["instance_iterations"] == int(res[name]["iter"].sum())
        s.close()
    for k in res["plain"]:
        assert np.array_equal(res["plain"][k], res["sliced"][k]), k
    # the default: no slices below 32 768 instances (one straggler chain whatever the order), none when the variable says 0
    for k in ("LOIKB_FLAT_SLICE", "LOIKB_LEAN_WG_PER_CU"):
        monkeypatch.delenv(k, raising=False)
    s = loik_amd.BatchedLoik(talos, B, **prm)
    s.Solve(*args)
    assert s.stats()["lean_requeues"] == 0
    for k in res["plain"]:
        assert np.array_equal(res["plain"][k], s.get(k)), k
    s.close()


@pytest.mark.parametrize("robot", ["talos32", "talos44"])
@pytest.mark.parametrize("sliced", [False, True])
@pytest.mark.parametrize("weight", ["diagonal", "general"])
def test_flat_engine_with_a_diagonal_reference_weight(robot, sliced, weight, monkeypatch):
    """A reference weight shared by the links that is not h I -- diag(d_1 .. d_6) (other weights on the angular than on the linear
    velocity) or a general symmetric 6x6 -- stays on the flat engine: the HM = 1 / 2 instantiations of k_flat2 (talos32) and
    k_flat1 (talos44) sum the weighted link velocities over the subtrees beside the velocities themselves.  k iterations field by
    field and end to end against the oracle; with the time slicing forced."""
    from loik_amd import workloads
    from helpers import problem_args
    model = loik_amd.builtin_model(robot)
    B = 700
    wl = workloads.talos_wholebody(B, seed=9, model=model) if robot == "talos44" else workloads.talos_c3(B, seed=9)
    Href = np.diag([0.4, 1.5, 0.7, 3.0, 0.2, 2.2])
    if weight == "general":
        Q = np.linalg.qr(np.random.default_rng(4).normal(size=(6, 6)))[0]
        Href = Q @ Href @ Q.T
        Href = 0.5 * (Href + Href.T)
    vref = np.array([0.02, -0.01, 0.03, 0.05, -0.04, 0.01])   # H_ref v_ref != 0: the reference term's subtree sums too
    args = (wl["q"], Href, vref, wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    for k in ("LOIKB_FLAT_SLICE", "LOIKB_LEAN_WG_PER_CU"):
        monkeypatch.delenv(k, raising=False)
    if sliced:
        monkeypatch.setenv("LOIKB_FLAT_SLICE", "4"); monkeypatch.setenv("LOIKB_LEAN_WG_PER_CU", "1")
    for k in (1, 3, 9):
        prm = dict(wl["params"], max_iter=k + 1, tol_abs=0.0, tol_primal_inf=0.0)
        s = loik_amd.BatchedLoik(model, B, **prm)
        s.Solve(*args)
        st = s.stats()
        assert st["flat_launches"] >= 1 and st["tail_instances"] == B, (s.plan(), st)
        got = {n: s.get(n) for n in FIELDS + SCALARS}
        for b in range(0, B, 97):
            r = ref.RefSolver(model, **prm)
            r.Solve(wl["q"][b], Href, vref, wl["c_ids"], wl["Ais"], wl["bis"][b], wl["lb"], wl["ub"])
            for n in FIELDS:
                want = r.field(n)
                if n in ("vis", "fis", "g"):
                    want = want[1:]
                assert_close(got[n][b], want, 1e-9, "%s b%d k%d" % (n, b, k))
            for n in SCALARS:
                assert_close(got[n][b], r.scalar(n), 1e-9, "%s b%d k%d" % (n, b, k))
        s.close()
    prm = dict(wl["params"], max_iter=400)
    out = ref.solve_batch(model, *args, nthreads=8, want_nu=True, **prm)
    s = loik_amd.BatchedLoik(model, B, **prm)
    s.Solve(*args)
    st = s.stats()
    assert st["flat_launches"] >= 1 and st["lean_escaped"] == 0 and st["tail_instances"] == B, (s.plan(), st)
    assert (st["lean_requeues"] > 0) == sliced, st
    assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=1e-9, what="%s H_ref, %s" % (weight, robot))
    s.close()


@pytest.mark.parametrize("weight", ["scalar", "general"])
def test_whole_body_osqp_rule_on_the_flat_engine(weight, monkeypatch):
    """OSQP's penalty rule on the 44-joint tree: k_flat1<.., MUR = 1> (one lane per joint) builds W / Dinv of its instance in-wave at every
    change of mu, as k_flat2 does for 17..32 joints (tests/test_gpu_parity.py::test_osqp_mu_rule_matches_oracle) -- mu0's slot from the
    table, no time slices.  k iterations field by field incl. mu, and end to end against the oracle."""
    from loik_amd import workloads
    model = loik_amd.builtin_model("talos44")
    B = 600
    wl = workloads.talos_wholebody(B, seed=13, model=model)
    Href = wl["H_ref"]
    if weight == "general":
        Q = np.linalg.qr(np.random.default_rng(6).normal(size=(6, 6)))[0]
        Href = Q @ np.diag([0.4, 1.5, 0.7, 3.0, 0.2, 2.2]) @ Q.T
        Href = 0.5 * (Href + Href.T)
    args = (wl["q"], Href, wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    for k in ("LOIKB_FLAT_SLICE", "LOIKB_LEAN_WG_PER_CU"):
        monkeypatch.delenv(k, raising=False)
    for k in (3, 9):   # (mu follows the residual ratio: after 30 iterations the roundings of the two summation orders are at 7e-8 in mu itself)
        prm = dict(wl["params"], max_iter=k + 1, tol_abs=0.0, tol_primal_inf=0.0, mu_update_strat=1)
        s = loik_amd.BatchedLoik(model, B, **prm)
        s.Solve(*args)
        st = s.stats()
        assert "k_flat1" in s.plan() and "OSQP" in s.plan(), s.plan()
        assert st["flat_launches"] >= 1 and st["tail_instances"] == B and st["lean_requeues"] == 0, (s.plan(), st)
        assert st["flat_built"] > 0, st   # (the rule moves mu from the second iteration on)
        got = {n: s.get(n) for n in FIELDS + SCALARS}
        for b in range(0, B, 97):
            r = ref.RefSolver(model, **prm)
            r.Solve(wl["q"][b], Href, wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"][b], wl["lb"], wl["ub"])
            for n in FIELDS:
                want = r.field(n)
                if n in ("vis", "fis", "g"):
                    want = want[1:]
                assert_close(got[n][b], want, 1e-8, "%s b%d k%d" % (n, b, k))
            for n in SCALARS:   # (mu among them)
                assert_close(got[n][b], r.scalar(n), 1e-8, "%s b%d k%d" % (n, b, k))
        s.close()
    prm = dict(wl["params"], max_iter=500, mu_update_strat=1)
    out = ref.solve_batch(model, *args, nthreads=8, want_nu=True, **prm)
    s = loik_amd.BatchedLoik(model, B, **prm)
    s.Solve(*args)
    st = s.stats()
    assert st["flat_launches"] >= 1 and st["lean_escaped"] == 0 and st["tail_instances"] == B and st["flat_built"] > B, (s.plan(), st)
    # (off the oracle's iteration count: the fuzz's budget for this rule -- an instance that needs 340 / 450 iterations under a penalty that
    #  follows the residual ratio stops with z known to ~10 x the residual tolerance, here 7.9e-6 at tol 1e-6, flags equal)
    assert_end_to_end(fetch_end_to_end(s, residuals=True), out, prm, same_frac=0.99, ztol=5e-9, off_ztol=1e-5, res_tol=(1e-8, 1e-6),
                      what="OSQP rule, whole body, %s H_ref" % weight)
    mu = s.get("mu")
    assert np.unique(np.round(np.log10(mu), 9)).size > 12
    s.close()


def test_flat_probe_and_finish_in_two_launches_changes_nothing(talos, monkeypatch):
    """Round 6: a time-sliced k_flat2 launch without an order runs as TWO launches -- the probe (every instance for LOIKB_FLAT_PROBE
    iterations at most, the residual noted at three marks, survivors parked), k_probe_sort (survivors ordered by the iterations they
    are predicted to need still), and the launch that takes that list front to back, every instance to completion.  A parked instance
    continues exactly where it stopped, so every result is bit-identical to the single unsliced launch; forced here on 1500 instances
    with short probes (5, 40 iterations) and with longer ones, with and without the lazily populated table."""
    from loik_amd import workloads
    B = 1500
    wl = workloads.talos_c3(B, seed=321)
    prm = dict(wl["params"], max_iter=400)
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    keys = ("LOIKB_FLAT_SLICE", "LOIKB_FLAT_PROBE", "LOIKB_FLAT_PROBE_MARK", "LOIKB_FLAT_BUILD", "LOIKB_FLAT_WINDOW")
    res = {}
    cases = (("plain", dict(LOIKB_FLAT_SLICE="0")),
             ("probe5", dict(LOIKB_FLAT_SLICE="288", LOIKB_FLAT_PROBE="5", LOIKB_FLAT_PROBE_MARK="2")),
             ("probe40", dict(LOIKB_FLAT_SLICE="288", LOIKB_FLAT_PROBE="40")),
             ("probe128_full_table", dict(LOIKB_FLAT_SLICE="288", LOIKB_FLAT_PROBE="128", LOIKB_FLAT_BUILD="0")),
             ("probe320", dict(LOIKB_FLAT_SLICE="288", LOIKB_FLAT_PROBE="320")),
             ("probe40_window", dict(LOIKB_FLAT_SLICE="288", LOIKB_FLAT_PROBE="40", LOIKB_FLAT_BUILD="1", LOIKB_FLAT_WINDOW="0,2")))
    for name, env in cases:
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = loik_amd.BatchedLoik(talos, B, **prm)
        s.Solve(*args)
        st = s.stats()
        assert st["flat_split_launches"] >= 1 and st["tail_instances"] == B
        assert st["flat_probe_launches"] == (0 if name == "plain" else 1), (name, st)
        if name != "plain":
            assert st["lean_requeues"] > (100 if name != "probe320" else 10), (name, st)   # (the survivors were parked once each)
        if name == "probe40_window":
            assert st["flat_built"] > 0, st   # (decades outside the window: built in-wave, in either launch)
        res[name] = {k: s.get(k) for k in ("iter", "converged", "primal_infeasible", "z", "nu", "mu", "yis", "fis", "vis")}
        assert st["instance_iterations"] == int(res[name]["iter"].sum()), (name, st["instance_iterations"], int(res[name]["iter"].sum()))
        s.close()
    for name, _ in cases[1:]:
        for k in res["plain"]:
            assert np.array_equal(res["plain"][k], res[name][k]), (name, k)


@pytest.mark.parametrize("B", [1, 5, 63])
def test_small_batches_run_on_the_one_instance_per_wavefront_engine(talos, B, monkeypatch):
    """Round 6 (VERDICT r05 "missing" item 4): the reference's own call is ONE problem (tests/loik-loid.cpp:987-1032).  Batches below 64
    instances used to run on k_tail (10-12 us per iteration); k_flat2 takes them now (2.3 us), through the short sequence of a small batch
    (list + ring + counters from one kernel, no order pass, n_unfinished from the launch's own counters).  Same bits as the same
    instances inside a larger batch and as the long sequence; the oracle's iteration counts."""
    from loik_amd import workloads
    wl = workloads.talos_c3(128, seed=77)
    prm = dict(wl["params"], max_iter=300)
    sub = lambda a: a[:B] if getattr(a, "ndim", 0) >= 1 and a.shape[0] == 128 else a
    args_full = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    args = tuple(sub(np.asarray(a)) for a in args_full)
    keys = ("iter", "converged", "primal_infeasible", "z", "nu", "mu", "yis", "fis", "vis")
    res = {}
    for name, env, BB, aa in (("small", {}, B, args), ("long_sequence", {"LOIKB_FLAT_SMALL_BATCH": "0"}, B, args), ("in_a_batch_of_128", {}, 128, args_full)):
        for k in ("LOIKB_FLAT_SMALL_BATCH", "LOIKB_FLAT_MIN_BATCH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = loik_amd.BatchedLoik(talos, BB, **prm)
        s.Solve(*aa)
        st = s.stats()
        assert st["flat_split_launches"] == 1 and st["tail_instances"] == BB and st["launches"] == 1, (name, st)
        res[name] = {k: np.asarray(s.get(k))[:B] for k in keys}
        conv, pinf = np.asarray(s.get("converged")).astype(bool), np.asarray(s.get("primal_infeasible")).astype(bool)
        assert st["n_unfinished"] == int((~conv & ~pinf).sum()), (name, st["n_unfinished"])
        s.Solve()   # (again on the same handle: the short sequence leaves no order behind, and needs none)
        assert s.stats()["flat_ordered"] == (1 if name == "long_sequence" else 0)   # (the long sequence takes its second solve longest first)
        for k in keys:
            assert np.array_equal(np.asarray(s.get(k))[:B], res[name][k]), (name, k)
        s.close()
    for name in ("long_sequence", "in_a_batch_of_128"):
        for k in keys:
            assert np.array_equal(res["small"][k], res[name][k]), (name, k)
    out = ref.solve_batch(talos, *args, nthreads=2, **prm)
    assert np.array_equal(res["small"]["iter"], out["iters"])
    assert np.abs(res["small"]["z"] - out["z"]).max() < 1e-9
    # k_tail, round 5's engine for such a batch, still agrees (another engine: to rounding)
    monkeypatch.setenv("LOIKB_FLAT_MIN_BATCH", "64")
    s = loik_amd.BatchedLoik(talos, B, **prm)
    s.Solve(*args)
    assert s.stats()["flat_launches"] == 0
    assert np.array_equal(np.asarray(s.get("iter")), out["iters"]) and np.abs(np.asarray(s.get("z")) - out["z"]).max() < 1e-9
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("robot,B", [("talos32", 1), ("talos32", 700), ("talos32", 2048), ("talos44", 3), ("talos44", 300)])
def test_short_sequence_ends_with_one_synchronisation_and_the_same_results(robot, B, monkeypatch):
    """Round 6, last session: a small batch's Solve() ends with k_small_finish (ONE workgroup: the list of the unfinished, the counts, and the
    launch's counters straight into the chunk's pinned host copy) and one synchronisation; a plain Solve() starts with k_reset_and_queue (the
    reset of ResetRecursion + ResetSolver, hpp:370-374, and the queue's set-up in one launch).  LOIKB_SMALL_FINISH=0 is the sequence as it
    was (k_reset, k_queue_init_iota, ..., k_list_unfinished, a copy, two synchronisations): same bits, same statistics -- over Solve(args),
    Solve() twice (the fused launch), another batch through Solve(args), Solve() again, and a warm-started handle."""
    from loik_amd import workloads
    make = workloads.talos_c3 if robot == "talos32" else workloads.talos_wholebody
    wa, wb = make(B, seed=91), make(B, seed=92)
    prm = dict(wa["params"], max_iter=250)
    args = lambda w: (w["q"], w["H_ref"], w["v_ref"], w["c_ids"], w["Ais"], w["bis"], w["lb"], w["ub"])
    keys = ("iter", "converged", "primal_infeasible", "z", "nu", "mu", "yis", "fis", "vis", "w")
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("LOIKB_SMALL_FINISH", mode)
        got = []
        for warm in (False, True):
            s = loik_amd.BatchedLoik(wa["model"], B, **dict(prm, warm_start=warm))
            for step in ("full_a", "plain", "plain", "full_b", "plain"):
                if step == "plain":
                    s.Solve()
                else:
                    s.Solve(*args(wa if step == "full_a" else wb))
                st = s.stats()
                assert st["flat_launches"] == 1 and st["launches"] == 1 and st["tail_instances"] == B, (mode, step, st)
                r = {k: np.asarray(s.get(k)) for k in keys}
                conv, pinf = r["converged"].astype(bool), r["primal_infeasible"].astype(bool)
                assert st["n_unfinished"] == int((~conv & ~pinf).sum()), (mode, step, st["n_unfinished"])
                assert st["instance_iterations"] == int(r["iter"].sum()), (mode, step)
                assert st["total_ms"] >= st["kernel_ms"] > 0.0, (mode, step, st)
                got.append(r)
            s.close()
        runs[mode] = got
    for a, b in zip(runs["0"], runs["1"]):
        for k in keys:
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("robot,ns_max", [("talos32", 6.0), ("talos44", 8.0)])
@pytest.mark.parametrize("sliced", [False, True])
def test_flat_kernels_iteration_rate_has_not_tipped_over(robot, ns_max, sliced, monkeypatch):
    """A guard, not a benchmark.  The flat kernels' register allocation sits close to a tipping point since their unit is scheduled by the
    iterative ILP scheduler: twice in round 6 a small source change made the compiler allocate scratch reloads INTO the iteration loop
    (13 -> 43 -> 123 per iteration; one build ran the whole body in 68 ms instead of 14), every parity test still green.  Here: 16 384
    instances (a batch whose launch is half straggler chain), wall time of the on-chip launch per instance-iteration -- 2.5-3.3 ns (Talos-32) /
    3.4-3.8 ns (whole body) on an idle MI355X, bounds at twice that (a loaded box, a cold clock), less than half of what the tipped build took.  scripts/r05/loopstat.py on the unit's assembly is
    the check that says WHY."""
    from loik_amd import workloads
    monkeypatch.setenv("LOIKB_FLAT_ORDER", "0")
    monkeypatch.setenv("LOIKB_FLAT_SLICE", "288" if sliced else "0")
    B = 16384
    wl = (workloads.talos_c3 if robot == "talos32" else workloads.talos_wholebody)(B)
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    best = 1e9
    for _ in range(3):
        s.Solve()
        st = s.stats()
        assert st["flat_launches"] == 1, st
        best = min(best, (st["tail_ms"] - st["hslots_ms"]) * 1e6 / st["instance_iterations"])
    s.close()
    assert best < ns_max, "%.2f ns per instance-iteration: has the kernel's loop acquired scratch reloads? (scripts/r05/loopstat.py)" % best


@pytest.mark.gpu
@pytest.mark.parametrize("robot,B,nc", [("talos32", 1, 1), ("talos32", 300, 1), ("talos44", 5, 4), ("talos32", 4096, 1)])
def test_results_in_one_call_are_the_getters_values(robot, B, nc):
    """loikb_get_results (round 6, last session): z, nu, w, vis, fis, yis of the reference's data object (loik-loid-data-optimized.hpp:118-178) in
    ONE call -- one gather launch into pinned host memory and one synchronisation while the batch's results fit 4 MiB, field by field above
    (B = 4096: 13 MB) --, any subset, bit for bit what six loikb_get calls return; also after a change of the constraint set (the row map of
    yis changes)."""
    from loik_amd import workloads
    wl = (workloads.talos_c3 if robot == "talos32" else workloads.talos_wholebody)(B, seed=55)
    prm = dict(wl["params"], max_iter=120)
    s = loik_amd.BatchedLoik(wl["model"], B, **prm)
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    assert len(wl["c_ids"]) == nc
    names = ("z", "nu", "w", "vis", "fis", "yis")
    one = {k: s.get(k) for k in names}
    allr = s.get_results()
    assert set(allr) == set(names)
    for k in names:
        assert allr[k].shape == one[k].shape and np.array_equal(allr[k], one[k]), k
    sub = s.get_results(("nu", "fis"))
    assert set(sub) == {"nu", "fis"} and np.array_equal(sub["nu"], one["nu"]) and np.array_equal(sub["fis"], one["fis"])
    # the scalar block (ABI 602): what get_iter() / get_convergence_status() / the residual getters read, from the same gather
    from loik_amd.capi import _SCALAR_FIELDS
    sc = s.get_results(("z", "scalars"))
    assert sc["scalars"].shape == (B, 33) and np.array_equal(sc["z"], one["z"])
    for k, name in enumerate(_SCALAR_FIELDS):
        assert np.array_equal(sc["scalars"][:, k], np.asarray(s.get(name))), name
    assert np.array_equal(sc["scalars"][:, s.SCALAR_ITER], np.asarray(s.get("iter")).astype(float))
    assert np.array_equal(sc["scalars"][:, s.SCALAR_STATUS], np.asarray(s.get("status")).astype(float))
    assert np.array_equal(sc["scalars"][:, s.SCALAR_MU_UPDATES], np.asarray(s.get("mu_updates")).astype(float))
    st_bits = sc["scalars"][:, s.SCALAR_STATUS].astype(int)
    assert np.array_equal((st_bits & 1) != 0, np.asarray(s.get("converged")).astype(bool))
    assert np.array_equal((st_bits & 2) != 0, np.asarray(s.get("primal_infeasible")).astype(bool))
    with pytest.raises(ValueError):
        s.get_results(("z", "His"))
    if nc > 1:   # one task less: yis has a row less per instance, the cached row map must follow
        s.RemoveEqConstraint(int(wl["c_ids"][-1]))
        s.Solve()
        r = s.get_results()
        assert r["yis"].shape == (B, nc - 1, 6)
        for k in names:
            assert np.array_equal(r[k], s.get(k)), k
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 40, 3000])
def test_a_handle_reused_for_other_problems_answers_as_a_fresh_one(talos, B):
    """SolveInit's small-batch machinery (round 6, last session): row maps cached on the device, uploads queued behind each other with ONE
    synchronisation, small host inputs read from a pinned buffer, the uniform inputs (shared A / bounds) sent from a host copy once per SolveInit,
    the joints' descriptors sent only when the task links change, q + FwdPassInit from one joint-parallel launch, the queue prepared by
    FwdPassInit's closing reset.  One handle taken through problems that differ in exactly those things -- another task link, A and the bounds shared
    or per instance, the full Solve(args), SolveInit + Solve(), the tailored Solve -- returns, bit for bit, what a fresh handle returns for each."""
    from helpers import feasible_batch
    la, lb_ = talos.getJointId("arm_left_7_joint"), talos.getJointId("leg_right_6_joint")
    probs = [("left wrist, shared A and bounds", feasible_batch(talos, B, la, 501, nu_scale=0.5)),
             ("right foot, shared", feasible_batch(talos, B, lb_, 502, nu_scale=0.5)),
             ("left wrist again, A per instance", feasible_batch(talos, B, la, 503, nu_scale=0.5, per_instance_A=True)),
             ("right foot, bounds per instance", feasible_batch(talos, B, lb_, 504, nu_scale=0.5, per_instance_bounds=True)),
             ("left wrist, shared again", feasible_batch(talos, B, la, 505, nu_scale=0.5))]
    prm = dict(FIXTURE, max_iter=150, tol_abs=1e-6, tol_rel=0.0)
    keys = ("iter", "converged", "primal_infeasible", "z", "nu", "w", "yis", "fis", "vis")
    args = lambda w: (w["q"], w["H_ref"], w["v_ref"], w["c_ids"], w["Ais"], w["bis"], w["lb"], w["ub"])
    one = loik_amd.BatchedLoik(talos, B, **prm)
    for k, (name, wl) in enumerate(probs):
        how = ("full", "init+solve", "full")[k % 3]
        if how == "full":
            one.Solve(*args(wl))
        else:
            one.SolveInit(*args(wl)); one.Solve()
        fresh = loik_amd.BatchedLoik(talos, B, **prm)
        fresh.Solve(*args(wl))
        for f in keys:
            assert np.array_equal(np.asarray(one.get(f)), np.asarray(fresh.get(f))), (name, how, f)
        r = one.get_results()
        for f in ("z", "nu", "w", "vis", "fis", "yis"):
            assert np.array_equal(r[f], np.asarray(fresh.get(f))), (name, f)
        # the tailored entry on the reused handle: a new q and target for the same link and A (hpp:596-695) against SolveInit + Solve of that problem
        wl2 = feasible_batch(talos, B, int(wl["c_ids"][0]), 600 + k, nu_scale=0.5)
        Ai = wl["Ais"][0] if np.asarray(wl["Ais"]).ndim == 3 else wl["Ais"][:, 0]
        one.Solve(wl2["q"], int(wl["c_ids"][0]), Ai, wl2["bis"][:, 0])
        fresh.Solve(wl2["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl2["bis"], wl["lb"], wl["ub"])
        for f in keys:
            assert np.array_equal(np.asarray(one.get(f)), np.asarray(fresh.get(f))), (name, "tailored", f)
        fresh.close()
    one.close()
