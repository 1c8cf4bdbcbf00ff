"""No refused robots ("missing" item 3 / task 5): every `parents[i] < i` tree solves.

The reference's visitors walk whatever pinocchio::Model they are given (/root/reference/include/loik/loik-loid-optimized.hxx:345-354,
:361-377: `for (JointIndex idx : joint_range)`).  k_solve, the streaming engine, keeps the leaf->root hand-over of a sweep in LDS slots, one
per pending branch, and a very bushy tree -- the fuzz's 42-joint trees with 8..10 children at one joint -- needs more of them than a CU has
LDS: such a model used to be refused (LOIKB_ERR_MODEL).  Now its batches go whole to the on-chip engines, which have no such slots, and
where those do not apply (more than 64 joints; options that ask for k_solve's own behaviour) to the plain pass-by-pass implementation
(k_pass_solve), the engine of last resort.  k = 1, 2, 5 iterations field by field and end to end against the oracle."""
import numpy as np
import pytest

import loik_amd
from helpers import FIXTURE, assert_close, assert_end_to_end, fetch_end_to_end, multi_task_batch, problem_args, random_rotation
from oracle import ref

FIELDS = ["nu", "z", "w", "vis", "fis", "g", "yis", "Aty", "Stf_plus_w"]
SCALARS = ["iter", "converged", "primal_infeasible", "primal_residual", "dual_residual", "mu", "tol_primal", "tol_dual"]


def bushy_tree(seed, nb, hub, hub_children, depth_first=True):
    """a tree of nb one-DoF joints in which joint `hub` has `hub_children` children, each the root of a chain / small subtree;
    numbered depth-first (subtrees contiguous) unless depth_first is False (children of the hub numbered first, then their subtrees)"""
    rng = np.random.default_rng(seed)
    parents = [0] + list(range(0, hub))          # a chain 1 .. hub
    rest = nb - hub
    sizes = np.full(hub_children, rest // hub_children); sizes[: rest % hub_children] += 1
    assert sizes.min() >= 1
    if depth_first:
        for sz in sizes:
            first = len(parents)
            parents.append(hub)
            for k in range(1, sz):
                parents.append(first + k - 1 if rng.random() < 0.7 else int(rng.integers(first, first + k)))
    else:
        firsts = []
        for sz in sizes:
            firsts.append(len(parents)); parents.append(hub)
        tails = list(firsts)
        for c, sz in enumerate(sizes):
            for k in range(1, sz):
                parents.append(tails[c]); tails[c] = len(parents) - 1
    assert len(parents) == nb + 1 and all(parents[i] < i for i in range(1, nb + 1))
    types, axis, placement = [0], [np.zeros(3)], [np.concatenate([np.eye(3).ravel(), np.zeros(3)])]
    for i in range(1, nb + 1):
        t = int(rng.integers(1, 9))
        a = np.zeros(3)
        if t in (7, 8):
            a = rng.normal(size=3); a /= np.linalg.norm(a)
        else:
            a[(t - 1) % 3] = 1.0
        types.append(t); axis.append(a)
        placement.append(np.concatenate([random_rotation(rng).ravel(), rng.uniform(-0.3, 0.3, size=3)]))
    m = loik_amd.Model(parents, types, np.array(axis), np.array(placement), q_lo=-np.ones(nb), q_hi=np.ones(nb),
                       name="bushy_%d_%d_%dx%d" % (seed, nb, hub, hub_children))
    children = np.bincount(np.asarray(m.parents[1:]), minlength=m.njoints)
    assert children.max() >= hub_children
    return m


CASES = {
    # the fuzz's refusal (profiles/r05_j_fuzz_3000.txt case 903): 42 joints, ten children at one joint -- depth-first: the flat engine (k_flat1)
    "42_joints_10_children": dict(nb=42, hub=2, children=10, depth_first=True, engine="flat"),
    # the same tree numbered breadth-first below the hub (outside the flat engine's domain; k_lean takes four children at most): the sweeps of
    # k_solve hand over along ten short chains one after the other and need few slots -- the streaming engine + k_tail, as for any tree
    "42_joints_10_children_not_depth_first": dict(nb=42, hub=2, children=10, depth_first=False, engine="solve"),
    # more than 64 joints: no on-chip engine -- the engine of last resort
    "100_joints_9_children": dict(nb=100, hub=3, children=9, depth_first=True, engine="pass"),
}


def _check_engine(s, st, engine, B):
    plan = s.plan()
    if engine == "solve":
        assert "too bushy" not in plan and st["flat_launches"] == 0 and st["tail_instances"] == B, (plan, st)
        return
    assert "too bushy for k_solve" in plan, plan
    if engine == "flat":
        assert st["flat_launches"] >= 1 and st["tail_instances"] == B, (plan, st)
    else:
        assert "k_pass_solve" in plan and st["tail_instances"] == 0 and st["launches"] == 1, (plan, st)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("nc", [1, 3])
def test_bushy_tree_is_solved_not_refused(case, nc):
    c = CASES[case]
    model = bushy_tree(77, c["nb"], c["hub"], c["children"], c["depth_first"])
    B = 130
    links = [model.njoints - 1, model.njoints - 1 - c["nb"] // c["children"], c["hub"] + 1][:nc]
    wl = multi_task_batch(model, B, links, 11 + nc, nu_scale=0.3)
    for k in (1, 2, 5):
        prm = dict(FIXTURE, num_eq_c=nc, max_iter=k + 1, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
        s = loik_amd.BatchedLoik(model, B, **prm)
        s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        _check_engine(s, s.stats(), c["engine"], B)
        got = {n: s.get(n) for n in FIELDS + SCALARS}
        got["His"] = s.His_full()
        for b in range(0, B, 19):
            r = ref.RefSolver(model, **prm)
            r.Solve(*problem_args(wl, b))
            for n in FIELDS:
                want = r.field(n)
                if n in ("vis", "fis", "g"):
                    want = want[1:]
                assert_close(got[n][b], want, 1e-9, "%s b%d k%d" % (n, b, k))
            assert_close(got["His"][b], r.His[1:], 1e-9, "His")
            for n in SCALARS:
                assert_close(got[n][b], r.scalar(n), 1e-9, "%s b%d k%d" % (n, b, k))
        s.close()
    prm = dict(FIXTURE, num_eq_c=nc, max_iter=300, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                          nthreads=8, want_nu=True, **prm)
    s = loik_amd.BatchedLoik(model, B, **prm)
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    _check_engine(s, s.stats(), c["engine"], B)
    assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=2e-10, what=case)
    s.close()


@pytest.mark.gpu
def test_bushy_tree_with_options_that_ask_for_k_solve_takes_the_engine_of_last_resort():
    """tail_max_instances < 0 ("the solve kernel alone") and max_launch_iters > 0 are k_solve's own controls: on a tree it cannot
    take, the solve runs on k_pass_solve -- split Solve() calls included (the reference's loop is one call; max_launch_iters is not
    honoured there, the answer is the one-shot solve's)."""
    model = bushy_tree(78, 42, 2, 10)
    B = 70
    wl = multi_task_batch(model, B, [model.njoints - 1], 5, nu_scale=0.3)
    prm = dict(FIXTURE, num_eq_c=1, max_iter=200, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"], nthreads=4, want_nu=True, **prm)
    for kw in (dict(tail_max_instances=-1), dict(max_launch_iters=7)):
        s = loik_amd.BatchedLoik(model, B, **prm, **kw)
        s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        assert "k_pass_solve" in s.plan(), s.plan()
        assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=2e-10, what=str(kw))
        s.close()
