"""The arithmetic of the flat engine (tests/flat_numpy.py: world-frame sums and the explicit scalar factor instead of the two
level-by-level recursions of /root/reference/include/loik/loik-loid-optimized.hxx:31-81, :102-163) against the CPU oracle:
same iteration counts, same flags, same answers.  CPU only."""
import numpy as np

import loik_amd
from loik_amd import workloads
from oracle import ref
from flat_numpy import Flat
from helpers import FIXTURE, multi_task_batch, random_tree


def _check(wl, same_frac=1.0, ztol=1e-9):
    out = ref.solve_batch(wl["model"], wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                          nthreads=4, want_nu=True, **wl["params"])
    fl = Flat(wl).solve()
    same = fl["iters"] == out["iters"]
    assert same.mean() >= same_frac, (same.mean(), fl["iters"][~same], out["iters"][~same])
    assert np.array_equal(fl["converged"][same], out["converged"][same])
    assert np.array_equal(fl["primal_infeasible"][same], out["primal_infeasible"][same])
    assert np.abs(fl["z"] - out["z"])[same].max() < ztol and np.abs(fl["nu"] - out["nu"])[same].max() < ztol
    assert np.allclose(fl["res"][same, 0], out["primal_residual"][same], rtol=1e-6, atol=1e-10)
    assert np.allclose(fl["res"][same, 1], out["dual_residual"][same], rtol=1e-6, atol=1e-10)
    return fl, out


def test_headline_workload_same_iteration_counts():
    """Talos, one task on the left wrist, tol 1e-6, adaptive mu: every instance of the sample incl. those that run all 999
    iterations stops where the oracle stops (8192 instances: scripts/r03/flat_proto.py, 100 %)"""
    wl = workloads.talos_c3(96)
    fl, out = _check(wl)
    assert (out["iters"] >= 999).any() or out["iters"].max() > 200   # the sample holds long runners


def test_two_constraints_general_reference_cost():
    """a random tree, two task constraints, a general symmetric H_ref and a non-zero v_ref, relative tolerances"""
    model = random_tree(4, 24)
    wl = multi_task_batch(model, 48, [model.njoints - 1, model.njoints // 2], 7)
    rng = np.random.default_rng(5)
    M = rng.normal(size=(6, 6))
    wl["H_ref"] = np.eye(6) + 0.1 * (M + M.T)
    wl["v_ref"] = 0.1 * rng.normal(size=6)
    wl["params"] = dict(FIXTURE, num_eq_c=2, max_iter=300, tol_abs=1e-7, tol_rel=1e-7)
    wl["model"] = model
    _check(wl, same_frac=0.97, ztol=1e-8)
