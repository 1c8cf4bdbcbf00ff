"""randomised cross-check of the default engine against the CPU oracle: random trees (1-DoF and multi-DoF joints), random
reference costs, 0-2 task constraints, shared / per-instance A and bounds.  Not part of the test suite (minutes)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import loik_amd
from loik_amd import workloads
from helpers import FIXTURE, random_tree, random_tree_multidof, multi_task_batch
from oracle import ref

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(12345)
worst = 0.0
bad = 0
for case in range(ncase):
    nb = int(rng.integers(3, 41))
    multidof = rng.random() < 0.3
    seed = int(rng.integers(1, 10000))
    model = random_tree_multidof(seed, nb, root_freeflyer=bool(rng.random() < 0.5), n_spherical=int(rng.integers(0, 2)),
                                 n_translation=int(rng.integers(0, 2))) if multidof and nb >= 5 else random_tree(seed, nb)
    nc = int(rng.integers(0, 3))
    B = int(rng.choice([70, 130, 256, 600]))
    links = [int(x) for x in rng.choice(np.arange(1, model.njoints), size=max(nc, 1), replace=False)]
    wl = multi_task_batch(model, B, links, seed + 1, bound=0.5, nu_scale=0.4, per_instance_A=bool(rng.random() < 0.3))
    if nc == 0:
        wl["c_ids"] = np.zeros(0, dtype=np.int32); wl["Ais"] = np.zeros((0, 6, 6)); wl["bis"] = np.zeros((B, 0, 6))
    kind = rng.integers(0, 3)
    if kind == 1:
        wl["H_ref"] = np.diag(rng.uniform(0.3, 2.0, size=6)); wl["v_ref"] = 0.2 * rng.normal(size=6)
    elif kind == 2:
        M = rng.normal(size=(6, 6)); wl["H_ref"] = M @ M.T / 6 + 0.5 * np.eye(6); wl["v_ref"] = 0.2 * rng.normal(size=6)
    if rng.random() < 0.3:
        wl["lb"] = -0.5 * (1 + 0.2 * rng.random((B, model.nv))); wl["ub"] = 0.5 * (1 + 0.2 * rng.random((B, model.nv)))
    prm = dict(FIXTURE, num_eq_c=nc, max_iter=int(rng.choice([60, 300, 1000])), tol_abs=float(rng.choice([1e-4, 1e-6, 1e-8])),
               tol_rel=float(rng.choice([0.0, 1e-6])))
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                          nthreads=8, **prm)
    s = loik_amd.BatchedLoik(model, B, **prm)
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    st = s.stats()
    it = s.get("iter"); same = it == out["iters"]
    dz = float(np.max(np.abs(s.get("z") - out["z"])[same])) if same.any() else 0.0
    flags_ok = np.array_equal(s.get("converged").astype(bool)[same], out["converged"][same]) and \
        np.array_equal(s.get("primal_infeasible").astype(bool)[same], out["primal_infeasible"][same])
    ok = same.mean() >= 0.95 and dz < 1e-6 and flags_ok
    worst = max(worst, dz); bad += not ok
    print("case %2d nb %2d nv %2d nc %d B %3d multidof %d Href %d max_iter %4d tol %.0e: same-iteration %.3f  max|dz| %.2e  flags %s  lean %d esc %d  %s" % (
        case, nb, model.nv, nc, B, multidof, kind, prm["max_iter"], prm["tol_abs"], same.mean(), dz, flags_ok, st["lean_launches"],
        st["lean_escaped"], "ok" if ok else "MISMATCH"), flush=True)
    s.close()
print("cases", ncase, "mismatches", bad, "worst |dz|", worst)
