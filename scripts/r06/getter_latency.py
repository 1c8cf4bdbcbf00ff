"""what one getter call costs a small batch (Talos-32, after a Solve()): the scalar fields, z, and two rebuilt members"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, loik_amd
from loik_amd import workloads
for B in (1, 64):
    wl = workloads.talos_c3(B, seed=3)
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    for f in ("iter", "converged", "primal_residual", "z", "vis", "His", "pis"):
        ts = []
        for _ in range(20):
            s.Solve()
            t = time.perf_counter(); s.get(f); ts.append(time.perf_counter() - t)
        print("B %3d get(%s): %.4f ms" % (B, f, min(ts) * 1e3), flush=True)
