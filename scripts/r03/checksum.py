"""bitwise fingerprint of the results of the headline and whole-body batches (A/B of two builds: every member must hash the same)"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
for mk, B in ((workloads.talos_c3, 16384), (workloads.talos_wholebody, 8192)):
    wl = mk(B)
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    s.Solve(); s.Solve()
    h = hashlib.sha256()
    for n in ("iter", "converged", "primal_infeasible", "mu", "z", "nu", "w", "vis", "fis", "g", "yis", "Aty", "primal_residual", "dual_residual", "Stf_plus_w"):
        h.update(np.ascontiguousarray(s.get(n)).tobytes())
    print(wl["name"], s.stats()["flat_launches"], int(s.get("iter").sum()), h.hexdigest()[:16])
    s.close()
