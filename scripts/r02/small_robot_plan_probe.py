"""small robots (Panda-7): lean engine vs k_solve + k_tail as the solves get longer (tolerance), fp32 (where the plan picked the
lean engine) -- run once with LOIKB_LEAN unset and once with LOIKB_LEAN=0"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads
for tol, mi in ((1e-3, 300), (1e-4, 300), (1e-5, 1000)):
    wl = workloads.panda_c5(65536, tol=tol)
    prm = dict(wl["params"], max_iter=mi)
    for prec, name in ((capi.F32, "fp32"), (capi.F64, "fp64")):
        s = loik_amd.BatchedLoik(wl["model"], 65536, precision=prec, **prm)
        s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        best = 1e9
        for _ in range(4):
            t = time.perf_counter(); s.Solve(); best = min(best, time.perf_counter() - t)
        print("LEAN=%s tol %g %s: %.3f ms, mean iters %.1f, converged %.3f, lean launches %d" % (
            os.environ.get("LOIKB_LEAN", "default"), tol, name, best * 1e3, s.get("iter").mean(), s.get("converged").mean(), s.stats()["lean_launches"]), flush=True)
        s.close()
