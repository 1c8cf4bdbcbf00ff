import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, loik_amd
from loik_amd import capi, workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for tol in (1e-3, 1e-4):
    wl = workloads.panda_c5(B, tol=tol)
    for name, prec in (("fp64", capi.F64), ("fp32", capi.F32)):
        s = loik_amd.BatchedLoik(wl["model"], B, precision=prec, **wl["params"])
        s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        t = []
        for i in range(6):
            s.Solve(); t.append(s.stats()["total_ms"])
        st = s.stats()
        print(os.environ.get("LOIKB_LEAN_SMALL", "-"), tol, name, " ".join("%.2f" % x for x in t), "lean", st["lean_launches"], "ordered", st["flat_ordered"], "conv %.3f" % s.get("converged").mean())
        s.close()
