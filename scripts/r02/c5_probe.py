"""C5 (Panda-7, B = 65536, tol 1e-3): where does the time go in fp64 and in fp32?  stats + plan of both handles"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads
wl = workloads.panda_c5(65536)
m = wl["model"]
for prec, name in ((capi.F64, "fp64"), (capi.F32, "fp32")):
    for kw in ({}, dict(tail_max_instances=-1), dict(tail_max_instances=1 << 20)):
        s = loik_amd.BatchedLoik(m, 65536, precision=prec, **wl["params"], **kw)
        s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        best = 1e9
        for _ in range(4):
            t = time.perf_counter(); s.Solve(); best = min(best, time.perf_counter() - t)
        st = s.stats()
        print(name, kw, "ms %.3f" % (best * 1e3), "iters mean %.2f" % s.get("iter").mean(), "conv %.3f" % s.get("converged").mean(),
              {k: st[k] for k in ("launches", "kernel_ms", "tail_ms", "tail_instances", "tail_launches", "lean_launches", "compactions", "hslots_ms", "total_ms")})
        if not kw:
            print("   plan:", s.plan())
        s.close()
