import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = 65536
wl = workloads.talos_c3(B)
m = wl["model"]
args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
for strat, kw in [(0, {}), (1, {}), (1, dict(tail_max_instances=1 << 20)), (1, dict(tail_max_instances=8192))]:
    prm = dict(wl["params"], mu_update_strat=strat)
    s = loik_amd.BatchedLoik(m, B, **prm, **kw)
    s.SolveInit(*args)
    ts = []
    for _ in range(4):
        s.Solve(); ts.append(s.stats()["total_ms"])
    it = s.get("iter"); conv = s.get("converged").astype(bool)
    st = s.stats()
    print("strat %d %s: %.2f ms  solves/s %.3g  mean it %.1f p99 %d hitmax %.4f conv %.3f infeas %.3f | %s" % (
        strat, kw, np.median(ts), conv.sum() / np.median(ts) * 1e3, it.mean(), np.quantile(it, .99), ((it >= 999) & ~conv).mean(), conv.mean(),
        s.get("primal_infeasible").astype(bool).mean(), s.plan()[:60]))
    s.close()
