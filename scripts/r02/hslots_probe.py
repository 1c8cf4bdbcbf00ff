import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = 65536
wl = workloads.talos_c3(B)
args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
for nd, klo in ((10, -2), (9, -1), (8, 0), (7, 0), (6, 0), (5, 0), (6, -1), (4, 0)):
    os.environ["LOIKB_LEAN_DECADES"] = str(nd); os.environ["LOIKB_LEAN_KLO"] = str(klo)
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.SolveInit(*args)
    hs = []
    for _ in range(3):
        s.Solve(); hs.append(s.stats()["hslots_ms"])
    st = s.stats()
    it = s.get("iter"); conv = s.get("converged")
    print("ndec %2d klo %2d: hslots %.3f ms  total %.2f ms  escaped %d  (solved %d, iterations %d)" % (nd, klo, min(hs), st["total_ms"], st["lean_escaped"], conv.sum(), it.sum()), flush=True)
    s.close()
