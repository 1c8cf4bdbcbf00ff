"""how many instances of the headline batch leave a window of K decades of mu starting at mu0 (k_fslots builds the window; an instance that
leaves it is finished by k_tail): escapes and times per window.  LOIKB_LEAN_KLO / LOIKB_LEAN_DECADES / LOIKB_LEAN_ADAPT=0"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = workloads.talos_c3(B)
a = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
os.environ["LOIKB_LEAN_ADAPT"] = "0"
os.environ["LOIKB_FLAT_ORDER"] = "0"
for klo, nd in ((-2, 10), (0, 8), (0, 6), (0, 5), (0, 4), (0, 3), (0, 2), (1, 3), (1, 4)):
    os.environ["LOIKB_LEAN_KLO"] = str(klo); os.environ["LOIKB_LEAN_DECADES"] = str(nd)
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.SolveInit(*a)
    s.Solve(); s.Solve()
    st = s.stats()
    mu = s.get("mu"); it = s.get("iter")
    print("decades %2d..%2d: escaped %6d (%.2f %%)  total %.2f ms  slots %.2f ms  tail engine share: %d inst" % (
        klo, klo + nd - 1, st["lean_escaped"], 100.0 * st["lean_escaped"] / B, st["total_ms"], st["hslots_ms"], st["lean_escaped"]))
    if nd == 10:
        k = np.round(np.log10(mu / 1e-2)).astype(int)
        print("   final decade histogram:", dict(zip(*np.unique(k, return_counts=True))))
        for lo, hi in ((0, 30), (30, 100), (100, 999), (999, 2000)):
            m = (it >= lo) & (it < hi)
            print("   iters [%d,%d): %d instances, final decade hist" % (lo, hi, m.sum()), dict(zip(*np.unique(k[m], return_counts=True))))
    s.close()
