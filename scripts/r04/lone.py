"""iteration time of a LONE instance (64-instance batch: launch time / longest instance's iterations), Talos-32 and the whole body"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loik_amd
from loik_amd import workloads
for name, mk in (("talos32", workloads.talos_c3), ("talos44 whole body", workloads.talos_wholebody)):
    wl = mk(64)
    s = loik_amd.BatchedLoik(wl["model"], 64, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    best = 1e9
    for _ in range(5):
        s.Solve()
        st = s.stats()
        own = st["tail_ms"] - st["hslots_ms"]
        best = min(best, own * 1e3 / max(int(s.get("iter").max()), 1))
    print("%s %s: lone instance %.3f us per iteration (launch %.2f ms, longest %d iterations)" % (os.environ.get("TAG", ""), name, best, own, int(s.get("iter").max())))
    s.close()
