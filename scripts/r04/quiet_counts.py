"""debug build (-DLOIKB_DBG_QUIET): how many iterations run the main loop's stopping logic, how many the quick look decides"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loik_amd
from loik_amd import workloads, capi
L = capi.lib()
for B in (64, 65536):
    wl = workloads.talos_c3(B)
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    z = (C.c_ulonglong * 32)()
    L.loikb_debug_tail_prof_all(z, 1)
    s.Solve()
    L.loikb_debug_tail_prof_all(z, 1)
    it = s.get("iter")
    print("B=%d: instance-iterations %d (longest %d); with main-loop logic %d, quick look quiet %d, exact not quiet %d (mu change wanted %d, certificate's first test passes %d), tail-solve iterations %d" % (
        B, int(it.sum()), int(it.max()), z[24], z[25], z[26], z[28], z[29], z[27]))
    s.close()
