"""How often does an iteration walk the full stopping logic?  (-DLOIKB_TAIL_PROF -DLOIKB_DBG_QUIET build: g_tail_prof_all[24..29])
[24] iterations with the main loop's logic, [26] of them NOT quiet, [28] mu changes, [29] the certificate's first test holds, [27] tail-solve iterations"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads
L = capi.lib()
for name, mk, B in (("talos32 headline", workloads.talos_c3, 65536), ("talos44 whole body", workloads.talos_wholebody, 65536)):
    wl = mk(B)
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    z = (C.c_ulonglong * 32)()
    L.loikb_debug_tail_prof_all(z, 1)
    s.Solve()
    a = (C.c_ulonglong * 32)()
    L.loikb_debug_tail_prof_all(a, 1)
    it = s.get("iter")
    print("%s: instance-iterations %d; with the main logic %d; not quiet %d (%.1f %%); mu changes %d (%.2f %%); certificate's first test holds %d (%.1f %%); tail-solve iterations %d" % (
        name, int(it.sum()), a[24], a[26], 100.0 * a[26] / max(a[24], 1), a[28], 100.0 * a[28] / max(a[24], 1), a[29], 100.0 * a[29] / max(a[24], 1), a[27]))
    long_ = it >= 999
    print("   instances at max_iter: %d (%.2f %%), their share of the iterations %.1f %%" % (long_.sum(), 100.0 * long_.mean(), 100.0 * it[long_].sum() / it.sum()))
    s.close()
