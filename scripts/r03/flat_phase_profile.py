"""cycles per phase of wavefront 0 of a k_flat launch (needs a -DLOIKB_TAIL_PROF build: scripts/r03/prof_build.sh)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads
L = capi.lib()
NAMES = {8: "loop top + decade slot change", 0: "p^base sums, tau", 1: "r' = W tau (products, shares, partials)",
         2: "nu = -W^T Dinv r', path sum, v", 4: "task dual update", 5: "subtree sum, f", 3: "per-joint work (box, w, norms)",
         6: "norm fold", 7: "epilogue + instance switch", 9: "  task: v -> LDS, A v - b, y (two fences)", 10: "  subtree prefix sums of E (DPP) + exchange",
         11: "  task: A^T y, A^T dy at the origin (two fences)"}
NAMES[4] = "  per-joint work on v, nu (box, w, g, norms)"
NAMES.update({12: "instance: ticket, list entry", 13: "instance: record loads, joint placement", 14: "instance: world placements (pointer jumping)",
              15: "instance: constraint blocks at the origin", 16: "instance: subtree sums, scalars", 17: "decade slot loads (HBM)",
              18: "stop: the getters' folds", 19: "instance: store"})
ORDER = [8, 0, 1, 2, 9, 10, 11, 4, 5, 3, 6, 7, 12, 13, 14, 15, 16, 17, 18, 19]
for B in [int(x) for x in sys.argv[1:]] or [64, 65536]:
    wl = workloads.talos_c3(B, seed=5)
    prm = dict(wl["params"])
    s = loik_amd.BatchedLoik(wl["model"], B, tail_max_instances=1 << 24, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    if B <= 256:   # a lone wavefront per instance: put the longest instance first, so that wavefront 0's timeline is a long solve's
        s.Solve()
        first = int(np.argmax(s.get("iter")))
        order = np.r_[first, np.delete(np.arange(B), first)]
        wl["q"], wl["bis"] = wl["q"][order], wl["bis"][order]
        s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    s.Solve()
    z = (C.c_ulonglong * 32)()
    L.loikb_debug_tail_prof_all(z, 1)
    s.Solve()
    st = s.stats()
    out = (C.c_ulonglong * 32)()
    assert L.loikb_debug_tail_prof(out) == 0
    n = out[8]
    idx = lambda k: k if k < 8 else 2 + k
    tot = sum(out[idx(k)] for k in ORDER)
    mhz = out[9] / 1e3
    print("B=%d (%s): tail %.2f ms (slots %.2f); wavefront 0: %d iterations, %.0f cycles = %.2f us per iteration (clock64 at %.0f MHz)" % (
        B, "flat" if st["flat_launches"] else "NOT flat", st["tail_ms"], st["hslots_ms"], n, tot / n, tot / n / mhz, mhz))
    for k in ORDER:
        print("   %-44s %8.0f cycles  %5.1f %%" % (NAMES[k], out[idx(k)] / n, 100.0 * out[idx(k)] / tot))
    al = (C.c_ulonglong * 32)()
    L.loikb_debug_tail_prof_all(al, 1)
    na, nw = al[8], al[9]
    tota = sum(al[idx(k)] for k in ORDER)
    print("  ALL %d wavefronts of the launch: %d iterations, %.0f cycles per iteration of wavefront time (incl. waits for the queue)" % (nw, na, tota / max(na, 1)))
    for k in ORDER:
        print("   %-44s %8.0f cycles  %5.1f %%" % (NAMES[k], al[idx(k)] / max(na, 1), 100.0 * al[idx(k)] / max(tota, 1)))
    s.close()
