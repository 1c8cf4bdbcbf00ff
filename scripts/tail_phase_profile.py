"""cycles per phase of one k_tail wavefront (needs a library built with -DLOIKB_TAIL_PROF:
   python -c "from loik_amd import _build; _build.build(force=True, extra_flags=('-DLOIKB_TAIL_PROF',))")"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads
L = capi.lib()
NAMES = ["0 H-cache check, p base", "1 leaf->root level loop", "2 root->leaf level loop", "3 per-joint work (f, box, w)",
         "4 task dual update", "5 residual exchange (g, s)", "6 norm reductions", "7 epilogue + instance switch"]
iters = 200
for B in [int(x) for x in sys.argv[1:]] or [64, 4096]:  # >= 64 instances: the lean kernel; LOIKB_LEAN=0: k_tail
    wl = workloads.talos_c3(B, seed=5)
    # the real stopping rule (with tolerances 0 mu drifts through the decades after convergence: not the lean kernel's case)
    prm = dict(wl["params"])
    s = loik_amd.BatchedLoik(wl["model"], B, tail_max_instances=1 << 24, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    for _ in range(2):
        s.Solve()
    st = s.stats()
    out = (C.c_ulonglong * 14)()
    assert L.loikb_debug_tail_prof(out) == 0
    n = out[8]
    tot = sum(out[:8]) + sum(out[10:14])
    print("B=%d: tail %.2f ms; wavefront 0: %d iterations, %.0f cycles per iteration%s" % (
        B, st["tail_ms"], n, tot / n, ("; clock64 runs at %.0f MHz against the 100 MHz wall clock" % (out[9] / 1e3)) if out[9] else ""))
    for k in range(8):
        print("   %-34s %8.0f cycles  %5.1f %%" % (NAMES[k], out[k] / n, 100.0 * out[k] / tot))
    for k, nm in enumerate(["8 loop top + decade slot load (was in 0)", "9 switch: store_instance", "10 switch: ticket + ring entry",
                            "11 switch: load_instance"]):
        print("   %-34s %8.0f cycles  %5.1f %%" % (nm, out[10 + k] / n, 100.0 * out[10 + k] / tot))
    s.close()
