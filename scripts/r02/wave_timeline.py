"""per-wavefront timeline of the lean launch (needs a -DLOIKB_TAIL_PROF build)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads
L = capi.lib()
B = 65536
wl = workloads.talos_c3(B)
s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
for _ in range(2):
    s.Solve()
st = s.stats()
buf = (C.c_ulonglong * (4096 * 6))()
assert L.loikb_debug_wave_dbg(buf) == 0
a = np.array(buf, dtype=np.float64).reshape(4096, 6)
a = a[a[:, 3] > 0]
t0 = a[:, 0].min()
start, last, end = (a[:, 0] - t0) / 1e5, (a[:, 1] - t0) / 1e5, (a[:, 2] - t0) / 1e5   # ms
print("slice %s: total %.2f ms; %d wavefronts; start spread %.3f ms" % (os.environ.get("LOIKB_LEAN_SLICE"), st["total_ms"], len(a), start.max()))
print("  last-work time: min %.2f  p10 %.2f  median %.2f  p90 %.2f  p99 %.2f  max %.2f" % (last.min(), *np.quantile(last, [.1, .5, .9, .99]), last.max()))
print("  exit time:      min %.2f  median %.2f  max %.2f" % (end.min(), np.median(end), end.max()))
print("  wave-iters: mean %.0f  min %d  max %d; both-groups-active share %.3f; switches per wave mean %.1f" % (a[:, 3].mean(), a[:, 3].min(), a[:, 3].max(), a[:, 4].sum() / a[:, 3].sum(), a[:, 5].mean()))
per_it = (last - start) * 1e3 / a[:, 3]
print("  us per wave-iteration (busy span / iters): mean %.2f  p10 %.2f p90 %.2f" % (per_it.mean(), *np.quantile(per_it, [.1, .9])))
# how many wavefronts still working over time
for t in np.arange(2, end.max() + 2, 2.0):
    print("   t=%5.1f ms: %4d wavefronts still have work" % (t, int((last > t).sum())))
