"""two (or N) headline batches in flight on one GPU (handles on streams of their own, one host thread each): ms per round of N batches"""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 65536
hs = []
for g in range(N):
    wl = workloads.talos_c3(B, seed=0x101C + 3 + g)
    s = loik_amd.BatchedLoik(wl["model"], B, flags=capi.OPT_OWN_STREAM, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    s.Solve(); s.Solve()
    hs.append(s)
steps = 5
bar = threading.Barrier(N + 1)
def work(s):
    bar.wait()
    for _ in range(steps):
        s.Solve()
    s.synchronize()
    bar.wait()
ts = [threading.Thread(target=work, args=(s,)) for s in hs]
for t in ts: t.start()
bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
for t in ts: t.join()
solved = sum(int(s.get("converged").astype(bool).sum()) for s in hs)
print("%s %d batches in flight: %.2f ms per round of %d, %.2f M solves/s, ordered launches %s" % (
    os.environ.get("TAG", ""), N, dt / steps * 1e3, N, solved * steps / dt / 1e6, [s.stats()["flat_ordered"] for s in hs]))
