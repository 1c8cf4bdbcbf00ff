"""why a lone handle's Solve() of a new batch is slower than the bench's back-to-back pool: the GPU's clock state after an idle gap?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = 65536
wl0 = workloads.talos_c3(B)
args = lambda w: (w["q"], w["H_ref"], w["v_ref"], w["c_ids"], w["Ais"], w["bis"], w["lb"], w["ub"])
batches = [workloads.talos_c3(B, seed=0x5EED + i) for i in range(8)]
def run(label, gap_s=0.0, prewarm=False, pregenerated=True):
    s = loik_amd.BatchedLoik(wl0["model"], B, **wl0["params"])
    warm = loik_amd.BatchedLoik(wl0["model"], B, **wl0["params"])
    warm.SolveInit(*args(wl0)); warm.Solve()
    rows = []
    for i in range(8):
        w = batches[i] if pregenerated else workloads.talos_c3(B, seed=0x5EED + i)
        t_i = time.perf_counter(); s.SolveInit(*args(w)); s.synchronize(); t_init = (time.perf_counter() - t_i) * 1e3
        if gap_s: time.sleep(gap_s)
        if prewarm:
            warm.Solve(); warm.Solve(); warm.synchronize()
        t0 = time.perf_counter(); s.Solve(); s.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        st = s.stats()
        if i >= 2: rows.append((dt, st["tail_ms"], st["hslots_ms"], t_init))
    r = np.array(rows).mean(axis=0)
    print("%-70s wall %.2f ms  on-chip launch %.2f  slots %.2f   (SolveInit %.1f ms)" % (label, r[0], r[1], r[2], r[3]))
    s.close(); warm.close()
run("one handle, batches generated between the solves (host busy ~0.3 s)", pregenerated=False)
run("one handle, batches ready: the gap is SolveInit only")
run("one handle, batches ready, 100 ms of idleness before Solve()", gap_s=0.1)
run("one handle, batches ready, two solves of ANOTHER handle right before Solve()", prewarm=True)
