"""a caller with ONE handle and a new batch before every solve (SolveInit + Solve) against the bench's pool of handles (each: one history
solve, then its fresh batch): wall time of Solve(), the call's HIP-event time, the slot kernel's share, the plan's decade window"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = 65536
wl0 = workloads.talos_c3(B)
args = lambda w: (w["q"], w["H_ref"], w["v_ref"], w["c_ids"], w["Ais"], w["bis"], w["lb"], w["ub"])
s = loik_amd.BatchedLoik(wl0["model"], B, **wl0["params"])
print("ONE handle, another batch before every solve:")
for i in range(7):
    w = workloads.talos_c3(B, seed=0x5EED + i)
    s.SolveInit(*args(w)); s.synchronize()
    t0 = time.perf_counter(); s.Solve(); s.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    st = s.stats()
    print("  solve %d: wall %.2f ms  total_ms %.2f  on-chip launch %.2f  slots %.2f  dry at %.2f  requeues %d  ordered %d | %s" % (
        i, dt, st["total_ms"], st["tail_ms"], st["hslots_ms"], st["queue_dry_ms"], st["lean_requeues"], st["flat_ordered"], s.plan().split("decades visited")[-1][:60]))
s.close()
print("pool: a handle per batch, one history solve each:")
pool = []
for i in range(5):
    h = loik_amd.BatchedLoik(wl0["model"], B, **wl0["params"])
    h.SolveInit(*args(wl0)); h.Solve()
    h.SolveInit(*args(workloads.talos_c3(B, seed=0x5EED + i)))
    pool.append(h)
for h in pool: h.synchronize()
t0 = time.perf_counter()
for i, h in enumerate(pool):
    t1 = time.perf_counter(); h.Solve(); dt = (time.perf_counter() - t1) * 1e3
    st = h.stats()
    print("  solve %d: wall %.2f ms  total_ms %.2f  on-chip launch %.2f  slots %.2f  dry at %.2f  requeues %d  ordered %d | %s" % (
        i, dt, st["total_ms"], st["tail_ms"], st["hslots_ms"], st["queue_dry_ms"], st["lean_requeues"], st["flat_ordered"], h.plan().split("decades visited")[-1][:60]))
print("  pool mean wall per solve (incl. stats()): %.2f ms" % ((time.perf_counter() - t0) * 1e3 / len(pool)))
for h in pool: h.close()
