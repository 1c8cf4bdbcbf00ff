"""finer sweep of the tail hand-over threshold / tail launch budget on the headline workload"""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loik_amd
from loik_amd import workloads
B = 65536
wl = workloads.talos_c3(B)
for rnd, tmax in itertools.product([32, 64, 128], [3072, 4096, 6144, 8192, 12288]):
    os.environ["LOIKB_TAIL_ROUND"] = str(rnd)
    s = loik_amd.BatchedLoik(wl["model"], B, tail_max_instances=tmax, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); s.Solve(); dt = time.perf_counter() - t
        best = min(best, dt)
    st = s.stats()
    print("round %4d tail_max %6d -> %.1f ms/step; kernel %.1f ms (tail %.1f ms for %d inst), launches %d" % (
        rnd, tmax, best * 1e3, st["kernel_ms"], st["tail_ms"], st["tail_instances"], st["launches"]))
    s.close()
