"""sweep of compaction ratio / iterations per launch on the headline workload"""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loik_amd
from loik_amd import workloads
B = 65536
wl = workloads.talos_c3(B)
for ratio, li in itertools.product([0.75, 0.85, 0.92, 0.97], [4, 6, 8, 12]):
    os.environ["LOIKB_COMPACT_RATIO"] = str(ratio)
    s = loik_amd.BatchedLoik(wl["model"], B, max_launch_iters=li, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); s.Solve(); dt = time.perf_counter() - t
        best = min(best, dt)
    st = s.stats()
    print("ratio %.2f launch_iters %2d -> %.1f ms/step; kernel %.1f ms (tail %.1f ms for %d inst), launches %d compactions %d" % (
        ratio, li, best * 1e3, st["kernel_ms"], st["tail_ms"], st["tail_instances"], st["launches"], st["compactions"]))
    s.close()
