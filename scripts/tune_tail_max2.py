import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loik_amd
from loik_amd import workloads
B = 65536
wl = workloads.talos_c3(B)
for chunks, tm, li in [(2, 24576, 0), (2, 32768, 0), (2, 40960, 0), (2, 49152, 0), (2, 65536, 0), (2, 1 << 20, 0), (1, 1 << 20, 0),
                       (2, 32768, 4), (2, 32768, 6), (2, 49152, 4), (2, 40960, 4), (4, 1 << 20, 0), (2, 32768, 12)]:
    os.environ["LOIKB_CHUNKS"] = str(chunks)
    s = loik_amd.BatchedLoik(wl["model"], B, tail_max_instances=tm, max_launch_iters=li, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    ts = []
    for _ in range(4):
        t = time.perf_counter(); s.Solve(); ts.append(time.perf_counter() - t)
    st = s.stats()
    print("chunks %d tail_max %7d launch_iters %2d -> best %.2f median %.2f ms/step; solve busy %.1f tail busy %.1f ms (%d inst), launches %d" % (
        chunks, tm, li, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3, st["solve_busy_ms"], st["tail_busy_ms"], st["tail_instances"], st["launches"]), flush=True)
    s.close()
