"""hand-over threshold / chunks with the lean tail kernel (LOIKB_LEAN=1)"""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loik_amd
from loik_amd import workloads
B = 65536
wl = workloads.talos_c3(B)
os.environ["LOIKB_LEAN"] = "1"; os.environ.setdefault("LOIKB_LEAN_KLO", "0"); os.environ.setdefault("LOIKB_LEAN_DECADES", "8")
for chunks, tm in itertools.product([1, 2], [24576, 32768, 40960, 53000, 1 << 20]):
    os.environ["LOIKB_CHUNKS"] = str(chunks)
    s = loik_amd.BatchedLoik(wl["model"], B, tail_max_instances=tm, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    ts = []
    for _ in range(5):
        t = time.perf_counter(); s.Solve(); ts.append(time.perf_counter() - t)
    st = s.stats()
    print("chunks %d tail_max %7d -> best %.2f median %.2f ms/step; solve busy %.1f tail busy %.1f ms (%d inst), launches %d" % (
        chunks, tm, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3, st["solve_busy_ms"], st["tail_busy_ms"], st["tail_instances"], st["launches"]), flush=True)
    s.close()
