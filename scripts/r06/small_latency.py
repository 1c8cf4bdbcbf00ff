"""Wall-clock latency of a cold Solve() for small batches (Talos-32, headline parameters), the reference's own use -- one problem per call,
tests/loik-loid.cpp:987-1032.  usage: small_latency.py [B ...]   (engine switches from the environment)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
Bs = [int(a) for a in sys.argv[1:]] or [1, 8, 64, 256, 1024]
for B in Bs:
    wl = workloads.talos_c3(B, seed=3)
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    ts = []
    for _ in range(30):
        t = time.perf_counter(); s.Solve(); ts.append(time.perf_counter() - t)
    st = s.stats()
    it = s.get("iter")
    print(json.dumps({"tag": os.environ.get("TAG", ""), "batch": B, "solve_wall_ms_min": round(min(ts) * 1e3, 4), "solve_wall_ms_median": round(float(np.median(ts)) * 1e3, 4),
                      "max_iterations": int(it.max()), "mean_iterations": round(float(it.mean()), 2), "kernel_ms": round(st["kernel_ms"], 4), "total_ms": round(st["total_ms"], 4),
                      "launches": st["launches"], "flat_launches": st["flat_launches"], "tail_launches": st["tail_launches"], "slots_ms": round(st["hslots_ms"], 4)}), flush=True)
    s.close()
