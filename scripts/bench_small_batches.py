"""cold Solve() latency for small batches (Talos-32, headline parameters): tail kernel from the first iteration
(default below the hand-over threshold) vs the solve kernel first (LOIKB_NO_DIRECT_TAIL=1)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loik_amd
from loik_amd import workloads
for B in (1, 64, 1024, 4096):
    wl = workloads.talos_c3(B, seed=3)
    row = {"batch": B}
    for tag, env in (("tail_from_start_ms", None), ("solve_kernel_first_ms", "1")):
        if env: os.environ["LOIKB_NO_DIRECT_TAIL"] = env
        else: os.environ.pop("LOIKB_NO_DIRECT_TAIL", None)
        s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
        s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        best = 1e9
        for _ in range(5):
            t = time.perf_counter(); s.Solve(); best = min(best, time.perf_counter() - t)
        row[tag] = round(best * 1e3, 3)
        row["max_iterations"] = int(s.get("iter").max()); row["mean_iterations"] = float(s.get("iter").mean())
        s.close()
    print(json.dumps(row), flush=True)
