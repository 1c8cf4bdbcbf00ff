"""micro-benchmarks of k_solve in fixed-iteration mode (no stopping logic): bulk rate vs single-wave latency"""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loik_amd
from loik_amd import capi, workloads

def run(B, iters, flags, reps=3, model=None):
    wl = workloads.talos_c3(B, seed=5)
    prm = dict(wl["params"], max_iter=iters + 1)
    s = loik_amd.BatchedLoik(wl["model"], B, flags=capi.OPT_FIXED_ITERS | flags, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    best = 1e9
    for _ in range(reps):
        s.Solve()
        st = s.stats()
        best = min(best, st["kernel_ms"])
    ii = st["instance_iterations"]
    print("B=%6d iters=%4d flags=%d: kernel %.2f ms -> %.1f us/iteration, %.1f M inst-it/s, algorithmic %.0f GB/s (%.3f of 8 TB/s)" % (
        B, iters, flags, best, best * 1e3 / iters, ii / best / 1e3, ii * st["bytes_per_instance_iteration"] / best / 1e6,
        ii * st["bytes_per_instance_iteration"] / best / 1e6 / 8000))
    s.close()

if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1:]] or [64, 256, 4096, 16384, 65536, 262144]
    for B in sizes:
        for flags in (0, capi.OPT_NO_H_CACHE):
            run(B, 100 if B <= 65536 else 30, flags)
