"""sweep of the hand-over policy (compaction / tail kernel thresholds) on the headline workload"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = workloads.talos_c3(B)
for kw in [dict(tail_max_instances=-1), dict(tail_max_instances=1024), dict(tail_max_instances=2048), dict(tail_max_instances=4096),
           dict(tail_max_instances=8192), dict(tail_max_instances=16384), dict(tail_max_instances=32768),
           dict(tail_max_instances=8192, max_launch_iters=4), dict(tail_max_instances=16384, max_launch_iters=4),
           dict(tail_max_instances=16384, max_launch_iters=16)]:
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"], **kw)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); s.Solve(); dt = time.perf_counter() - t
        best = min(best, dt)
    st = s.stats()
    print(kw, "-> %.1f ms/step; kernel %.1f ms (tail %.1f ms for %d inst), launches %d, compactions %d, %.1f M inst-it/s, solves/s %.0f" % (
        best * 1e3, st["kernel_ms"], st["tail_ms"], st["tail_instances"], st["launches"], st["compactions"],
        st["instance_iterations"] / best / 1e6, s.get("converged").sum() / best))
    s.close()
