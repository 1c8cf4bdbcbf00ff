"""throughput of the cooperative tail kernel: every instance runs exactly `iters` ADMM iterations (tol = 0)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads
iters = 200
for B in [int(x) for x in sys.argv[1:]] or [64, 1024, 2048, 4096, 8192, 16384]:
    wl = workloads.talos_c3(B, seed=int(os.environ.get("MB_SEED", "5")))
    prm = dict(wl["params"], max_iter=iters + 1, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    s = loik_amd.BatchedLoik(wl["model"], B, max_launch_iters=1, tail_max_instances=1 << 24, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    best = 1e9
    for _ in range(3):
        s.Solve(); st = s.stats(); best = min(best, st["tail_ms"])
    n = st["tail_instances"] * (iters - 1)
    print("B=%6d: tail %.2f ms for %d inst x %d iters -> %.2f us per iteration-round, %.1f M inst-it/s" % (
        B, best, st["tail_instances"], iters - 1, best * 1e3 / (iters - 1), n / best / 1e3))
    s.close()
