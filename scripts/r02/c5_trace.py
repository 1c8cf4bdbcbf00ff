import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd.workloads import make_workload, FIXTURE_PARAMS
m = loik_amd.builtin_model("panda7")
wl = make_workload(m, 65536, m.njoints - 1, 5, bound=2.0, snap_prob=0.0, nu_scale=0.5)
prm = dict(FIXTURE_PARAMS, max_iter=200, tol_abs=1e-3, tol_rel=0.0)
s = loik_amd.BatchedLoik(m, 65536, **prm)
print(s.plan())
s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
print(s.plan())
for _ in range(3):
    t = time.perf_counter(); s.Solve(); dt = time.perf_counter() - t
    st = s.stats()
    print("wall %.3f ms  total_ms %.3f kernel_ms %.3f launches %d chunks %d tail_inst %d" % (dt * 1e3, st["total_ms"], st["kernel_ms"], st["launches"], st["chunks"], st["tail_instances"]))
