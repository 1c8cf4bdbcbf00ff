"""floating-base Talos (free-flyer root_joint + 32 revolute joints, nv = 38): the multi-DoF row of SURVEY.md 8(f),
same task / tolerances / batch as the headline configuration"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loik_amd
from loik_amd import workloads
from loik_amd.workloads import FIXTURE_PARAMS, make_workload

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = loik_amd.builtin_model("talos32_freeflyer")
link = m.getJointId("arm_left_7_joint")
wl = make_workload(m, B, link, 0x101C + 7, bound=0.5, snap_prob=0.0)
prm = dict(FIXTURE_PARAMS, max_iter=1000, tol_abs=1e-6, tol_rel=0.0)
s = loik_amd.BatchedLoik(m, B, **prm)
s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
best = 1e9
for _ in range(4):
    t = time.perf_counter(); s.Solve(); best = min(best, time.perf_counter() - t)
st = s.stats()
conv = s.get("converged").astype(bool)
print(json.dumps(dict(config="talos32_freeflyer B=%d tol1e-6 fp64" % B, device_joints=m.nv, ms_per_solve=best * 1e3,
                      solves_per_s=float(conv.sum() / best), inst_iter_per_s=st["instance_iterations"] / best,
                      converged_fraction=float(conv.mean()), primal_infeasible_fraction=float(s.get("primal_infeasible").mean()),
                      mean_iters=float(s.get("iter").mean()), launches=st["launches"], team=st["team"], chunks=st["chunks"],
                      tail_instances=st["tail_instances"], solve_busy_ms=st["solve_busy_ms"], tail_busy_ms=st["tail_busy_ms"])))
s.close()
