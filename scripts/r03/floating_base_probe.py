import sys; sys.path.insert(0,"/root/repo")
import loik_amd, numpy as np
from loik_amd import workloads
m = loik_amd.builtin_model("talos32_freeflyer")
B=65536
wl = workloads.make_workload(m, B, m.getJointId("arm_left_7_joint"), 7, bound=0.5, snap_prob=0.0)
prm = dict(workloads.FIXTURE_PARAMS, max_iter=1000, tol_abs=1e-6, tol_rel=0.0)
s = loik_amd.BatchedLoik(m, B, **prm)
s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
for i in range(4):
    s.Solve(); st=s.stats(); print(round(st["total_ms"],2), "slots %.2f" % st["hslots_ms"], st["flat_launches"], st["flat_ordered"], st["lean_launches"])
print(s.plan())
