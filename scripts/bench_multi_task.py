"""several simultaneous task constraints on the Talos headline batch (num_eq_c = 1, 2, 4): the lean tail kernel takes
num_eq_c <= 1; more constraints run in the solve kernel + k_tail"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import loik_amd
from loik_amd.workloads import FIXTURE_PARAMS
from helpers import multi_task_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = loik_amd.builtin_model("talos32")
names = ["arm_left_7_joint", "arm_right_7_joint", "leg_left_6_joint", "leg_right_6_joint"]
for nc in (1, 2, 4):
    links = [m.getJointId(n) for n in names[:nc]]
    wl = multi_task_batch(m, B, links, 5, bound=0.5, nu_scale=0.4)
    wl["Ais"] = np.tile(np.eye(6)[None], (nc, 1, 1)); 
    from loik_amd import workloads
    wl["bis"] = np.stack([workloads.link_velocity(m, wl["q"], wl["nu_star"], l) for l in links], axis=1)
    prm = dict(FIXTURE_PARAMS, num_eq_c=nc, max_iter=1000, tol_abs=1e-6, tol_rel=0.0)
    s = loik_amd.BatchedLoik(m, B, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); s.Solve(); best = min(best, time.perf_counter() - t)
    st = s.stats(); conv = s.get("converged").astype(bool)
    print(json.dumps(dict(config="talos32 B=%d, %d task constraint(s)" % (B, nc), ms_per_solve=best * 1e3, solves_per_s=float(conv.sum() / best),
                          inst_iter_per_s=st["instance_iterations"] / best, converged_fraction=float(conv.mean()), mean_iters=float(s.get("iter").mean()),
                          lean_launches=st["lean_launches"], tail_instances=st["tail_instances"], chunks=st["chunks"])), flush=True)
    s.close()
