import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import loik_amd
from helpers import FIXTURE, feasible_batch
talos = loik_amd.builtin_model("talos32")
link = talos.getJointId("arm_left_7_joint")
B = 5000
wl = feasible_batch(talos, B, link, 555, nu_scale=0.5)
prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0)
def run(rnd):
    os.environ["LOIKB_TAIL_ROUND"] = str(rnd)
    s = loik_amd.BatchedLoik(talos, B, max_launch_iters=2, tail_max_instances=1 << 20, **prm)
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    return s
a, b = run(100000), run(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
ia, ib = a.get("iter"), b.get("iter")
print("iter equal:", (ia == ib).mean(), "n diff", (ia != ib).sum())
for name in ["z", "nu", "w", "mu", "primal_residual"]:
    x, y = a.get(name), b.get(name)
    d = np.abs(x - y).reshape(B, -1).max(axis=1)
    bad = np.flatnonzero(d > 0)
    print(name, "n bad", len(bad), "max", d.max(), "iters of bad", ia[bad][:10], ib[bad][:10])
print(a.stats()); print(b.stats())
