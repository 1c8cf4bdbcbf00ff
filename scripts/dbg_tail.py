import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import loik_amd
from loik_amd import capi
from oracle import ref
from helpers import FIXTURE, feasible_batch, problem_args
m = loik_amd.builtin_model("panda7")
wl = feasible_batch(m, 4, m.njoints - 1, 21, nu_scale=0.5)
prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
def run(**kw):
    s = loik_amd.BatchedLoik(m, 4, **prm, **kw)
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    return s
a = run(flags=capi.OPT_NO_H_CACHE, tail_max_instances=-1)
b = run(flags=capi.OPT_NO_H_CACHE)
print("iters", a.get("iter"), b.get("iter"), b.stats())
for name in ["z", "vis", "fis", "pis", "r", "UDinv", "Dinv", "His", "w", "g"]:
    print(name, np.abs(a.get(name) - b.get(name)).max())
r = ref.RefSolver(m, **prm); r.Solve(*problem_args(wl, 0))
print("oracle iter", r.get_iter(), "pis diff a", np.abs(a.get("pis")[0] - r.pis[1:]).max(), "b", np.abs(b.get("pis")[0] - r.pis[1:]).max())
print(a.get("pis")[0][:2]); print(b.get("pis")[0][:2]); print(r.pis[1:3])
