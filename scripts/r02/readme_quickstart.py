import sys; sys.path.insert(0, "/root/repo")
import numpy as np, loik_amd
from loik_amd import workloads
model = loik_amd.builtin_model("talos32")
wl = workloads.talos_c3(4096)
s = loik_amd.BatchedLoik(model, 4096, max_iter=1000, tol_abs=1e-6, tol_rel=0.0, **workloads.FIXTURE_PARAMS)
s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
z, it, ok = s.get("z"), s.get("iter"), s.get_convergence_status()
s.integrate(0.1); s.Solve(None, int(wl["c_ids"][0]), wl["Ais"], wl["bis"][:, 0])
print(z.shape, it.mean(), ok.mean()); print(s.plan())
