import sys, numpy as np
sys.path.insert(0,'/root/repo')
from loik_amd import workloads
from oracle import ref
B=8192
wl=workloads.talos_c3(B)
prm=dict(wl["params"])
print({k:prm[k] for k in ("max_iter","tol_abs","tol_rel")})
args=(wl["model"], wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
full=ref.solve_batch(*args, nthreads=16, want_nu=False, **prm)
it=full["iters"]; conv=full["converged"]
print("mean",it.mean(),"max",it.max(),"conv",conv.mean(), "inf", full["primal_infeasible"].mean())
print("hist", np.percentile(it,[10,25,50,75,90,95,97,98,99,99.5]))
for thr in (32,64,100,150,200,300,400,600,800): print(thr, "survive", (it>thr).mean(), "work below", np.minimum(it,thr).sum()/it.sum())
np.save('/tmp/it_full.npy', it)
for k in (8,16,32,64):
    p=dict(prm, max_iter=k+1, tol_abs=0.0, tol_primal_inf=0.0)
    o=ref.solve_batch(*args, nthreads=16, want_nu=False, **p)
    np.save('/tmp/res_%d.npy'%k, np.stack([o["primal_residual"], o["dual_residual"]]))
    print(k, list(o.keys())[:12])
