import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from loik_amd import workloads
from oracle import ref
from helpers import problem_args
B=8192
wl=workloads.talos_c3(B)
prm=dict(wl["params"])
it=np.load('/tmp/it_full.npy').astype(int)
surv=np.flatnonzero(it>32)
P=np.zeros((len(surv),1000),np.float32); D=np.zeros_like(P); M=np.zeros_like(P)
for n,b in enumerate(surv):
    r=ref.RefSolver(wl["model"],**prm)
    r.Solve(*problem_args(wl,int(b)))
    a=r.solver_info(2); P[n,:len(a)]=a
    a=r.solver_info(5); D[n,:len(a)]=a
    a=r.solver_info(6); M[n,:len(a)]=a
    if n<2: print(len(a), it[b], [r.solver_info(l)[:3] for l in range(9)])
np.save('/tmp/traj_P.npy',P); np.save('/tmp/traj_D.npy',D); np.save('/tmp/traj_M.npy',M); np.save('/tmp/traj_S.npy',surv)
