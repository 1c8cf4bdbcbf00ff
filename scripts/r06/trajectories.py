"""Residual trajectories of the headline workload's instances through the CPU oracle (test infrastructure; a scheduling study input).
Writes /tmp/traj6.npz: it (iterations), conv, pinf, and for the instances with it > 16: P, D, M [n][1000] float32 (primal, dual, mu)."""
import sys, numpy as np, multiprocessing as mp
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from loik_amd import workloads
from oracle import ref
from helpers import problem_args
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
wl = workloads.talos_c3(B)
prm = dict(wl["params"])

def work(rng):
    out = []
    for b in rng:
        r = ref.RefSolver(wl["model"], **prm)
        r.Solve(*problem_args(wl, int(b)))
        p = np.asarray(r.solver_info(2), np.float32); d = np.asarray(r.solver_info(5), np.float32); m = np.asarray(r.solver_info(6), np.float32)
        out.append((b, int(r.scalar("iter")), int(r.scalar("converged")), int(r.scalar("primal_infeasible")), p, d, m))
    return out

if __name__ == '__main__':
    chunks = np.array_split(np.arange(B), 64)
    with mp.Pool(8) as pool:
        res = pool.map(work, chunks)
    it = np.zeros(B, int); conv = np.zeros(B, int); pinf = np.zeros(B, int)
    P = np.zeros((B, 1000), np.float32); D = np.zeros_like(P); M = np.zeros_like(P)
    for ch in res:
        for b, i, c, f, p, d, m in ch:
            it[b] = i; conv[b] = c; pinf[b] = f
            P[b, :len(p)] = p; D[b, :len(d)] = d; M[b, :len(m)] = m
    np.savez_compressed('/tmp/traj6.npz', it=it, conv=conv, pinf=pinf, P=P, D=D, M=M)
    print("mean", it.mean(), "max", it.max(), "conv", conv.mean(), "pinf", pinf.mean())
