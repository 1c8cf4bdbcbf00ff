"""max(primal, dual) of the headline batch's instances at a few marks, and their iteration counts, through the CPU oracle (test
infrastructure; input of the scheduling study).  Writes /tmp/marks6_<B>.npz."""
import sys, numpy as np, multiprocessing as mp
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from loik_amd import workloads
from oracle import ref
from helpers import problem_args
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else None
MARKS = np.array([8, 16, 24, 32, 48, 64, 80, 96, 112, 128, 160, 192, 224, 256, 288, 320, 384, 448, 512])
wl = workloads.talos_c3(B) if SEED is None else workloads.talos_c3(B, seed=SEED)
prm = dict(wl["params"])

def work(rng):
    out = []
    for b in rng:
        r = ref.RefSolver(wl["model"], **prm)
        r.Solve(*problem_args(wl, int(b)))
        p = np.asarray(r.solver_info(2)); d = np.asarray(r.solver_info(5)); m = np.asarray(r.solver_info(6))
        n = len(p)
        R = np.zeros(len(MARKS), np.float32); Pm = np.zeros_like(R); Dm = np.zeros_like(R); Mm = np.zeros_like(R)
        for k, mk in enumerate(MARKS):
            if mk <= n: Pm[k] = p[mk - 1]; Dm[k] = d[mk - 1]; Mm[k] = m[mk - 1]
        out.append((b, int(r.scalar("iter")), int(r.scalar("converged")), int(r.scalar("primal_infeasible")), Pm, Dm, Mm))
    return out

if __name__ == '__main__':
    chunks = np.array_split(np.arange(B), 256)
    with mp.Pool(8) as pool:
        res = pool.map(work, chunks)
    it = np.zeros(B, int); conv = np.zeros(B, int); pinf = np.zeros(B, int)
    P = np.zeros((B, len(MARKS)), np.float32); D = np.zeros_like(P); M = np.zeros_like(P)
    for ch in res:
        for b, i, c, f, p, d, m in ch:
            it[b] = i; conv[b] = c; pinf[b] = f; P[b] = p; D[b] = d; M[b] = m
    np.savez_compressed('/tmp/marks6_%d%s.npz' % (B, "" if SEED is None else "_s%d" % SEED), it=it, conv=conv, pinf=pinf, P=P, D=D, M=M, marks=MARKS)
    print("mean", it.mean(), "max", it.max(), "conv", conv.mean(), "pinf", pinf.mean())
