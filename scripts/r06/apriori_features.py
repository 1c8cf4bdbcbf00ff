"""Do a-priori features of the problem data (the task twist b, the Jacobian of the task link) predict the ADMM iteration count?  (CPU oracle;
scheduling study of round 6, last session: no -- the best, |b|, has rank correlation 0.50 and leaves long runners in the last 5 % of the batch.)
usage: apriori_features.py [B]  -> /tmp/study/apriori_<B>.npz and the table"""
import sys, numpy as np, multiprocessing as mp
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from loik_amd import workloads
from oracle import ref
from helpers import problem_args
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
wl = workloads.talos_c3(B)
prm = dict(wl["params"])
def work(rng):
    out = []
    r = None
    for b in rng:
        r = ref.RefSolver(wl["model"], **prm)
        r.Solve(*problem_args(wl, int(b)))
        out.append((b, int(r.scalar("iter")), int(r.scalar("converged")), int(r.scalar("primal_infeasible"))))
    return out
def features(path):
    """second half (run after the first wrote its .npz): do features of the PROBLEM DATA rank the long runners first?"""
    d = np.load(path)
    it, J, b = d['it'], d['J'], d['b']
    B = len(it)
    cols = np.where(np.abs(J).sum((0, 1)) > 0)[0]
    Jc = J[:, :, cols]
    nu_ls = np.einsum('bij,bj->bi', np.linalg.pinv(Jc), b)
    sv = np.linalg.svd(Jc, compute_uv=False)
    long = it >= 500
    for name, f in [("max|J^+ b| / bound", np.abs(nu_ls).max(1) / 0.5), ("1 / sigma_min(J)", 1.0 / sv[:, -1]), ("|b|", np.linalg.norm(b, axis=1)),
                    ("box violation of J^+ b", np.clip(np.abs(nu_ls) - 0.5, 0, None).sum(1))]:
        rank = np.empty(B, int); rank[np.argsort(-f)] = np.arange(B)
        r = rank[long]
        print("%-24s rank correlation with the count %.3f; the %d instances of >= 500 iterations: median rank %.3f, 90 %% %.3f, 99 %% %.3f, last %.3f of the batch" % (
            name, np.corrcoef(np.argsort(np.argsort(f)), np.argsort(np.argsort(it)))[0, 1], long.sum(), np.median(r) / B, np.quantile(r, .9) / B, np.quantile(r, .99) / B, r.max() / B))


if __name__ == '__main__':
    chunks = np.array_split(np.arange(B), 128)
    with mp.Pool(16) as pool:
        res = pool.map(work, chunks)
    it = np.zeros(B, int); conv = np.zeros(B, int); pinf = np.zeros(B, int)
    for ch in res:
        for b, i, c, f in ch: it[b] = i; conv[b] = c; pinf[b] = f
    model = wl["model"]; link = int(wl["c_ids"][0]); nv = model.nv
    q = wl["q"]
    J = np.zeros((B, 6, nv))
    for k in range(nv):
        e = np.zeros((B, nv)); e[:, k] = 1.0
        J[:, :, k] = workloads.link_velocity(model, q, e, link)
    b = wl["bis"].reshape(B, 6)
    np.savez_compressed('/tmp/study/apriori_%d.npz' % B, it=it, conv=conv, pinf=pinf, J=J, b=b, nu_star=wl["nu_star"])
    print("mean", it.mean(), "max", it.max(), "conv", conv.mean(), "pinf", pinf.mean(), "n>=500", (it >= 500).sum(), "n>=900", (it >= 900).sum())
    features('/tmp/study/apriori_%d.npz' % B)
