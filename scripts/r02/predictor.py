"""How well does the state after Q iterations predict a long runner?  (headline workload)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = 65536
wl = workloads.talos_c3(B)
m, prm = wl["model"], wl["params"]
args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
s = loik_amd.BatchedLoik(m, B, **prm)
s.Solve(*args)
it = s.get("iter"); flips = s.get("mu_updates"); conv = s.get("converged").astype(bool)
print("full: mean it %.1f  median %d  p90 %d p99 %d  max %d; >=999: %d" % (it.mean(), np.median(it), np.quantile(it, .9), np.quantile(it, .99), it.max(), (it >= 999).sum()))
hist = np.histogram(it, bins=[0, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 998, 1001])
print("hist", list(zip(hist[1][:-1], hist[0])))
print("work share by bucket", [(int(lo), round(float(it[(it >= lo) & (it < hi)].sum()) / it.sum(), 3)) for lo, hi in zip(hist[1][:-1], hist[1][1:])])
s.close()
for Q in (16, 24, 32, 48, 64):
    s = loik_amd.BatchedLoik(m, B, **dict(prm, max_iter=Q + 1))
    s.Solve(*args)
    itq = s.get("iter"); fq = s.get("mu_updates"); mu = s.get("mu"); st = s.get("status")
    alive = (st & 8 == 0) | ((itq >= Q) & ~(s.get("converged").astype(bool)) & ~(s.get("primal_infeasible").astype(bool)))
    alive = (itq >= Q) & ~s.get("converged").astype(bool) & ~s.get("primal_infeasible").astype(bool)
    rem = it[alive] - Q
    print("Q=%d: alive %d (%.1f%%), remaining work %.3g inst-it (%.1f%% of total); long runners (>=999) among alive: %d" % (
        Q, alive.sum(), 100 * alive.mean(), rem.sum(), 100 * rem.sum() / it.sum(), (it[alive] >= 999).sum()))
    # predictor: flips in the first Q iterations
    f = fq[alive]
    order = np.argsort(-f, kind="stable")
    long_ = it[alive][order] >= 999
    n_long = long_.sum()
    pos = np.flatnonzero(long_)
    print("   sorted by flips desc: long runners sit at ranks: median %d, p90 %d, max %d of %d" % (np.median(pos), np.quantile(pos, .9), pos.max(), alive.sum()))
    for thr in range(0, int(f.max()) + 1):
        sel = f >= thr
        if sel.sum() == 0: break
        print("   flips>=%d: %6d instances, contains %4d/%d long; mean remaining %.0f" % (thr, sel.sum(), (it[alive][sel] >= 999).sum(), n_long, rem[sel].mean()))
    # corr of remaining with flips
    print("   corr(rem, flips) = %.3f; corr(rem, log mu) = %.3f" % (np.corrcoef(rem, f)[0, 1], np.corrcoef(rem, np.log10(mu[alive]))[0, 1]))
    s.close()
