"""C5 trade-off table against the fp64 oracle: Panda-7, B = 65536, tol 1e-3 / 1e-4, fp32 fast / accurate; where the large
deviations come from (iterations, mu of the instance)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads
from oracle import ref
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for tol in (1e-3, 1e-4):
    wl = workloads.panda_c5(B, tol=tol)
    m, prm = wl["model"], wl["params"]
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    out = ref.solve_batch(m, *args, nthreads=16, **prm)
    for name, prec, flags in (("fp64", capi.F64, 0), ("fp32 fast", capi.F32, 0), ("fp32 accurate", capi.F32, capi.OPT_F32_ACCURATE)):
        s = loik_amd.BatchedLoik(m, B, precision=prec, flags=flags, **prm)
        s.SolveInit(*args); s.Solve(); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): s.Solve()
        s.synchronize()
        dt = (time.perf_counter() - t0) / 3
        z, c = s.get("z"), s.get("converged").astype(bool)
        it, mu = s.get("iter"), s.get("mu")
        both = c & out["converged"]
        dz = np.abs(z - out["z"]).max(axis=1)
        d = dz[both]
        big = both & (dz > tol)
        print("tol %.0e %-14s %.3f ms %.2f M solves/s  conv %.4f (oracle %.4f) mismatch %.4f | dz median %.2e p90 %.2e p99 %.2e max %.2e | iters %.1f (oracle %.1f) | big: n=%d iters %.1f mu median %.1e (all: %.1e)" % (
            tol, name, dt * 1e3, c.sum() / dt / 1e6, c.mean(), out["converged"].mean(), (c != out["converged"]).mean(),
            np.median(d), np.quantile(d, 0.9), np.quantile(d, 0.99), d.max(), it.mean(), out["iters"].mean(),
            big.sum(), it[big].mean() if big.any() else 0, np.median(mu[big]) if big.any() else 0, np.median(mu[both])))
        s.close()
