"""Talos-32 in fp32 at tol 1e-3 (the fp32 contract's tolerance): k_lean (the engine fp32 handles run on) against k_flat<float>
(LOIKB_FLAT_F32=1, an experiment), both against the fp64 result of the same handle type"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = workloads.talos_c3(B)
m = wl["model"]
for tol in (1e-3, 1e-4):
    prm = dict(wl["params"], tol_abs=tol)
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    ref_z = ref_c = None
    for name, prec, flags, env in (("fp64", capi.F64, 0, None), ("fp32 k_lean", capi.F32, 0, None), ("fp32 k_lean accurate", capi.F32, capi.OPT_F32_ACCURATE, None),
                                   ("fp32 k_flat", capi.F32, 0, "1"), ("fp32 k_flat accurate", capi.F32, capi.OPT_F32_ACCURATE, "1")):
        os.environ.pop("LOIKB_FLAT_F32", None)
        if env: os.environ["LOIKB_FLAT_F32"] = env
        s = loik_amd.BatchedLoik(m, B, precision=prec, flags=flags, **prm)
        s.SolveInit(*args); s.Solve(); s.Solve(); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): s.Solve()
        s.synchronize()
        dt = (time.perf_counter() - t0) / 3
        z, c, it = s.get("z"), s.get("converged").astype(bool), s.get("iter")
        st = s.stats()
        if ref_z is None: ref_z, ref_c = z, c
        both = c & ref_c
        d = np.abs(z - ref_z).max(axis=1)[both]
        print("tol %.0e %-22s %.2f ms %.2f M solves/s conv %.4f iters %.1f | vs fp64: dz median %.1e p99 %.1e max %.1e | flat %d lean %d escaped %d" % (
            tol, name, dt * 1e3, c.sum() / dt / 1e6, c.mean(), it.mean(), np.median(d), np.quantile(d, 0.99), d.max(), st["flat_launches"], st["lean_launches"], st["lean_escaped"]))
        s.close()
