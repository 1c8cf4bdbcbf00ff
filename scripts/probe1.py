import numpy as np, sys, time
sys.path.insert(0,'.')
import loik_amd
from loik_amd import workloads
from oracle import ref
print("devices", loik_amd.device_count())
wl = workloads.talos_c3(512, seed=11)
m, prm = wl["model"], dict(wl["params"])
# per-iteration parity: k iterations, no stopping (tol=0 never converges; infeasibility off)
for k in (1,2,3,5,10):
    p = dict(prm); p.update(max_iter=k+1, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    s = loik_amd.BatchedLoik(m, 512, **p)
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    worst = {}
    for b in range(0, 512, 37):
        r = ref.RefSolver(m, **p)
        r.Solve(wl["q"][b], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"][b], wl["lb"], wl["ub"])
        for name, gv, rv in [("z", s.get("z")[b], r.z), ("nu", s.get("nu")[b], r.nu), ("w", s.get("w")[b], r.w),
                          ("vis", s.get("vis")[b], r.vis[1:]), ("fis", s.get("fis")[b], r.fis[1:]),
                          ("g", s.get("g")[b], r.g[1:]), ("yis", s.get("yis")[b], r.yis), ("pis", s.get("pis")[b], r.pis[1:]),
                          ("His", s.His_full()[b], r.His[1:]), ("liMi", s.get("liMi")[b], r.liMi[1:]),
                          ("pres", s.get("primal_residual")[b], r.scalar("primal_residual")), ("dres", s.get("dual_residual")[b], r.scalar("dual_residual")),
                          ("mu", s.get("mu")[b], r.scalar("mu")), ("iter", s.get("iter")[b], r.get_iter())]:
            d = np.abs(np.asarray(gv,dtype=float) - np.asarray(rv,dtype=float)); den = np.maximum(np.abs(np.asarray(rv,dtype=float)), 1e-300)
            e = float(np.minimum(d, d/den).max())
            worst[name] = max(worst.get(name, 0.0), e)
    print("k=%d" % k, " ".join("%s=%.1e" % kv for kv in worst.items()))
    s.close()
import __graft_entry__ as g
g.smoke()
# timing
for B in (4096, 65536):
    wl = workloads.talos_c3(B, seed=3)
    s = loik_amd.BatchedLoik(m, B, **wl["params"])
    for rep in range(2):
        t = time.time()
        s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        dt = time.time() - t
        st = s.stats()
        it = s.get("iter")
        print("B=%d wall %.1f ms kernel %.1f ms inst-iters %d -> %.1f M inst-it/s (kernel); conv %.3f inf %.3f; iters mean %.1f max %d; alg GB/s %.1f" % (
            B, dt*1e3, st["kernel_ms"], st["instance_iterations"], st["instance_iterations"]/st["kernel_ms"]/1e3, s.get("converged").mean(), s.get("primal_infeasible").mean(), it.mean(), it.max(),
            st["instance_iterations"]*st["bytes_per_instance_iteration"]/st["kernel_ms"]/1e6))
    s.close()
