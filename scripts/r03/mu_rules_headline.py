"""the three penalty rules on the headline workload (Talos-32, B = 65536): DEFAULT (the reference's), MAXEIGENVALUE and OSQP
(extensions of this library; upstream declares both and throws): time, solves/s, converged / flagged / at max_iter"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for wname, wl in (("talos32 C3", workloads.talos_c3(B)), ("talos44 whole body", workloads.talos_wholebody(B))):
    for name, strat in (("DEFAULT", 0), ("MAXEIGENVALUE", 3), ("OSQP", 1)):
        prm = dict(wl["params"], mu_update_strat=strat)
        s = loik_amd.BatchedLoik(wl["model"], B, **prm)
        s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        s.Solve(); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): s.Solve()
        s.synchronize()
        dt = (time.perf_counter() - t0) / 3
        conv, inf, it = s.get("converged").astype(bool), s.get("primal_infeasible").astype(bool), s.get("iter")
        st = s.stats()
        print(json.dumps({"workload": wname, "rule": name, "ms_per_batch": dt * 1e3, "solves_per_s": conv.sum() / dt, "converged": float(conv.mean()),
                          "flagged_infeasible": float(inf.mean()), "at_max_iter": float((it >= prm["max_iter"] - 1).mean()),
                          "mean_iterations": float(it.mean()), "engine": "flat" if st["flat_launches"] else "lean" if st["lean_launches"] else "k_solve+k_tail"}), flush=True)
        s.close()
