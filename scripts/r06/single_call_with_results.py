"""What a drop-in caller pays for ONE problem per call: Solve() AND the members the reference's data object holds afterwards (z, nu, w, vis, fis,
yis -- what include/loik_amd/loik.hpp copies after every solve).  usage: single_call_with_results.py [B ...]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
FIELDS = ("z", "nu", "w", "vis", "fis", "yis")
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 64]:
    wl = workloads.talos_c3(B, seed=3)
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    t_solve, t_get, t_many = [], [], []
    has_many = hasattr(s, "get_results")
    for _ in range(30):
        t0 = time.perf_counter(); s.Solve(); t1 = time.perf_counter()
        r = {k: s.get(k) for k in FIELDS}; t2 = time.perf_counter()
        t_solve.append(t1 - t0); t_get.append(t2 - t1)
        if has_many:
            t3 = time.perf_counter(); r2 = s.get_results(); t_many.append(time.perf_counter() - t3)
            assert all(np.array_equal(np.asarray(r[k]), np.asarray(r2[k])) for k in FIELDS)
    print(json.dumps({"batch": B, "solve_ms": round(min(t_solve) * 1e3, 4), "six_gets_ms": round(min(t_get) * 1e3, 4),
                      "get_results_ms": round(min(t_many) * 1e3, 4) if has_many else None,
                      "solve_plus_results_ms": round((min(t_solve) + (min(t_many) if has_many else min(t_get))) * 1e3, 4)}), flush=True)
    s.close()
