"""One problem per call through the three entry points a caller has (Talos-32, the headline's parameters): Solve() on a resident problem, the
full Solve(q, H_ref, v_ref, ids, Ais, bis, lb, ub) = SolveInit + Solve (loik-loid-optimized.hpp:475-580: what IKBench-style callers use), the
tailored Solve(q, c_id, Ai, bi) (:596-695) -- each with the six result members fetched (loikb_get_results)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
for B in [int(a) for a in sys.argv[1:]] or [1, 8]:
    wl = workloads.talos_c3(B, seed=3)
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    s.Solve(*args)
    t = {"plain": [], "full": [], "tailored": [], "solve_init": [], "results": []}
    for _ in range(30):
        t0 = time.perf_counter(); s.Solve(); t["plain"].append(time.perf_counter() - t0)
        t0 = time.perf_counter(); s.Solve(*args); t["full"].append(time.perf_counter() - t0)
        t0 = time.perf_counter(); s.Solve(wl["q"], int(wl["c_ids"][0]), wl["Ais"][0], wl["bis"][:, 0]); t["tailored"].append(time.perf_counter() - t0)
        t0 = time.perf_counter(); s.SolveInit(*args); s.synchronize(); t["solve_init"].append(time.perf_counter() - t0)
        s.Solve()
        t0 = time.perf_counter(); s.get_results(); t["results"].append(time.perf_counter() - t0)
        t0 = time.perf_counter(); s.get_results(s.RESULT_FIELDS); t.setdefault("results_and_scalars", []).append(time.perf_counter() - t0)
        t0 = time.perf_counter(); s.get("iter"); s.get("converged"); s.get("primal_infeasible"); t.setdefault("three_scalar_gets", []).append(time.perf_counter() - t0)
    print(json.dumps(dict(batch=B, iterations=int(np.asarray(s.get("iter")).max()), **{k + "_ms": round(min(v) * 1e3, 4) for k, v in t.items()})), flush=True)
    s.close()
