"""The headline batch with its inputs on the HOST before and its answer on the host after every solve (what `value` leaves out):
SolveInit from host arrays (transposition + FK), Solve(), z back -- and, for the record, all six result members back."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wls = [workloads.talos_c3(B, seed=900 + k) for k in range(4)]
s = loik_amd.BatchedLoik(wls[0]["model"], B, **wls[0]["params"])
nv, nb = wls[0]["model"].nv, wls[0]["model"].njoints - 1
zbuf = np.zeros((B, nv)); cbuf = np.zeros(B, dtype=np.int32)   # (the caller's own arrays, reused: a fresh 17 MB numpy array per call costs 9 ms of page faults)
rows = []
for k, wl in enumerate(wls):
    t0 = time.perf_counter()
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"]); s.synchronize()
    t1 = time.perf_counter(); s.Solve(); t2 = time.perf_counter()
    s.get("z", out=zbuf); t3 = time.perf_counter()
    s.get("converged", out=cbuf); conv = int(cbuf.sum()); t4 = time.perf_counter()
    r = s.get_results(); t5 = time.perf_counter()
    rows.append(dict(solve_init_ms=(t1 - t0) * 1e3, solve_ms=(t2 - t1) * 1e3, z_ms=(t3 - t2) * 1e3, flags_ms=(t4 - t3) * 1e3, all_results_ms=(t5 - t4) * 1e3, converged=conv))
r = rows[1:]   # (the first batch of a handle allocates)
m = {k: float(np.mean([x[k] for x in r])) for k in r[0]}
m["batch"] = B
m["host_to_host_ms_z_and_flags"] = m["solve_init_ms"] + m["solve_ms"] + m["z_ms"] + m["flags_ms"]
m["pcie_inclusive_solves_per_s"] = m["converged"] / (m["host_to_host_ms_z_and_flags"] * 1e-3)
m["resident_solves_per_s"] = m["converged"] / (m["solve_ms"] * 1e-3)
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in m.items()}))
