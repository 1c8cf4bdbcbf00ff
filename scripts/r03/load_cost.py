"""what loading + storing an instance costs the flat engine: the headline batch with max_iter = 2 / 3 / 5 / 9 (every instance does
exactly 1 / 2 / 4 / 8 iterations: tol_abs = 0): launch time against the iteration count -> intercept = load + store"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = workloads.talos_c3(B)
rows = []
for k in (1, 2, 4, 8, 16):
    prm = dict(wl["params"], max_iter=k + 1, tol_abs=0.0, tol_primal_inf=0.0)
    s = loik_amd.BatchedLoik(wl["model"], B, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    t = []
    for _ in range(4):
        s.Solve(); st = s.stats(); t.append(st["tail_ms"] - st["hslots_ms"])
    rows.append((k, np.mean(t[1:]), st["hslots_ms"]))
    s.close()
ks = np.array([r[0] for r in rows], float); ts = np.array([r[1] for r in rows])
a, b = np.polyfit(ks, ts, 1)
waves = 2048
print("B=%d: launch ms by iterations %s; slots %.2f ms" % (B, ["%d: %.3f" % (r[0], r[1]) for r in rows], rows[0][2]))
print("fit: %.3f ms per iteration of the batch + %.3f ms -> per instance on %d resident wavefronts: %.2f us per iteration, %.1f us load + store" % (
    a, b, waves, a * 1e3 * waves / B, b * 1e3 * waves / B))
