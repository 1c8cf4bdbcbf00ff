"""headline batch with per-link reference costs (UpdateReferences: one SPD weight + one target per link): the flat engine's
per-link instantiation (HM = 3) against k_lean's (LOIKB_FLAT=0), in one process"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = workloads.talos_c3(B)
nj = wl["model"].njoints
rng = np.random.default_rng(5)
H = np.zeros((nj, 6, 6)); v = np.zeros((nj, 6))
for i in range(nj):
    M = rng.normal(size=(6, 6)) * 0.2
    H[i] = np.eye(6) * rng.uniform(0.5, 2.0) + M @ M.T
    v[i] = rng.normal(size=6) * 0.05
for tag, env in (("flat (HM=3)", None), ("k_lean per link", "0")):
    if env is None: os.environ.pop("LOIKB_FLAT", None)
    else: os.environ["LOIKB_FLAT"] = env
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    s.UpdateReferences(H, v)
    rows = []
    for i in range(6):
        s.Solve()
        st = s.stats()
        rows.append((st["total_ms"], st["tail_ms"], st["hslots_ms"]))
    r = np.array(rows[2:])
    print("%-16s B=%d: total %.2f ms  launch %.2f ms  slots %.2f ms  converged %.4f  iters %d  flat %d lean %d" % (
        tag, B, r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), s.get_convergence_status().mean(), st["instance_iterations"],
        st["flat_launches"], st["lean_launches"]))
    s.close()
