"""whole-body workload (talos44, four tasks), a few solves: per-solve times (for A/B runs inside ONE gpurun call)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wl = workloads.talos_wholebody(B)
s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
rows = []
for i in range(n):
    s.Solve()
    st = s.stats()
    rows.append((st["total_ms"], st["tail_ms"], st["hslots_ms"]))
    if os.environ.get("VERBOSE"): print("   solve %d: total %.2f  launch %.2f  slots %.2f  built (cumulative) %d  requeues %d  ordered %d" % (i, st["total_ms"], st["tail_ms"], st["hslots_ms"], st.get("flat_built", -1), st["lean_requeues"], st["flat_ordered"]))
r = np.array(rows[2:])
conv = s.get("converged").astype(bool)
print("%s whole body B=%d: total %.2f ms  launch %.2f ms  slots %.2f ms  (min %.2f)  iters %d  %.3f M solves/s  flat %d" % (
    os.environ.get("TAG", ""), B, r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), r[:, 0].min(), st["instance_iterations"],
    conv.sum() / r[:, 0].mean() / 1e3, st["flat_launches"]))
