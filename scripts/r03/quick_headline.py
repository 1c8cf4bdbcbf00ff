"""headline batch, a few solves: per-solve times of the on-chip launch and of the slot kernels (for A/B runs inside ONE gpurun call)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wl = workloads.talos_c3(B)
s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
rows = []
for i in range(n):
    s.Solve()
    st = s.stats()
    rows.append((st["total_ms"], st["tail_ms"], st["hslots_ms"], st["flat_ordered"]))
r = np.array(rows[2:])
print("   per solve: " + " ".join("%.2f%s" % (x[0], "o" if x[3] else "") for x in rows))
print("%s B=%d: total %.2f ms  on-chip launch %.2f ms  of which slots %.2f ms   (min total %.2f)  iters %d flat %d  requeues %d  queue dry at %.2f ms" % (
    os.environ.get("TAG", ""), B, r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), r[:, 0].min(), st["instance_iterations"], st["flat_launches"],
    st["lean_requeues"], st["queue_dry_ms"]))
