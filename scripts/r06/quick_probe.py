"""headline batch in arrival order (LOIKB_FLAT_ORDER=0: every solve as a handle's first), a few solves: the two-launch schedule against the
single time-sliced launch.  usage: quick_probe.py B n   (LOIKB_FLAT_PROBE / _MARK from the environment)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
os.environ.setdefault("LOIKB_FLAT_ORDER", "0")
wl = workloads.talos_c3(B)
s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
rows = []
for i in range(n):
    s.Solve()
    st = s.stats()
    rows.append((st["total_ms"], st["tail_ms"], st["hslots_ms"], st["probe_ms"], st["queue_dry_ms"]))
r = np.array(rows[2:])
print("%-28s B=%6d: total %.2f ms (min %.2f)  on-chip %.2f  slots %.2f  probe+sort %.2f  finish %.2f  | queue dry at %.2f  requeues %d  built %d  probe launches %d" % (
    os.environ.get("TAG", ""), B, r[:, 0].mean(), r[:, 0].min(), r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean(), (r[:, 1] - r[:, 2] - r[:, 3]).mean(),
    r[:, 4].mean(), st["lean_requeues"], st["flat_built"], st["flat_probe_launches"]), flush=True)
