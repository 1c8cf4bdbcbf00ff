"""k_tail as the engine of a whole batch (OSQP penalty rule, 32 768 Talos instances): arrival order against longest first"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
wl = workloads.talos_c3(B)
prm = dict(wl["params"], mu_update_strat=1)
for order in ("0", "1"):
    os.environ["LOIKB_FLAT_ORDER"] = order
    s = loik_amd.BatchedLoik(wl["model"], B, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    rows = []
    for i in range(6):
        s.Solve(); st = s.stats(); rows.append((st["total_ms"], st["flat_ordered"], st["tail_launches"], st["launches"]))
    print("OSQP B=%d order=%s: per solve %s  (%s)" % (B, order, " ".join("%.2f%s" % (r[0], "o" if r[1] else "") for r in rows), s.plan()[:60]))
    z = s.get("z"); it = s.get("iter")
    if order == "0": z0, it0 = z, it
    else: print("   bit-identical to arrival order:", bool(np.array_equal(z, z0) and np.array_equal(it, it0)))
    s.close()
