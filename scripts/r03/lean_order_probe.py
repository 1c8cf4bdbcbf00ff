"""k_lean (LOIKB_FLAT=0) on the headline batch: arrival order against longest first from the handle's previous solve"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
os.environ["LOIKB_FLAT"] = "0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = workloads.talos_c3(B)
for order in ("0", "1"):
    os.environ["LOIKB_FLAT_ORDER"] = order
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    rows = []
    for i in range(6):
        s.Solve(); st = s.stats(); rows.append((st["total_ms"], st["flat_ordered"], st["lean_launches"], st["flat_launches"]))
    print("k_lean order=%s: per solve %s  (lean %d flat %d)" % (order, " ".join("%.2f%s" % (r[0], "o" if r[1] else "") for r in rows), rows[-1][2], rows[-1][3]))
    z = s.get("z"); it = s.get("iter")
    if order == "0": z0, it0 = z, it
    else: print("   bit-identical to arrival order:", bool(np.array_equal(z, z0) and np.array_equal(it, it0)))
    s.close()
