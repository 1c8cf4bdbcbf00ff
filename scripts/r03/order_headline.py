"""headline / whole-body batch: solves in arrival order (LOIKB_FLAT_ORDER=0: round-robin time slicing) against longest first from
the handle's previous solve, and the same with the targets re-drawn before every solve (the order is then a prediction)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for robot in ("talos32", "talos44"):
    mk = workloads.talos_c3 if robot == "talos32" else workloads.talos_wholebody
    wl = mk(B)
    for order in ("0", "1"):
        os.environ["LOIKB_FLAT_ORDER"] = order
        s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
        s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        rows = []
        for i in range(7):
            s.Solve()
            st = s.stats()
            rows.append((st["total_ms"], st["tail_ms"], st["hslots_ms"], st["queue_dry_ms"], st["flat_ordered"], st["lean_requeues"]))
        r = np.array(rows)
        print("%s order=%s same batch again: first solve %.2f ms, later %.2f ms (launch %.2f, queue dry at %.2f), ordered %d, requeues %d" % (
            robot, order, r[0, 0], r[2:, 0].mean(), r[2:, 1].mean(), r[2:, 3].mean(), int(r[-1, 4]), int(r[-1, 5])))
        # the targets scaled per instance by U(0.6, 1) before every solve (same configurations): the previous solve's order is
        # then a prediction; and a batch drawn afresh every time: the order is noise
        rng = np.random.default_rng(1)
        for what in ("targets rescaled every solve", "another batch every solve"):
            rows = []
            for i in range(6):
                if what.startswith("targets"):
                    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"] * rng.uniform(0.6, 1.0, size=(B, 1, 1)), wl["lb"], wl["ub"])
                else:
                    w2 = mk(B, seed=100 + i)
                    s.SolveInit(w2["q"], w2["H_ref"], w2["v_ref"], w2["c_ids"], w2["Ais"], w2["bis"], w2["lb"], w2["ub"])
                s.Solve()
                st = s.stats()
                rows.append((st["total_ms"], st["queue_dry_ms"], st["instance_iterations"]))
            r = np.array(rows)
            print("%s order=%s %s: %.2f ms (queue dry at %.2f; %.2f M instance-iterations)" % (robot, order, what, r[1:, 0].mean(), r[1:, 1].mean(), r[1:, 2].mean() / 1e6))
        s.close()
