"""headline batch under the three penalty rules (DEFAULT, OSQP, MAXEIGENVALUE) on the engines the plan picks: ms per solve of a repeated batch,
converged share, mean iterations, in-wave builds -- one JSON line per rule (profiles/r05_*_mu_rules.jsonl)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
whole = len(sys.argv) > 2 and sys.argv[2] == "wholebody"   # (the 44-joint tree: k_flat1)
wl = workloads.talos_wholebody(B) if whole else workloads.talos_c3(B)
for name, strat in (("DEFAULT", 0), ("OSQP", 1), ("OSQP unsliced", 1), ("OSQP k_solve + k_tail (LOIKB_FLAT=0)", 1), ("MAXEIGENVALUE", 3)):
    os.environ.pop("LOIKB_FLAT_SLICE", None); os.environ.pop("LOIKB_FLAT", None)
    if "unsliced" in name: os.environ["LOIKB_FLAT_SLICE"] = "0"
    if "LOIKB_FLAT=0" in name: os.environ["LOIKB_FLAT"] = "0"
    prm = dict(wl["params"], mu_update_strat=strat)
    s = loik_amd.BatchedLoik(wl["model"], B, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    rows = []
    for i in range(5):
        s.Solve()
        st = s.stats()
        rows.append((st["total_ms"], st["tail_ms"], st["hslots_ms"], st["flat_ordered"], st["flat_built"]))
    it = s.get("iter"); conv = s.get("converged").astype(bool)
    first, rest = rows[0], np.array(rows[2:])
    print(json.dumps({"rule": name, "robot": "talos44" if whole else "talos32", "batch": B, "plan": s.plan()[:120], "first_solve_ms": round(first[0], 3), "repeat_solve_ms": round(float(rest[:, 0].mean()), 3),
                      "slots_ms": round(float(rest[:, 2].mean()), 3), "converged_share": round(float(conv.mean()), 4), "mean_iterations": round(float(it.mean()), 2),
                      "hit_max_iter_share": round(float((it >= prm["max_iter"] - 1).mean()), 5), "in_wave_builds_per_solve": int(rest[:, 4].mean()),
                      "solves_per_s_first": round(float(conv.sum()) / first[0] * 1e3), "solves_per_s_repeat": round(float(conv.sum()) / float(rest[:, 0].mean()) * 1e3),
                      "flat_split_launches": st["flat_split_launches"], "tail_instances": st["tail_instances"]}))
    s.close()
