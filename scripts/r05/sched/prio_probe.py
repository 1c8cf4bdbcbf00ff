"""one fresh headline batch under the round-robin and the priority scheduler: time, parks, iterations executed"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = workloads.talos_c3(B)
for prio in ("0", "1"):
    os.environ["LOIKB_FLAT_PRIO"] = prio
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    for rep in range(2):
        s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        st = s.stats()
        print(json.dumps({"prio": prio, "rep": rep, "total_ms": round(st["total_ms"], 3), "tail_ms": round(st["tail_ms"], 3), "requeues": st["lean_requeues"],
                          "instance_iterations": int(s.get("iter").sum()), "plan": s.plan()[:80], **{k: st[k] for k in st if "iter" in k or "flat" in k}}))
        os.environ["LOIKB_FLAT_ORDER"] = "0"
    s.close()
