import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
out = {}
for name, wl in [("c3", workloads.talos_c3(65536)), ("ff", None)]:
    if wl is None:
        m = loik_amd.builtin_model("talos32_freeflyer")
        wl = workloads.talos_c3(65536, model=m)
    m, prm = wl["model"], wl["params"]
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    s = loik_amd.BatchedLoik(m, 65536, **prm)
    s.Solve(*args)
    out[name + "_iter"] = s.get("iter"); out[name + "_flips"] = s.get("mu_updates")
    print(name, "ms", s.stats()["total_ms"], "mean it", out[name + "_iter"].mean())
    s.close()
    for Q in (8, 16, 24, 32, 48):
        s = loik_amd.BatchedLoik(m, 65536, **dict(prm, max_iter=Q + 1))
        s.Solve(*args)
        out["%s_flips_q%d" % (name, Q)] = s.get("mu_updates")
        out["%s_mu_q%d" % (name, Q)] = s.get("mu")
        out["%s_pr_q%d" % (name, Q)] = s.get("primal_residual")
        out["%s_du_q%d" % (name, Q)] = s.get("dual_residual")
        s.close()
np.savez_compressed(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r02_pred.npz"), **out)
