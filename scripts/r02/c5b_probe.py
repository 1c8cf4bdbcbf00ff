import json, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import loik_amd
from loik_amd import capi
from loik_amd.workloads import make_workload, FIXTURE_PARAMS
m = loik_amd.builtin_model("panda7")
wl = make_workload(m, 65536, m.njoints - 1, 5, bound=2.0, snap_prob=0.0, nu_scale=0.5)
prm = dict(FIXTURE_PARAMS, max_iter=300, tol_abs=1e-3, tol_rel=0.0)
keep = []
for prec, name in ((capi.F64, "fp64"), (capi.F32, "fp32"), (capi.F64, "fp64"), (capi.F32, "fp32")):
    s = loik_amd.BatchedLoik(m, 65536, precision=prec, **prm)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    ts = []
    for _ in range(8):
        t = time.perf_counter(); s.Solve(); ts.append(time.perf_counter() - t)
    st = s.stats()
    it = s.get("iter")
    print(name, "ms", [round(x * 1e3, 3) for x in ts], "iters mean %.2f max %d" % (it.mean(), it.max()),
          {k: (round(st[k], 3) if isinstance(st[k], float) else st[k]) for k in ("launches", "kernel_ms", "tail_ms", "tail_instances", "tail_launches", "compactions", "total_ms", "solve_busy_ms", "tail_busy_ms", "chunks")})
    keep.append(s) if os.environ.get('KEEP') else s.close()
