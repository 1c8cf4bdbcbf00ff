"""the other BASELINE.json configurations (parity-test cases, not the bench line): throughput table for the docs"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads

def run(name, wl, B, reps=3, **kw):
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"], **kw)
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); s.Solve(); dt = time.perf_counter() - t
        best = min(best, dt)
    st = s.stats()
    conv = s.get("converged").astype(bool)
    out = dict(config=name, batch=B, ms_per_solve=best * 1e3, solves_per_s=float(conv.sum() / best) if conv.any() else None,
               instances_per_s=B / best, inst_iter_per_s=st["instance_iterations"] / best,
               algorithmic_GBs=st["instance_iterations"] * st["bytes_per_instance_iteration"] / best / 1e9,
               converged_fraction=float(conv.mean()), mean_iters=float(s.get("iter").mean()), launches=st["launches"],
               compactions=st["compactions"], tail_instances=st["tail_instances"])
    print(json.dumps(out), flush=True)
    res = s
    return res

# C2: Panda-7, B=4096, fixed 50 ADMM iterations, mu frozen, fp64
wl = workloads.panda_c2(4096)
run("C2 panda7 B=4096 fixed-50 fp64", wl, 4096, flags=capi.OPT_FIXED_ITERS).close()
# C3: headline
wl = workloads.talos_c3(65536)
run("C3 talos32 B=65536 tol1e-6 fp64", wl, 65536).close()
# C4-like: Talos, 131072 instances per GPU (the per-GPU share of B=1,048,576 over 8 GPUs), cold solves
wl = workloads.talos_c3(131072, seed=11)
run("C4 talos32 B=131072/GPU tol1e-6 fp64", wl, 131072, reps=2).close()
# C5: Panda-7 B=65536 fp32 vs fp64 at the tightest tolerance fp32 reaches comfortably
from loik_amd.workloads import make_workload, FIXTURE_PARAMS
m = loik_amd.builtin_model("panda7")
wl = make_workload(m, 65536, m.njoints - 1, 5, bound=2.0, snap_prob=0.0, nu_scale=0.5)
wl["model"] = m
for tol in (1e-3, 1e-4):
    wl["params"] = dict(FIXTURE_PARAMS, max_iter=300, tol_abs=tol, tol_rel=0.0)
    s64 = run("C5 panda7 B=65536 tol%g fp64" % tol, wl, 65536, reps=10)
    c64, z64 = s64.get("converged").astype(bool), s64.get("z")
    s64.close()   # (a second live handle's streams change how the small launches of the next one overlap: measure one at a time)
    s32 = run("C5 panda7 B=65536 tol%g fp32" % tol, wl, 65536, reps=10, precision=capi.F32)
    both = c64 & s32.get("converged").astype(bool)
    dz = np.abs(z64 - s32.get("z"))[both].max(axis=1)
    print(json.dumps(dict(config="C5 |z32-z64|_inf over instances converged in both", tol=tol, median=float(np.median(dz)),
                          p99=float(np.percentile(dz, 99)), max=float(dz.max()), both_fraction=float(both.mean()))), flush=True)
    s32.close()
