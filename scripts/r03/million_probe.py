"""one GPU, 2^20 Talos instances in one handle (C4's whole planner population on one device): memory, time, properties"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import loik_amd
from loik_amd import workloads
from oracle import ref
B = 1 << 20
wl = workloads.talos_c3(B, seed=99)
m, prm = wl["model"], wl["params"]
t = time.perf_counter()
s = loik_amd.BatchedLoik(m, B, **prm)
s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
print("create + SolveInit %.2f s" % (time.perf_counter() - t), flush=True)
try:
    import torch
    free, total = torch.cuda.mem_get_info(0)
    print("device memory in use after SolveInit: %.2f GB of %.1f GB" % ((total - free) / 1e9, total / 1e9), flush=True)
except Exception as e:
    print("no memory figure:", e)
for k in range(3):
    t = time.perf_counter(); s.Solve(); dt = time.perf_counter() - t
    conv = s.get("converged").astype(bool)
    print("solve %d: %.1f ms, %.2f M solves/s, converged %.4f, %s" % (k, dt * 1e3, conv.sum() / dt / 1e6, conv.mean(), {x: s.stats()[x] for x in ("flat_launches", "lean_escaped", "hslots_ms", "tail_ms", "queue_dry_ms")}), flush=True)
z, nu = s.get("z"), s.get("nu")
assert np.all(z <= wl["ub"] + 1e-12) and np.all(z >= wl["lb"] - 1e-12) and np.max(np.abs(nu - z)[conv]) < 1e-6
idx = np.arange(0, B, 4099)
out = ref.solve_batch(m, wl["q"][idx], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"][idx], wl["lb"], wl["ub"], nthreads=8, **prm)
same = s.get("iter")[idx] == out["iters"]
print("strided oracle sample of %d: same-iteration %.3f, max |dz| over those %.2e" % (idx.size, same.mean(), np.abs(z[idx] - out["z"])[same].max()))
print(s.plan())
