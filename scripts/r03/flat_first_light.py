"""first light of the flat engine: headline workload at a small batch against the oracle, engine trace on stderr"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import loik_amd
from loik_amd import workloads
from oracle import ref

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = workloads.talos_c3(B)
m, prm = wl["model"], wl["params"]
args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
out = ref.solve_batch(m, *args, nthreads=16, **prm)
for env in ({"LOIKB_FLAT": "0"}, {"LOIKB_FLAT": "1"}):
    os.environ.update(env)
    s = loik_amd.BatchedLoik(m, B, **prm)
    print(env, s.plan())
    s.Solve(*args)
    t0 = time.time(); s.Solve(*args); dt = time.time() - t0
    it, z = s.get("iter"), s.get("z")
    conv, inf = s.get("converged").astype(bool), s.get("primal_infeasible").astype(bool)
    same = it == out["iters"]
    err = np.abs(z - out["z"]).max(1)
    st = s.stats()
    print("  same iteration count %d / %d; max|dz| same %.2e rest %.2e; conv %d/%d pinf %d/%d; flags equal on same: %s %s; %.2f ms wall, kernel %.2f ms, slots %.2f ms, escaped %d flat launches %d"
          % (same.sum(), B, err[same].max(), err[~same].max() if (~same).any() else 0, conv.sum(), out["converged"].sum(), inf.sum(),
             out["primal_infeasible"].sum(), np.array_equal(conv[same], out["converged"][same]),
             np.array_equal(inf[same], out["primal_infeasible"][same]), dt * 1e3, st["kernel_ms"], st["hslots_ms"], st["lean_escaped"], st["flat_launches"]))
    if not same.all():
        bad = np.nonzero(~same)[0][:10]
        print("  off:", [(int(b), int(it[b]), int(out["iters"][b])) for b in bad])
    s.close()
