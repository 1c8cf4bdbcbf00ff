"""What does a long runner pay for company?  N copies of ONE 999-iteration instance of the headline workload (every wavefront of the launch
iterates in step for the whole launch): launch time / 999 for N = 1 .. 4096 -- one wavefront per CU (256), per SIMD (1024), two per SIMD (2048).
usage: python scripts/r06/pairing_probe.py   (TAG / LOIKB_* from the environment)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads

wl = workloads.talos_c3(4096)
s = loik_amd.BatchedLoik(wl["model"], 4096, **wl["params"])
s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
it = s.get("iter")
s.close()
k = int(np.argmax(it))
print("%s instance %d of talos_c3(4096): %d iterations" % (os.environ.get("TAG", ""), k, it[k]))
for N in [int(x) for x in os.environ.get("PROBE_N", "1,64,256,512,1024,1536,2048,3072,4096").split(",")]:
    q = np.repeat(wl["q"][k:k + 1], N, axis=0)
    b = np.repeat(wl["bis"][k:k + 1], N, axis=0)
    s = loik_amd.BatchedLoik(wl["model"], N, **wl["params"])
    s.SolveInit(q, wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], b, wl["lb"], wl["ub"])
    best = 1e9
    for _ in range(4):
        s.Solve()
        st = s.stats()
        own = st["tail_ms"] - st["hslots_ms"]
        best = min(best, own)
    itn = s.get("iter")
    assert itn.min() == itn.max() == it[k], (itn.min(), itn.max())
    print("%s N=%5d: launch %.3f ms = %.3f us per iteration of a wavefront   flat launches %d requeues %d built %d" % (
        os.environ.get("TAG", ""), N, best, best * 1e3 / it[k] / max(1, (N + 2047) // 2048), st["flat_launches"], st["lean_requeues"], st.get("flat_built", -1)))
    s.close()
