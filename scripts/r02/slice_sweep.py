import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
if which == "ff":
    wl = workloads.talos_c3(B, model=loik_amd.builtin_model("talos32_freeflyer"))
else:
    wl = workloads.talos_c3(B)
m, prm = wl["model"], wl["params"]
args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
ref_z = None
for sl in [int(x) for x in os.environ.get("SLICES", "0,16,32,48,64,96,128,256").split(",")]:
    os.environ["LOIKB_LEAN_SLICE"] = str(sl)
    s = loik_amd.BatchedLoik(m, B, **prm)
    s.SolveInit(*args)
    ts = []
    for _ in range(4):
        s.Solve()
        ts.append(s.stats()["total_ms"])
    st = s.stats()
    z, it = s.get("z"), s.get("iter")
    if ref_z is None:
        ref_z, ref_it = z, it
    print("%s B=%d slice %3d: %.2f ms (min %.2f)  requeues %6d  inst-it %d  same-as-slice0: it %s z %s" % (
        which, B, sl, np.median(ts), min(ts), st["lean_requeues"], st["instance_iterations"], np.array_equal(it, ref_it),
        np.array_equal(z, ref_z)), flush=True)
    s.close()
