import sys
sys.path.insert(0, "/root/repo")
import numpy as np, loik_amd
from loik_amd import workloads
B = int(sys.argv[1])
wl = workloads.talos_c3(B)
s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
for i in range(2):
    s.Solve()
    st = s.stats()
    print({k: st[k] for k in ("launches", "tail_launches", "lean_launches", "flat_launches", "flat_split_launches", "lean_escaped", "lean_requeues", "tail_instances", "n_unfinished")})
