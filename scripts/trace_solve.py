import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loik_amd
from loik_amd import workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kw = {}
if len(sys.argv) > 2:
    kw["tail_max_instances"] = int(sys.argv[2])
if len(sys.argv) > 3:
    kw["flags"] = int(sys.argv[3])
wl = workloads.talos_c3(B)
s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"], **kw)
s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
s.Solve()
os.environ["LOIKB_TRACE"] = "1"
s.Solve()
