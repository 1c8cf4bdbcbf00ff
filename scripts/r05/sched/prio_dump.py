"""(debug build, -DLOIKB_PQ_DEBUG) one fresh headline batch under the priority scheduler: the park records against what the instances really needed"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
B = 65536
wl = workloads.talos_c3(B)
os.environ["LOIKB_FLAT_PRIO"] = "1"; os.environ["LOIKB_TRACE"] = "1"; os.environ["LOIKB_PQ_DUMP"] = "gpurun_out/prio/parks.txt"
s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
it = s.get("iter")
np.save("gpurun_out/prio/iters.npy", it)
r = np.loadtxt("gpurun_out/prio/parks.txt")
idx = r[:, 0].astype(int); k = r[:, 1]; pred = r[:, 2]; t = r[:, 4]
true_rem = it[idx] - k
print("parks", len(r), "instances parked more than once", len(idx) - len(np.unique(idx)))
first = k <= 130
print("first parks:", first.sum(), " later parks:", (~first).sum())
for lo, hi in ((0, 50), (50, 100), (100, 200), (200, 400), (400, 700), (700, 2000)):
    m = first & (true_rem >= lo) & (true_rem < hi)
    if m.sum(): print("true remaining [%d,%d): n %5d  predicted: p10 %5d med %5d p90 %5d   under-predicted by > 1.5x+32: %d" % (lo, hi, m.sum(), *np.percentile(pred[m], [10, 50, 90]), (true_rem[m] > 1.5 * pred[m] + 32).sum()))
late = ~first
if late.any(): print("later parks: time ms p10/med/p90/max", np.round(np.percentile(t[late], [10, 50, 90, 100]), 2), " their iter med", np.median(k[late]), " true remaining med/max", np.median(true_rem[late]), true_rem[late].max())
e = np.loadtxt("gpurun_out/prio/parks.txt.ends")   # class of the last pop, time of the last pop (-1: never parked), end time
cls, tpop, tend = e[:, 0], e[:, 1], e[:, 2]
order = np.argsort(-tend)[:25]
print("end of launch: the last 25 instances to finish (instance, iterations, last pop ms, class of it, end ms)")
for b in order: print("  %6d %4d  pop %7.3f  class %3d  end %7.3f   parks %d" % (b, it[b], tpop[b], cls[b], tend[b], (idx == b).sum()))
print("instances ending after 8.5 ms:", (tend > 8.5).sum(), " of them never parked:", ((tend > 8.5) & (tpop < 0)).sum())
pp = tpop >= 0
print("popped instances: end - pop (ms) per iteration run after the pop: ", np.round(np.percentile(((tend - tpop)[pp] * 1e3) / np.maximum(1, (it[pp] - 128)), [10, 50, 90]), 2), "us")
hm = it >= 999
print("hit-max instances:", hm.sum(), " pop time p10/50/90/max", np.round(np.percentile(tpop[hm], [10, 50, 90, 100]), 2), " end p10/50/90/max", np.round(np.percentile(tend[hm], [10, 50, 90, 100]), 2))
