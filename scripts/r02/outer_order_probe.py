"""How well does the previous planner step's iteration count predict the next one's (C4 outer loop, warm-started tailored
solves)?  Rank correlation, recall of the long runners, and a list-scheduling estimate of the lean launch (4096 slots, run to
completion) for three queue orders: arrival, previous step's count descending, clairvoyant."""
import heapq, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads

B, T, dt = 65536, 6, 0.1
mode = sys.argv[1] if len(sys.argv) > 1 else "random"
wl = workloads.talos_c3(B, seed=21)
model, link = wl["model"], int(wl["c_ids"][0])
prm = dict(wl["params"], warm_start=True)
rng = np.random.default_rng(77)
nu_star = np.repeat(rng.uniform(-0.5, 0.5, size=(1, B, model.nv)), T, axis=0) if mode == "tracking" else rng.uniform(-0.5, 0.5, size=(T, B, model.nv))
s = loik_amd.BatchedLoik(model, B, **prm)
s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
q = wl["q"].copy()
its = []
for t in range(T):
    b_t = workloads.link_velocity(model, q, nu_star[t], link)[:, None, :]
    if t > 0:
        s.integrate(dt)
    s.Solve(None, link, wl["Ais"], b_t)
    its.append(s.get("iter").astype(np.int64))
    q = q + dt * s.get("z")


def sched(iters, order, slots=4096, t_it=9.2e-3, t_sw=6e-3):
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for b in order:
        t0 = heapq.heappop(free)
        t1 = t0 + t_sw + iters[b] * t_it
        end = max(end, t1)
        heapq.heappush(free, t1)
    return end


def rank(a):
    r = np.empty(len(a)); r[np.argsort(a, kind="stable")] = np.arange(len(a)); return r


for t in range(1, T):
    prev, cur = its[t - 1], its[t]
    rho = np.corrcoef(rank(prev), rank(cur))[0, 1]
    long_cur = cur >= 900
    recall = (prev[long_cur] >= 300).mean() if long_cur.any() else float("nan")
    res = dict(mode=mode, step=t, spearman=round(float(rho), 3), long_now=int(long_cur.sum()),
               long_now_that_were_ge300_before=round(float(recall), 3),
               ms_arrival=round(sched(cur, np.arange(B)), 2), ms_prev_desc=round(sched(cur, np.argsort(-prev, kind="stable")), 2),
               ms_clairvoyant=round(sched(cur, np.argsort(-cur, kind="stable")), 2), mean_iter=round(float(cur.mean()), 1))
    print(json.dumps(res), flush=True)
