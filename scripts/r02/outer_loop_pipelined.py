"""C4 outer loop (T warm-started tailored solves, q resident on the device) with the batch split over K handles, each on its own
stream and host thread (LOIKB_OPT_OWN_STREAM): a sub-batch starts its next planner step as soon as ITS stragglers are done,
while the other sub-batches' launches fill the machine -- no sub-batch waits for the whole batch's ragged end."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import capi, workloads

B, T, dt = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 6, 0.1
wl = workloads.talos_c3(B, seed=21)
model, link = wl["model"], int(wl["c_ids"][0])
prm = dict(wl["params"], warm_start=True)
rng = np.random.default_rng(77)
nu_star = rng.uniform(-0.5, 0.5, size=(T, B, model.nv))
# targets of every step from the INITIAL q (a planner would re-target on the device; what is timed is solve + integrate)
b_all = [workloads.link_velocity(model, wl["q"], nu_star[t], link)[:, None, :] for t in range(T)]


def run(K):
    per = B // K
    solvers, slices = [], []
    for k in range(K):
        sl = slice(k * per, (k + 1) * per)
        s = loik_amd.BatchedLoik(model, per, flags=capi.OPT_OWN_STREAM if K > 1 else 0, **prm)
        s.SolveInit(wl["q"][sl], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"][sl], wl["lb"], wl["ub"])
        s.Solve(None, link, wl["Ais"], b_all[0][sl])    # warm-up (allocations, first touch); state is reset by SolveInit below
        s.SolveInit(wl["q"][sl], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"][sl], wl["lb"], wl["ub"])
        solvers.append(s); slices.append(sl)
    solved = [0] * K
    bar = threading.Barrier(K + 1)

    def work(k):
        s, sl = solvers[k], slices[k]
        bar.wait()
        for t in range(T):
            if t > 0:
                s.integrate(dt)
            s.Solve(None, link, wl["Ais"], b_all[t][sl])
            solved[k] += int(s.get("converged").sum())
        s.synchronize()
        bar.wait()
    th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    for x in th: x.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); el = time.perf_counter() - t0
    for x in th: x.join()
    for s in solvers: s.close()
    return dict(handles=K, batch_each=per, steps=T, total_ms=round(el * 1e3, 2), ms_per_planner_step_of_the_whole_batch=round(el * 1e3 / T, 2),
                planner_steps_per_s=round(B * T / el), solves_per_s=round(sum(solved) / el))


for K in (1, 2, 4, 8, 1):
    print(json.dumps(run(K)), flush=True)
