"""event simulation of scheduling policies for the lean kernel: S slots, every iteration costs t_it per slot"""
import heapq, sys
import numpy as np
d = np.load("gpurun_out/r02_pred.npz")
T_IT = 10.4e-3  # ms per iteration of a resident instance (slot speed: 21.8k cycles at 2.1 GHz)
SW = 0.02       # ms to store + load an instance (switch)

def simulate(jobs_phases, S):
    """jobs_phases: list of rounds; each round = list of (job_id, iterations) in fetch order.  Round r+1 starts being fetched
    when the queue of round r is EMPTY (slots free up individually).  Returns makespan, time queue drained per round."""
    free = [0.0] * S
    heapq.heapify(free)
    end = 0.0
    drains = []
    for rnd in jobs_phases:
        for _, n in rnd:
            t = heapq.heappop(free)
            t2 = t + SW + n * T_IT
            end = max(end, t2)
            heapq.heappush(free, t2)
        drains.append(min(free))
    return end, drains

def study(name, S):
    it = d[name + "_iter"].astype(int)
    N = it.size
    print("== %s: S=%d  total %.3g inst-it, lower bounds: work %.2f ms, longest %.2f ms" % (name, S, it.sum(), it.sum() * T_IT / S, it.max() * T_IT))
    ids = np.arange(N)
    e, dr = simulate([list(zip(ids, it))], S)
    print("   current (one queue, run to completion):           %.2f ms (queue dry at %.2f)" % (e, dr[0]))
    order = np.argsort(-it)
    e, dr = simulate([list(zip(ids[order], it[order]))], S)
    print("   oracle LPT:                                        %.2f ms" % e)
    for Q in (8, 16, 24, 32, 48):
        fl = d["%s_flips_q%d" % (name, Q)].astype(int)
        r0 = [(i, min(it[i], Q)) for i in ids]
        surv = ids[it > Q]
        # (a) survivors in index order
        e_a, _ = simulate([r0, [(i, it[i] - Q) for i in surv]], S)
        # (b) sorted by flips desc
        o = surv[np.argsort(-fl[surv], kind="stable")]
        e_b, _ = simulate([r0, [(i, it[i] - Q) for i in o]], S)
        # (c) two-level: flips >= F first (run to completion), then the rest with a second quantum Q2, then their survivors
        best = None
        for F in (3, 4, 5, 6):
            for Q2 in (32, 64, 128):
                hi = surv[fl[surv] >= F]; lo = surv[fl[surv] < F]
                hi = hi[np.argsort(-fl[hi], kind="stable")]
                lo2 = [(i, min(it[i] - Q, Q2)) for i in lo]
                lo_s = lo[it[lo] > Q + Q2]
                # order: lo second-chance first (short), then hi + lo survivors
                e1, _ = simulate([r0, lo2, [(i, it[i] - Q) for i in hi] + [(i, it[i] - Q - Q2) for i in lo_s]], S)
                e2, _ = simulate([r0, lo2, [(i, it[i] - Q - Q2) for i in lo_s] + [(i, it[i] - Q) for i in hi]], S)
                e3, _ = simulate([r0, [(i, it[i] - Q) for i in hi], lo2, [(i, it[i] - Q - Q2) for i in lo_s]], S)
                for tag, e in (("lo2,hi+los", e1), ("lo2,los+hi", e2), ("hi,lo2,los", e3)):
                    if best is None or e < best[0]:
                        best = (e, F, Q2, tag)
        print("   Q=%2d: survivors %5d | index order %.2f | flips desc %.2f | best two-level %.2f (F=%d Q2=%d %s)" % (
            Q, surv.size, e_a, e_b, best[0], best[1], best[2], best[3]))
    # multi-round doubling quanta, each round sorted by flips? (only final flips known) -> skip
    # rounds with geometric quanta, survivors-first
    for qs in ([32, 64, 128, 256, 2000], [16, 32, 64, 128, 256, 2000], [24, 64, 160, 400, 2000]):
        done = np.zeros(N, dtype=int)
        rounds = []
        alive = ids
        for q in qs:
            rounds.append([(i, min(it[i] - done[i], q)) for i in alive])
            done[alive] += np.minimum(it[alive] - done[alive], q)
            alive = alive[it[alive] > done[alive]]
        e, _ = simulate(rounds, S)
        print("   rounds %s: %.2f ms" % (qs, e))

for name in ("c3", "ff"):
    study(name, 4096 if name == "c3" else 2048)

def las(name, S, qs, verbose=False):
    it = d[name + "_iter"].astype(int)
    N = it.size
    ids = np.arange(N)
    done = np.zeros(N, dtype=int)
    rounds = []
    alive = ids
    for q in qs:
        rounds.append([(i, min(it[i] - done[i], q)) for i in alive])
        done[alive] += np.minimum(it[alive] - done[alive], q)
        alive = alive[it[alive] > done[alive]]
        if alive.size == 0:
            break
    e, dr = simulate(rounds, S)
    if verbose:
        print("      drains", ["%.2f" % x for x in dr], "sizes", [len(r) for r in rounds])
    return e

print("---- LAS quanta search")
cands = {
    "24,64,160,400": [24, 64, 160, 400, 2000],
    "16,48,128,320": [16, 48, 128, 320, 2000],
    "32,96,256": [32, 96, 256, 2000],
    "24,40,64,128,256": [24, 40, 64, 128, 256, 2000],
    "16,16,32,64,128,256": [16, 16, 32, 64, 128, 256, 2000],
    "32,32,64,128,256": [32, 32, 64, 128, 256, 2000],
    "24,24,48,96,192,384": [24, 24, 48, 96, 192, 384, 2000],
    "20,30,50,100,200,300": [20, 30, 50, 100, 200, 300, 2000],
    "32,64,128,256,256": [32, 64, 128, 256, 256, 2000],
    "24,64,160,250,250": [24, 64, 160, 250, 250, 2000],
    "24,64,128,128,128,128,128": [24, 64, 128, 128, 128, 128, 128, 2000],
    "32x31": [32] * 31 + [2000],
    "64x15": [64] * 15 + [2000],
}
for k, qs in cands.items():
    print("   %-28s c3 %.2f   ff %.2f" % (k, las("c3", 4096, qs), las("ff", 2048, qs)))
las("c3", 4096, cands["24,64,160,400"], True)
las("c3", 4096, cands["32x31"], True)
