"""event-driven simulation of in-kernel scheduling policies (correct precedence: an instance is on one slot at a time)"""
import heapq, sys
import numpy as np
d = np.load("gpurun_out/r02_pred.npz")
T_IT = 10.4e-3; SW = 0.012

def run(it, S, policy):
    """policy(level) -> quantum for an instance that has completed `level` slices; queue discipline given by `pick`.
    State: FIFO queues per level.  pick order = policy.order(levels with entries)."""
    N = len(it)
    nlev = policy["nlev"]
    queues = [[] for _ in range(nlev)]
    heads = [0] * nlev
    queues[0] = list(range(N))
    done_it = np.zeros(N, dtype=np.int64)
    level = np.zeros(N, dtype=np.int64)
    ev = []  # (time, slot)
    t = 0.0
    free = list(range(S))
    running = {}
    end = 0.0
    def nonempty():
        return [l for l in range(nlev) if heads[l] < len(queues[l])]
    def dispatch(now):
        nonlocal end
        while free:
            ls = nonempty()
            if not ls:
                break
            l = policy["pick"](ls)
            i = queues[l][heads[l]]; heads[l] += 1
            q = policy["quantum"](l)
            n = min(it[i] - done_it[i], q)
            s = free.pop()
            tf = now + SW + n * T_IT
            heapq.heappush(ev, (tf, s, i, n))
    dispatch(0.0)
    while ev:
        tf, s, i, n = heapq.heappop(ev)
        done_it[i] += n
        end = max(end, tf)
        if done_it[i] < it[i]:
            # continue in place if nothing is waiting (no migration), else requeue at next level
            waiting = len(nonempty()) > 0
            nl = min(level[i] + 1, nlev - 1)
            level[i] = nl
            if not waiting:
                q = policy["quantum"](nl)
                n2 = min(it[i] - done_it[i], q)
                heapq.heappush(ev, (tf + n2 * T_IT, s, i, n2))
                continue
            queues[nl].append(i)
        free.append(s)
        dispatch(tf)
    return end

def study(name, S):
    it = d[name + "_iter"].astype(np.int64)
    lb = max(it.sum() * T_IT / S, it.max() * T_IT)
    print("== %s S=%d lower bound %.2f" % (name, S, lb))
    INF = 10 ** 9
    pols = {}
    pols["fifo run-to-completion"] = dict(nlev=1, quantum=lambda l: INF, pick=lambda ls: ls[0])
    for q in (32, 64, 128):
        pols["LAS rr q=%d (lowest level first)" % q] = dict(nlev=40, quantum=lambda l, q=q: q, pick=lambda ls: ls[0])
    for qs in ([24, INF], [32, INF], [48, INF], [24, 64, INF], [32, 96, INF], [16, 32, 64, 128, INF], [32, 64, 128, 256, INF], [24, 64, 160, 400, INF]):
        n = len(qs)
        pols["levels %s, HIGHEST level first" % qs] = dict(nlev=n, quantum=lambda l, qs=qs: qs[l], pick=lambda ls: ls[-1])
        pols["levels %s, lowest level first" % qs] = dict(nlev=n, quantum=lambda l, qs=qs: qs[l], pick=lambda ls: ls[0])
    for k, p in pols.items():
        print("   %-62s %.2f ms" % (k, run(it, S, p)))

study("c3", 4096)
study("ff", 2048)

def study2(name, S):
    it = d[name + "_iter"].astype(np.int64)
    INF = 10 ** 9
    print("== %s: discovery first, then highest level first" % name)
    for qs in ([8, 16, 32, 64, 128, 256, INF], [16, 16, 32, 64, 128, 256, INF], [16, 32, 64, 128, 256, INF], [16, 48, 128, INF], [24, 40, 64, 128, INF], [32, 32, 64, 128, 256, INF], [16, 1000]):
        n = len(qs)
        pol = dict(nlev=n, quantum=lambda l, qs=qs: qs[l], pick=lambda ls: 0 if 0 in ls else ls[-1])
        print("   %-50s %.2f ms" % (qs, run(it, S, pol)))
study2("c3", 4096)
study2("ff", 2048)

def run_sorted(it, S, Q, score, second=None):
    """discovery pass (Q iterations each, FIFO), then survivors by score descending, run to completion.
    second=(Q2, score2_fn): the low half gets a second bounded slice first and is re-scored"""
    N = len(it)
    free = [0.0] * S
    heapq.heapify(free)
    avail = np.zeros(N)  # time at which the instance's previous slice ended
    end = 0.0
    for i in range(N):
        t = heapq.heappop(free)
        tf = t + SW + min(it[i], Q) * T_IT
        avail[i] = tf; end = max(end, tf)
        heapq.heappush(free, tf)
    surv = np.flatnonzero(it > Q)
    order = surv[np.argsort(-score[surv], kind="stable")]
    for i in order:
        t = max(heapq.heappop(free), avail[i])
        tf = t + SW + (it[i] - Q) * T_IT
        end = max(end, tf)
        heapq.heappush(free, tf)
    return end

def study3(name, S):
    it = d[name + "_iter"].astype(np.int64)
    print("== %s: discovery pass then survivors sorted by a score" % name)
    for Q in (8, 16, 24, 32, 48):
        fl = d["%s_flips_q%d" % (name, Q)].astype(float)
        pr = np.log10(np.maximum(d["%s_pr_q%d" % (name, Q)], 1e-300)); du = np.log10(np.maximum(d["%s_du_q%d" % (name, Q)], 1e-300))
        mu = np.log10(d["%s_mu_q%d" % (name, Q)])
        res = np.maximum(pr, du)
        print("   Q=%2d: oracle %.2f | flips %.2f | max residual %.2f | flips+res %.2f | random %.2f" % (
            Q, run_sorted(it, S, Q, it.astype(float)), run_sorted(it, S, Q, fl), run_sorted(it, S, Q, res),
            run_sorted(it, S, Q, fl + 2 * (res + 6)), run_sorted(it, S, Q, np.random.default_rng(0).random(it.size))))
study3("c3", 4096)
study3("ff", 2048)

def run_pin(it, S, Q0, score, thr, q=64, a_pin=10**9, max_pinned_frac=1.0):
    """every instance first gets Q0 iterations.  Then: score >= thr -> pinned (runs to completion on its slot, no preemption);
    others are time-sliced round-robin (quantum q, requeue only if something waits) and pinned once attained >= a_pin."""
    N = len(it)
    fifo = list(range(N)); head = 0
    done = np.zeros(N, dtype=np.int64)
    ev = []; free = list(range(S)); end = 0.0
    def dispatch(now):
        nonlocal head
        while free and head < len(fifo):
            i = fifo[head]; head += 1
            first = done[i] == 0
            if first:
                n = min(it[i], Q0)
            else:
                n = min(it[i] - done[i], q)
            s = free.pop()
            heapq.heappush(ev, (now + SW + n * T_IT, s, i, n))
    dispatch(0.0)
    while ev:
        tf, s, i, n = heapq.heappop(ev)
        done[i] += n; end = max(end, tf)
        if done[i] < it[i]:
            pinned = score[i] >= thr or done[i] >= a_pin
            waiting = head < len(fifo)
            if pinned or not waiting:
                n2 = (it[i] - done[i]) if pinned else min(it[i] - done[i], q)
                heapq.heappush(ev, (tf + n2 * T_IT, s, i, n2))
                continue
            fifo.append(i)
        free.append(s)
        dispatch(tf)
    return end

def study4(name, S):
    it = d[name + "_iter"].astype(np.int64)
    print("== %s: pin predicted-long, time-slice the rest" % name)
    for Q in (16, 32):
        fl = d["%s_flips_q%d" % (name, Q)].astype(float)
        pr = np.log10(np.maximum(d["%s_pr_q%d" % (name, Q)], 1e-300)); du = np.log10(np.maximum(d["%s_du_q%d" % (name, Q)], 1e-300))
        res = np.maximum(pr, du)
        for label, sc, thrs in (("flips", fl, (3, 5, 7)), ("maxres", res, (-2.0, -3.0, -4.0))):
            for thr in thrs:
                for a_pin in (10**9, 256):
                    e = run_pin(it, S, Q, sc, thr, 64, a_pin)
                    print("   Q0=%d %s>=%g pinned %5d  a_pin %-10d -> %.2f ms" % (Q, label, thr, int(((sc >= thr) & (it > Q)).sum()), a_pin, e))
study4("c3", 4096)
study4("ff", 2048)
