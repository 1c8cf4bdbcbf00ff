"""Discrete-event estimate: k_flat2's work queue run to completion in arrival order (what it does) against round-robin time
slicing inside the launch (an instance whose quantum expires while others wait goes to the back of the queue), with the
measured parameters of round 3: 2048 resident wavefronts (one instance each), 3.98 us per iteration while both wavefronts of a
SIMD are busy, 2.9 us for a wavefront alone on its SIMD, `switch_us` per store + load of an instance.  Iteration counts: the
oracle's on the headline workload.  usage: python scripts/r03/sim_slicing.py [B]"""
import sys, heapq
import numpy as np
sys.path.insert(0, ".")
from loik_amd import workloads
from oracle import ref

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = workloads.talos_c3(B)
out = ref.solve_batch(wl["model"], wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"], nthreads=16, **wl["params"])
its = out["iters"].astype(int)
print("instances %d, iterations %d, mean %.1f, at max %d" % (B, its.sum(), its.mean(), (its >= 999).sum()))
SLOTS = 2048


def simulate(quantum, switch_us, t_pair=3.98, t_lone=2.9, load_us=None):
    """event loop over slots; speed of a slot depends on how many slots are busy (SIMD pairs fill up evenly)"""
    load_us = switch_us / 2 if load_us is None else load_us
    from collections import deque
    q = deque((i, its[i]) for i in range(B))     # (instance, remaining iterations)
    busy = 0
    t = 0.0
    ev = []   # (finish time of the current slice, remaining after slice, instance)
    def it_time(nbusy):
        f = min(max((nbusy - 1024) / 1024.0, 0.0), 1.0)   # share of SIMDs holding two wavefronts
        return t_lone + f * (t_pair - t_lone)
    # (speed is evaluated when a slice starts: good enough for an estimate)
    def start(now):
        nonlocal busy
        while q and busy < SLOTS:
            i, rem = q.popleft()
            busy += 1
            run = rem if quantum is None else min(rem, quantum)
            heapq.heappush(ev, (now + load_us + run * it_time(busy), rem - run, i))
    start(0.0)
    while ev:
        t, rem, i = heapq.heappop(ev)
        busy -= 1
        if rem > 0:
            if q:          # others wait: to the back of the queue (store now, load when it is taken again)
                q.append((i, rem)); t_extra = switch_us - load_us
            else:          # nothing waits: it simply continues
                busy += 1
                run = rem if quantum is None else min(rem, quantum)
                heapq.heappush(ev, (t + run * it_time(busy), rem - run, i))
                continue
        start(t)
    return t / 1e3


for sw in (10.0, 25.0, 40.0):
    base = simulate(None, sw)
    print("switch %4.0f us: run to completion %.2f ms" % (sw, base), end="")
    for qn in (32, 64, 128, 256):
        print("   q=%d: %.2f" % (qn, simulate(qn, sw)), end="")
    print()
