"""throughput against batch size on one GPU (Talos headline task): the ragged end of a solve (the serial chains of the
1000-iteration instances, ~10 ms) is amortised over a longer bulk phase"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loik_amd
from loik_amd import workloads
for B in [int(x) for x in sys.argv[1:]] or [16384, 32768, 65536, 131072, 262144, 524288]:
    wl = workloads.talos_c3(B, seed=0x101C + 3)
    s = loik_amd.BatchedLoik(wl["model"], B, **wl["params"])
    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); s.Solve(); best = min(best, time.perf_counter() - t)
    st = s.stats(); conv = s.get("converged").astype(bool)
    print(json.dumps(dict(batch=B, ms_per_solve=best * 1e3, solves_per_s=float(conv.sum() / best), inst_iter_per_s=st["instance_iterations"] / best,
                          lean_launches=st["lean_launches"], decade_slots_ms=st["hslots_ms"])), flush=True)
    s.close()
