#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native batched LoIK solver.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Metric (BASELINE.json): IK solves/s to 1e-6 residual, Talos humanoid, batch = 65536 per GPU, fp64, adaptive mu.
A "step" is one cold `Solve()` (the reference's hot loop, /root/reference/include/loik/loik-loid-optimized.hpp:368-455)
over one batch of 65536 synthetic problem instances whose inputs were placed in HBM by `SolveInit()` before the
timed region -- the same split the reference's own timing test uses (`SolveInit` once, then time `Solve()`,
/root/reference/tests/loik-loid.cpp:987-1032).  A "solve" is an instance that stops with primal AND dual residual
below 1e-6 (`get_convergence_status()`); instances that trip the reference's infeasibility certificate or hit
max_iter are executed and timed but not counted.  The batch shards over GPUs with no exchange step (instances are
independent): every rank solves its own 65536 instances, no collective on the data path ("scaling": "weak").

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel -- the one that ran most of the ADMM
instance-iterations of the timed steps -- against HBM bandwidth with the ALGORITHMIC byte model of SURVEY.md 8(d):
bytes per ADMM instance-iteration = sizeof(scalar)*(203 nb + 108 nc), times the instance-iterations the kernel executed,
over the time during which at least one of its launches was executing (HIP events recorded by the library on the
streams the kernels are launched on).  By default the whole batch runs in `k_lean` (loik_amd/csrc/loik_lean.hpp: one
joint per lane, an instance's whole ADMM state in registers/LDS until it stops, two wavefronts per SIMD; `k_hslots`
precomputes H_i/Dinv_i per decade of mu before it): its HBM traffic is one load and one store per instance plus the
decade slots (reported as `traffic`), so the streaming model's roofline does not bind it -- a `frac` above 1 says exactly
that -- and what does (fp64 issue, the serial chains of the 1000-iteration instances) is stated in `regime`.  With
LOIKB_LEAN=0 two kernels share a solve (`k_solve`: one instance per lane through HBM-resident tiles, HBM-bound; `k_tail`:
the one-wavefront-per-SIMD predecessor of `k_lean`); the kernel that ran fewer instance-iterations is then reported
beside the dominant one (`other_kernel`).
`cpu_baseline` times the CPU oracle (a line-faithful port of the reference solver, NOT upstream libloik) on a
bounded sample of the same workload on all host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(wl, budget_s=15.0):
    """oracle (kind "port") on the box's host cores, bounded sample of the same workload"""
    from oracle import ref
    cores = os.cpu_count() or 1
    m, prm = wl["model"], wl["params"]

    def run(n):
        t = time.perf_counter()
        out = ref.solve_batch(m, wl["q"][:n], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"][:n], wl["lb"],
                              wl["ub"], nthreads=cores, native=True, **prm)
        return time.perf_counter() - t, out

    n0 = min(wl["q"].shape[0], 8 * cores)
    t0, _ = run(n0)  # pilot (also warms the thread pool)
    t0, _ = run(n0)
    n = int(min(wl["q"].shape[0], max(n0, n0 * budget_s / max(t0, 1e-4))))
    n = max(cores, (n // cores) * cores)
    dt, out = run(n)
    # one thread, one instance: microseconds per ADMM iteration, as the reference's own timing test measures it
    # (SolveInit once, then Solve() with max_iter = 2, i.e. exactly one iteration; /root/reference/tests/loik-loid.cpp:987-1032)
    one_us = None
    try:
        t1 = time.perf_counter()
        o1 = ref.solve_batch(m, wl["q"][:64], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"][:64], wl["lb"],
                             wl["ub"], nthreads=1, native=True, **dict(prm, max_iter=201, tol_abs=0.0, tol_primal_inf=0.0))
        one_us = (time.perf_counter() - t1) / max(int(o1["iters"].sum()), 1) * 1e6
    except Exception:
        pass
    return dict(value=float(out["converged"].sum() / dt), unit="solves/s", cores=cores, kind="port",
                single_thread_us_per_iteration=one_us,
                sample="first %d instances of the same workload, %d threads, %.1f s, %.0f ADMM instance-iterations/s; "
                       "oracle/loik_ref.c = line-faithful C port of the reference solver (not upstream libloik)"
                       % (n, cores, dt, out["iters"].sum() / dt),
                instance_iterations_per_s=float(out["iters"].sum() / dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--max-launch-iters", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the data path has no collective; the process group only carries the timing barrier / max-reduce
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import loik_amd
    from loik_amd import workloads

    def device_sync():
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception:
            pass

    def barrier():
        device_sync()
        if dist is not None:
            dist.barrier()
        device_sync()

    ndev = loik_amd.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    device = local_rank
    if device >= ndev:
        # fewer visible GPUs than ranks (only for smoke-testing the multi-process path on a 1-GPU box)
        if os.environ.get("LOIKB_ALLOW_SHARED_GPU") != "1":
            raise SystemExit("rank %d has no GPU of its own (%d visible); set LOIKB_ALLOW_SHARED_GPU=1 to share" % (rank, ndev))
        device = local_rank % ndev

    B = args.batch
    wl = workloads.talos_c3(B, seed=0x101C + 3 + rank)
    model, prm = wl["model"], wl["params"]
    solver = loik_amd.BatchedLoik(model, B, device=device, flags=args.flags, max_launch_iters=args.max_launch_iters,
                                  **prm)
    t_init = time.perf_counter()
    solver.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    t_init = time.perf_counter() - t_init  # includes the PCIe upload of the host-side synthetic inputs

    for _ in range(args.warmup):
        solver.Solve()
    kernel_ms = 0.0
    tail_ms = 0.0
    inst_iters = 0
    launches = 0
    tail_inst = 0
    tail_iters = 0
    tail_launches = 0
    solve_wall_ms = 0.0
    solve_busy_ms = 0.0
    tail_busy_ms = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.Solve()
        st = solver.stats()
        kernel_ms += st["kernel_ms"]
        tail_ms += st["tail_ms"]
        tail_inst += st["tail_instances"]
        tail_iters += st["tail_instance_iterations"]
        tail_launches += st["tail_launches"]
        solve_wall_ms += st["total_ms"]
        solve_busy_ms += st["solve_busy_ms"]
        tail_busy_ms += st["tail_busy_ms"]
        inst_iters += st["instance_iterations"]
        launches += st["launches"]
    barrier()
    elapsed = time.perf_counter() - t0

    conv = solver.get("converged").astype(bool)
    it = solver.get("iter")
    infeas = solver.get("primal_infeasible").astype(bool)
    n_solved = int(conv.sum())
    hit_max = int(((it >= prm["max_iter"] - 1) & ~conv).sum())
    from loik_amd import sharding
    elapsed, tot = sharding.aggregate(dist, elapsed, dict(solved=n_solved, iters=int(it.sum()), batch=B))
    total_solved, total_iters, total_B = tot["solved"], tot["iters"], tot["batch"]
    stats = dict(infeasible=int(infeas.sum()), unfinished=hit_max)

    if rank == 0:
        bytes_iter = st["bytes_per_instance_iteration"]
        # per kernel: algorithmic bytes (SURVEY.md 8(d) per-unit figure x instance-iterations it ran) over the time during
        # which at least one of its launches was executing (HIP events on the launch streams; the batch may be solved
        # as concurrent chunks, so launches of one kernel overlap) -- and the HBM bytes the PMC counters saw
        # (scripts/pmc_traffic.sh; rocprofv3 cannot run inside this process, the committed summary is used when it
        # describes this workload)
        pmc = None
        tj = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tj) and B == 65536:
            try:
                pmc = json.load(open(tj))["kernels"]
            except Exception:
                pmc = None

        def kernel_entry(key, name, iters, launches_k, sum_ms, busy_ms, extra, pmc=pmc):
            ach = iters * bytes_iter / (busy_ms * 1e-3) / 1e9 if busy_ms > 0 else 0.0
            traffic, note = None, "no PMC summary for this workload (run scripts/pmc_traffic.sh on the GPU box)"
            if key == "k_tail" and pmc and "k_lean" in pmc and lean:  # the lean kernel and its slot precomputation
                pmc = dict(pmc, k_tail={k_: pmc["k_lean"][k_] + pmc.get("k_hslots", {}).get(k_, 0.0) for k_ in pmc["k_lean"]})
            if pmc and key in pmc and launches_k > 0:
                kb = pmc[key]["fetch_bytes_per_step"] + pmc[key]["write_bytes_per_step"]
                traffic = kb / max(launches_k / args.steps, 1)
                note = "profiles/traffic_latest.json: %.3g HBM bytes of %s per Solve() step (%.0f launches there, %.0f here)" % (
                    kb, key, pmc[key]["dispatches_per_step"], launches_k / args.steps)
            e = {"kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                 "traffic": traffic, "traffic_note": note,
                 "units_per_launch": iters / max(launches_k, 1), "avg_launch_ms": sum_ms / max(launches_k, 1),
                 "launches_per_step": launches_k / args.steps, "sum_of_launch_ms_per_step": sum_ms / args.steps,
                 "busy_ms_per_step": busy_ms / args.steps,
                 "launch_concurrency": sum_ms / busy_ms if busy_ms > 0 else None,
                 "share_of_instance_iterations": iters / max(inst_iters, 1),
                 "instance_iterations_per_s": iters / (busy_ms * 1e-3) if busy_ms > 0 else None}
            e.update(extra)
            return e

        solve_ms = kernel_ms - tail_ms
        solve_iters = inst_iters - tail_iters
        solve_launches = launches - tail_launches
        k_solve = kernel_entry(
            "k_solve", "k_solve<double, team of %d wavefronts per 64-instance tile>" % st["team"], solve_iters,
            solve_launches, solve_ms, solve_busy_ms,
            {"regime": "one instance per lane, state streamed through HBM every iteration: HBM-bound in bulk"})
        lean = st["lean_launches"] > 0
        tail_name = ("k_lean<double> (tail kernel at two wavefronts per SIMD: a 32-lane group per instance, one joint per lane, "
                     "state in registers/LDS, H/Dinv/UDinv of the decades of mu precomputed by k_hslots)" if lean else
                     "k_tail<double> (a 32-lane group per instance, one joint per lane, state in registers/LDS)")
        k_tail = kernel_entry(
            "k_tail", tail_name, tail_iters, tail_launches, tail_ms, tail_busy_ms,
            {"regime": "the state of an instance stays on chip for its whole solve: the kernel's HBM traffic is one load "
                       "and one store per instance plus the decade slots (`traffic`), so the streaming byte model's "
                       "roofline does not bind it; what does is fp64 VALU issue latency along the tree levels "
                       "(DESIGN.md section 4: phase timeline, scripts/ubench/fp64_issue.hip)",
             "instances_per_step": tail_inst / args.steps,
             "valu_note": "PMC (profiles/r01_j_k_lean_pmc_summary.txt): 1707 VALU instructions per wavefront-iteration, "
                          "two wavefronts per SIMD keep its fp64 VALU ~63 % busy (k_tail, one per SIMD: ~31 %); "
                          "10 % of the wavefront cycles wait on LDS; HBM ~0.6 TB/s",
             "lean_launches_per_step": st["lean_launches"], "lean_escaped_last_step": st["lean_escaped"],
             "decade_slots_ms_last_step": st["hslots_ms"]})
        dominant, other = (k_tail, k_solve) if tail_iters >= solve_iters else (k_solve, k_tail)
        line = {
            "metric": "IK solves/sec to 1e-6 residual, Talos humanoid, batch=65536 per GPU",
            "value": total_solved * args.steps / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": wl["name"],
                "robot": "talos32 (fixed base, 32 x 1-DoF, Talos topology)",
                "batch_per_gpu": B,
                "task": "6-D velocity task on arm_left_7_joint, A=I, b=J(q) nu*, nu*~U(-0.5,0.5)^32, box +-0.5",
                "stop": "tol_abs=1e-6, tol_rel=0, max_iter=1000, reference fixture rho/mu/scale, adaptive mu (DEFAULT)",
                "parallelism": "%d independent shard(s), no collective" % world,
                "solved_fraction": total_solved / total_B,
                "flagged_infeasible_fraction_rank0": stats["infeasible"] / B,
                "hit_max_iter_fraction_rank0": stats["unfinished"] / B,
                "mean_admm_iterations": total_iters / total_B,
                "instance_iterations_per_s": total_iters * args.steps / elapsed,
                "solve_init_s_rank0_incl_pcie": t_init,
            },
            "roofline": dict(
                {"bound": "hbm",
                 "bytes_per_unit": bytes_iter,
                 "unit_def": "one ADMM iteration of one instance: 8 B x (203 nb + 108 nc), nb=32, nc=1 (the three-sweep "
                             "streaming model of SURVEY.md 8(d))",
                 "achieved_def": "algorithmic bytes of all launches of the kernel / time with >= 1 of its launches "
                                 "executing (= bytes per launch / average launch duration x launch_concurrency); "
                                 "frac > 1 means the kernel does not move the model's bytes: it keeps state on chip"},
                **dominant,
                **{"other_kernel": other,
                   "concurrent_chunks": st["chunks"],
                   "concurrency_note": "the batch is solved as %d independent chunk(s), each on its own stream; "
                                       "avg_launch_ms is per launch as timed by HIP events (launches of the chunks "
                                       "overlap and slow each other down), `aggregate` is all kernels' algorithmic bytes "
                                       "against the stream wall time of the whole Solve()" % st["chunks"],
                   "aggregate_algorithmic_GBps": inst_iters * bytes_iter / (solve_wall_ms * 1e-3) / 1e9 if solve_wall_ms > 0 else None,
                   "aggregate_frac_of_peak": inst_iters * bytes_iter / (solve_wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if solve_wall_ms > 0 else None}),
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(wl)
            except Exception as e:  # the GPU number must survive a broken host toolchain
                line["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": "failed: %r" % (e,)}
        print(json.dumps(line), flush=True)
    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
