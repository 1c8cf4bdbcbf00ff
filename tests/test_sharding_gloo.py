"""CPU-only, world_size 2 over gloo: the N > 1 path of bench.py (contiguous batch shards, no data-path collective,
barrier + max-of-time + sum-of-counters aggregation).  The per-shard solver here is the CPU oracle standing in for
the GPU; what is under test is the host logic that the GPU run uses unchanged (loik_amd/sharding.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from loik_amd import sharding, workloads
    from oracle import ref
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = workloads.talos_c3(96, seed=123)
    prm = dict(wl["params"], max_iter=60)
    sh = sharding.shard_workload(wl, rank, world)
    dist.barrier()
    out = ref.solve_batch(wl["model"], sh["q"], sh["H_ref"], sh["v_ref"], sh["c_ids"], sh["Ais"], sh["bis"], sh["lb"],
                          sh["ub"], nthreads=1, **prm)
    dist.barrier()
    elapsed, tot = sharding.aggregate(dist, 1.0 + rank, dict(solved=int(out["converged"].sum()),
                                                             iters=int(out["iters"].sum()), batch=sh["q"].shape[0]))
    lo, hi = sharding.shard_bounds(96, rank, world)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), z=out["z"], lo=lo, hi=hi, elapsed=elapsed, **tot)
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from loik_amd import sharding, workloads
    from oracle import ref
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    wl = workloads.talos_c3(96, seed=123)
    prm = dict(wl["params"], max_iter=60)
    full = ref.solve_batch(wl["model"], wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"],
                           wl["ub"], nthreads=2, **prm)
    z = np.zeros_like(full["z"])
    covered = np.zeros(96, dtype=int)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        z[int(d["lo"]):int(d["hi"])] = d["z"]
        covered[int(d["lo"]):int(d["hi"])] += 1
        # every rank sees the same aggregate: max of the times, sums of the counters
        assert float(d["elapsed"]) == 2.0
        assert int(d["solved"]) == int(full["converged"].sum())
        assert int(d["iters"]) == int(full["iters"].sum())
        assert int(d["batch"]) == 96
    assert np.all(covered == 1)          # shards tile the batch exactly once
    assert np.array_equal(z, full["z"])  # no exchange step: a shard's answers do not depend on the other shard


def test_shard_bounds_tile_any_batch():
    from loik_amd import sharding
    for total in (1, 7, 64, 65536, 1048576 + 3):
        for world in (1, 2, 4, 8):
            edges = [sharding.shard_bounds(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
