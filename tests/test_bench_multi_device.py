"""CPU-only: the host logic of `bench.py --gpus N` (one process, one host thread + solver handle per device, contiguous
shards, wall clock around all of them, sum of the solves) with a stand-in solver -- the CPU oracle -- in place of the HIP
library, which refuses to run without a GPU.  What is under test is bench.py's sharding / timing / aggregation code, the
same code the GPU run uses; the oracle is only the stand-in that makes it executable here."""
import json
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from loik_amd import workloads  # noqa: E402
from oracle import ref  # noqa: E402

CREATED = []


class FakeSolver:
    """BatchedLoik's surface as bench.py uses it, backed by the CPU oracle"""

    def __init__(self, model, batch, device=0, flags=0, max_launch_iters=0, **prm):
        self.model, self.B, self.device, self.prm = model, batch, device, prm
        self.thread = None
        self.nsolve = 0
        self.closed = False
        CREATED.append(self)

    def SolveInit(self, *a):
        self.args = a

    def Solve(self, *a):
        self.thread = threading.get_ident()
        if len(a) == 4:   # the tailored form Solve(q, c_id, Ai, bi): stand-in = a cold solve of that configuration and target
            q, c_id, Ai, bi = a
            args = list(self.args)
            args[0], args[4], args[5] = q, np.asarray(Ai).reshape(1, 6, 6), np.asarray(bi).reshape(self.B, 1, 6)
            self.args = tuple(args)
            self.tailored = getattr(self, "tailored", []) + [np.asarray(bi).copy()]
        self.out = ref.solve_batch(self.model, *self.args, nthreads=1, **dict(self.prm, warm_start=False))
        self.nsolve += 1

    def synchronize(self):
        pass

    def stats(self):
        it = int(self.out["iters"].sum())
        return dict(kernel_ms=1.0, tail_ms=1.0, tail_instances=self.B, tail_instance_iterations=it, tail_launches=1,
                    total_ms=1.0, solve_busy_ms=0.0, tail_busy_ms=1.0, instance_iterations=it, launches=1, hslots_ms=0.1,
                    lean_launches=1, flat_launches=1, flat_split_launches=1, flat_ordered=1, queue_dry_ms=0.6, bytes_per_instance_iteration=8.0 * (203 * 32 + 108), team=4, chunks=1,
                    lean_escaped=0)

    def get(self, name):
        return {"converged": self.out["converged"].astype(np.int32), "iter": self.out["iters"],
                "primal_infeasible": self.out["primal_infeasible"].astype(np.int32)}[name]

    def close(self):
        self.closed = True


def _run(argv, capsys, ndev=2):
    CREATED.clear()
    line = bench.main(argv, solver_factory=FakeSolver, device_count=ndev)
    printed = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(printed) == 1 and json.loads(printed[0])["value"] == pytest.approx(line["value"])
    return line


def test_gpus_2_weak_runs_two_devices_in_one_process(capsys, monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    line = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "24", "--no-cpu-baseline", "--repeat-batch"], capsys)
    weak = sorted(CREATED[:2], key=lambda s: s.device)   # (the shards are constructed concurrently, each on its own thread)
    assert [s.device for s in weak] == [0, 1] and all(s.B == 24 for s in weak)
    assert weak[0].thread != weak[1].thread          # one host thread per device
    assert all(s.nsolve == 3 for s in weak)          # warmup + steps
    assert all(s.closed for s in CREATED)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["batch_total"] == 48
    # different instances per device (seed + shard index), value = all shards' solves over the common wall time
    assert not np.array_equal(weak[0].args[0], weak[1].args[0])
    solved = sum(int(s.out["converged"].sum()) for s in weak)
    assert line["value"] == pytest.approx(solved * 2 / (line["ms_per_step"] * 2e-3))
    # the strong leg of the same run: the 1-GPU workload split contiguously
    st = line["strong_scaling"]
    strong = sorted(CREATED[2:], key=lambda s: s.device)
    assert len(strong) == 2 and [s.B for s in strong] == [12, 12] and st["batch_total"] == 24
    full = workloads.talos_c3(24, seed=0x101C + 3)
    assert np.array_equal(np.concatenate([s.args[0] for s in strong]), full["q"])
    assert line["roofline"]["bound"] == "fp64_valu" and line["roofline"]["frac"] < 1.0
    assert "cpu_baseline" not in line
    # both legs are labelled objects of the line; the strong one carries BASELINE.json's wording
    wk = line["weak_scaling"]
    assert wk["value"] == line["value"] and wk["batch_total"] == 48 and "per GPU" in wk["metric"]
    # every device's launch split into bulk and the chain of its long runners, for both legs
    assert [c["gpu"] for c in line["per_shard"]] == [0, 1] and all(c["batch"] == 24 and "lone_chain_ms" in c for c in line["per_shard"])
    assert [c["batch"] for c in st["per_shard"]] == [12, 12]
    assert st["n_gpus"] == 2 and "batch=24 in total" in st["metric"]


def test_gpus_1_and_strong_flag(capsys, monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    line = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--batch", "16", "--no-cpu-baseline", "--repeat-batch"], capsys, ndev=1)
    assert line["n_gpus"] == 1 and "strong_scaling" not in line and len(CREATED) == 1
    line = _run(["--gpus", "4", "--steps", "1", "--warmup", "0", "--batch", "32", "--scaling", "strong",
                 "--no-cpu-baseline", "--repeat-batch"], capsys, ndev=4)
    assert line["scaling"] == "strong" and line["config"]["batch_per_gpu"] == 8 and line["config"]["batch_total"] == 32
    assert [s.device for s in CREATED] == [0, 1, 2, 3]


def test_default_line_solves_a_fresh_batch_every_step(capsys, monkeypatch):
    """`value` is what a caller with NEW problems gets -- every step (warm-up steps included) is one handle's first
    solve of a batch no handle has solved: W + K handles per shard, each with one earlier solve of another batch (its history) and
    its own batch resident before the timed region; value = the steps' solved instances over the wall time"""
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    line = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "16", "--no-cpu-baseline", "--no-strong-leg"], capsys)
    assert len(CREATED) == 2 * 4 and all(s.nsolve == 2 for s in CREATED) and all(s.closed for s in CREATED)
    by_dev = {d: [s for s in CREATED if s.device == d] for d in (0, 1)}
    qs = [s.args[0] for s in CREATED]
    for i in range(len(qs)):                      # every handle ended on a batch of its own
        for j in range(i + 1, len(qs)):
            assert not np.array_equal(qs[i], qs[j])
    # the timed steps are the last three handles of each shard; their solved counts make up `value`
    solved = sum(int(s.out["converged"].sum()) for d in (0, 1) for s in by_dev[d][1:])
    assert line["value"] == pytest.approx(solved / (line["ms_per_step"] * 3e-3))
    assert "FIRST solve" in line["config"]["schedule"]


def test_config_c4_runs_tailored_steps_over_its_targets(capsys, monkeypatch):
    """--config c4 (BASELINE.json config 4): every shard is one GPU's share, a step is the tailored Solve(q_t, c_id, Ai, b_t) on target
    k mod 4 of the share's own sequence, value = the steps' solved instances over the wall time, no variants, weak by construction"""
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    line = _run(["--config", "c4", "--gpus", "2", "--steps", "5", "--warmup", "1", "--batch", "12", "--no-cpu-baseline"], capsys)
    assert len(CREATED) == 2 and all(s.B == 12 and s.nsolve == 6 for s in CREATED)
    assert line["config"]["bench_config"] == "c4" and "tailored" in line["metric"] and "strong_scaling" not in line and "whole_body_variant" not in line
    for s in CREATED:
        assert len(s.tailored) == 6
        assert np.array_equal(s.tailored[0], s.tailored[4]) and not np.array_equal(s.tailored[0], s.tailored[1])   # T = 4, cycled
    assert not np.array_equal(CREATED[0].tailored[0], CREATED[1].tailored[0])                                     # every share its own sequence
    assert line["config"]["batch_total"] == 24 and 0.0 <= line["config"]["solved_fraction"] <= 1.0


def test_more_shards_than_devices_needs_opt_in(capsys, monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("LOIKB_ALLOW_SHARED_GPU", raising=False)
    with pytest.raises(SystemExit):
        bench.main(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8", "--no-cpu-baseline"],
                   solver_factory=FakeSolver, device_count=1)
    monkeypatch.setenv("LOIKB_ALLOW_SHARED_GPU", "1")
    line = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8", "--no-cpu-baseline", "--no-strong-leg", "--repeat-batch"],
                capsys, ndev=1)
    assert [s.device for s in CREATED] == [0, 0] and line["config"]["shared_gpu_smoke_test"] is True


def test_effective_cpus_respects_affinity():
    n, note = bench.effective_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0)) and "affinity" in note


def test_cpu_baseline_is_self_consistent():
    """threads x single-thread rate and the measured multi-thread rate agree within 2x (the round-1 line did not)"""
    wl = workloads.talos_c3(512, seed=5)
    for attempt in range(3):   # (a wall-clock measurement on a shared host: one noisy sample is not a verdict)
        cb = bench.cpu_baseline(wl, budget_s=1.0)
        assert cb["kind"] == "port" and cb["cores"] == bench.effective_cpus()[0]
        if cb["consistent"]:
            break
    assert cb["consistent"], cb
