"""Summarise the PMC passes of scripts/profile_round.sh into one JSON (per kernel, per Solve() step of the profiled bench run).
HBM bytes as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE from separate passes, KiB units,
FETCH_SIZE doubled on gfx950 (it reports half the bytes of wide coalesced streaming reads); WRITE_SIZE uncorrected."""
import collections
import csv
import glob
import json
import os
import sys

out_dir, nsolves = sys.argv[1], int(sys.argv[2])  # nsolves = warmup + steps of every profiled run
KEYS = ["k_solve", "k_tail", "k_lean", "k_hslots", "k_flat2", "k_flat", "k_fslots", "k_move", "k_sched"]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(lambda: collections.defaultdict(int))
for path in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        key = next((k for k in KEYS if k in name), None)
        if key is None:
            continue
        tot[key][row["Counter_Name"]] += float(row["Counter_Value"])
        ndisp[key][row["Counter_Name"]] += 1
out = {"source": "rocprofv3 --kernel-trace --pmc, separate passes (scripts/profile_round.sh); FETCH_SIZE/WRITE_SIZE in KiB, "
                 "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE uncorrected",
       "batch": 65536, "solves_profiled": nsolves, "compute_units": 256, "shader_clock_ghz": 2.4, "kernels": {}}
for k, d in tot.items():
    e = {}
    anyc = next(iter(d))
    e["dispatches_per_step"] = ndisp[k][anyc] / nsolves
    if "FETCH_SIZE" in d:
        e["fetch_bytes_per_step"] = 2.0 * d["FETCH_SIZE"] * 1024 / nsolves
    if "WRITE_SIZE" in d:
        e["write_bytes_per_step"] = d["WRITE_SIZE"] * 1024 / nsolves
    for c, label in [("SQ_INSTS_VALU", "valu_insts_per_step"), ("SQ_INSTS_SALU", "salu_insts_per_step"),
                     ("SQ_INSTS_LDS", "lds_insts_per_step"), ("SQ_WAVE_CYCLES", "wave_cycles_per_step"),
                     ("SQ_BUSY_CYCLES", "busy_cycles_per_step"), ("SQ_WAIT_INST_LDS", "wait_inst_lds_per_step"),
                     ("SQ_ACTIVE_INST_VALU", "valu_active_quadcycles_per_step"), ("SQ_ACTIVE_INST_ANY", "any_active_quadcycles_per_step"),
                     ("SQ_WAIT_ANY", "wait_any_per_step"), ("SQ_WAIT_INST_ANY", "wait_inst_any_per_step"),
                     ("GRBM_GUI_ACTIVE", "gui_active_cycles_x8_per_step"), ("SQ_LDS_IDX_ACTIVE", "lds_idx_active_cycles_per_step"),
                     ("SQ_ACTIVE_INST_LDS", "lds_active_quadcycles_per_step")]:
        if c in d:
            e[label] = d[c] / nsolves
    if "SQ_ACTIVE_INST_LDS" in d and d["SQ_ACTIVE_INST_LDS"] > 0 and "SQ_LDS_BANK_CONFLICT" in d:
        e["lds_bank_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_ACTIVE_INST_LDS"]
    out["kernels"][k] = e
# per DISPATCH figures (the bench's default line runs W + K handles, each with one history solve before its timed one: dispatches
# per "step" of the profiled command is not 1) and instructions per instance-iteration of the dominant kernel
# GRBM_GUI_ACTIVE is summed over the eight XCDs: / 8 = the shader cycles the dispatch took AT THE CLOCK IT RAN AT (a launch that keeps
# every SIMD on fp64 work runs at ~2.06 GHz, not at the 2.4 GHz of the data sheet: the busy fractions below are of THOSE cycles)
for k, e in out["kernels"].items():
    n = max(e.get("dispatches_per_step", 1.0) * nsolves, 1.0)
    for src, dst in (("fetch_bytes_per_step", "fetch_bytes_per_dispatch"), ("write_bytes_per_step", "write_bytes_per_dispatch"),
                     ("valu_insts_per_step", "valu_insts_per_dispatch"), ("lds_insts_per_step", "lds_insts_per_dispatch"),
                     ("salu_insts_per_step", "salu_insts_per_dispatch"), ("valu_active_quadcycles_per_step", "valu_active_quadcycles_per_dispatch"),
                     ("wave_cycles_per_step", "wave_cycles_per_dispatch"), ("wait_any_per_step", "wait_any_per_dispatch"),
                     ("gui_active_cycles_x8_per_step", "gui_active_cycles_x8_per_dispatch"),
                     ("lds_idx_active_cycles_per_step", "lds_idx_active_cycles_per_dispatch"),
                     ("lds_active_quadcycles_per_step", "lds_active_quadcycles_per_dispatch")):
        if src in e:
            e[dst] = e[src] * nsolves / n
for k, e in out["kernels"].items():
    if "gui_active_cycles_x8_per_dispatch" in e:
        cyc = e["gui_active_cycles_x8_per_dispatch"] / 8.0
        e["shader_cycles_per_dispatch"] = cyc
        simd_cycles = 4 * out["compute_units"] * cyc
        if "valu_active_quadcycles_per_dispatch" in e:
            e["valu_busy_frac_of_actual_cycles"] = e["valu_active_quadcycles_per_dispatch"] * 4.0 / simd_cycles
        if "lds_idx_active_cycles_per_dispatch" in e:
            e["lds_pipe_busy_frac_of_actual_cycles"] = e["lds_idx_active_cycles_per_dispatch"] / (out["compute_units"] * cyc)
        if "wait_any_per_dispatch" in e and "wave_cycles_per_dispatch" in e:
            e["wave_time_in_waitcnt_frac"] = e["wait_any_per_dispatch"] / e["wave_cycles_per_dispatch"]
try:
    line = json.loads(open(os.path.join(out_dir, "bench_line.json")).read().strip().splitlines()[-1])
    out["bench_ms_per_step"] = line["ms_per_step"]
    out["instance_iterations_per_launch"] = line["roofline"]["units_per_launch"]
    kl = out["kernels"].get(line["roofline"]["kernel"])
    if kl and "valu_insts_per_dispatch" in kl:
        kl["valu_insts_per_instance_iteration"] = kl["valu_insts_per_dispatch"] / out["instance_iterations_per_launch"]
        if "lds_insts_per_dispatch" in kl:
            kl["lds_insts_per_instance_iteration"] = kl["lds_insts_per_dispatch"] / out["instance_iterations_per_launch"]
except Exception as e:  # noqa: BLE001
    out["bench_line_note"] = "no bench line: %r" % (e,)
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["csrc_sha16"] = bench.csrc_sha16()
except Exception as e:  # noqa: BLE001
    out["csrc_sha16"] = None
print(json.dumps(out, indent=1))
