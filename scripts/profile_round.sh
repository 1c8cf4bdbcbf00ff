#!/bin/bash
# One call on the GPU box -> everything profiles/ needs for a round tag:
#   gpurun_out/<tag>/bench_line.json     python bench.py (default flags)
#   gpurun_out/<tag>/kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>/pmc.json            HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes, as MI355X_MICROARCH.md
#                                        prescribes) + VALU / LDS counters per kernel and Solve() step
# PMC passes run with --kernel-trace only (no other trace domain).  usage: scripts/profile_round.sh <tag> [bench args]
set -u
TAG=${1:-r03}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 900 python "$REPO/bench.py" "$@" > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench exit $?"; cat "$OUT/bench_line.json"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python "$REPO/bench.py" --no-cpu-baseline --no-variants "$@" > "$OUT/bench_under_rocprof.log" 2>&1
echo "trace exit $?"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
STEPS=3
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_SMEM"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  mkdir -p "$D"
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$D" -o pmc -- python "$REPO/bench.py" --no-cpu-baseline --no-variants --steps 2 --warmup 1 "$@" > "$D/log.txt" 2>&1
  echo "pmc [$C] exit $?"
done
python "$REPO/scripts/pmc_round_summary.py" "$OUT" $STEPS > "$OUT/pmc.json"
cat "$OUT/pmc.json"
# phase timeline of wavefront 0 (a -DLOIKB_TAIL_PROF build, rebuilt back afterwards)
cd "$REPO" && timeout 600 bash scripts/r03/prof_build.sh > "$OUT/k_flat_phase_timeline.txt" 2>&1 < /dev/null; cat "$OUT/k_flat_phase_timeline.txt"
# keep the merge-back small: the raw traces stay on the box
rm -rf "$OUT/trace"
find "$OUT" -name "*.csv" -size +2M -delete
