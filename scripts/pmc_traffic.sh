#!/bin/bash
# HBM traffic of one bench step from the PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), kernel-trace only; units are KiB; on gfx950 FETCH_SIZE
# reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled here.
# usage (GPU box): scripts/pmc_traffic.sh <tag>     -> gpurun_out/pmc_<tag>_{fetch,write}/ + profiles-ready JSON
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$REPO/gpurun_out/pmc_${TAG}_$C
  mkdir -p "$OUT"
  (cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT" -o pmc -- python "$REPO/bench.py" --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/log.txt" 2>&1)
  echo "$C exit $?"
done
python "$REPO/scripts/pmc_traffic_summary.py" "$REPO/gpurun_out/pmc_${TAG}_FETCH_SIZE/pmc_counter_collection.csv" "$REPO/gpurun_out/pmc_${TAG}_WRITE_SIZE/pmc_counter_collection.csv" 3 > "$REPO/gpurun_out/traffic_${TAG}.json"
cat "$REPO/gpurun_out/traffic_${TAG}.json"
