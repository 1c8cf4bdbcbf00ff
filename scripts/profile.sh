#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (run on the GPU box via gpurun).
# usage: scripts/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_under_rocprof.log" 2>&1
echo "rocprofv3 exit: $?"
find "$OUT" -name "*stats*.csv" | head
