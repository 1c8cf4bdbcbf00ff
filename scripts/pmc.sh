#!/bin/bash
# usage: scripts/pmc.sh <tag> "<counters>" <cmd...>   -- PMC pass (kernel-trace only, no other trace domains)
TAG=$1; shift; CNT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT" -o pmc -- "$@" > "$OUT/log.txt" 2>&1
echo "exit $?"; ls "$OUT"
