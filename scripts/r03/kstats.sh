#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short headline run; prints the per-kernel table.  usage: kstats.sh [B] [solves]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/kstats
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python "$REPO/scripts/r03/quick_headline.py" ${1:-65536} ${2:-4} > "$OUT/log.txt" 2>&1
echo "rocprof exit $?"; tail -2 "$OUT/log.txt"
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("%-44s calls %4s  avg %10.3f ms  total %10.3f ms  %5s %%" % (r["Name"].split("(")[0].replace("void ", "").replace("loikb::", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
else echo "no kernel_stats.csv"; fi
rm -rf "$OUT/trace"
