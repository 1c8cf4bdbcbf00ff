#!/bin/bash
# PMC passes (rocprofv3 --kernel-trace --pmc, one counter group per pass) of a short headline run; per-kernel sums.
# usage: pmc.sh [B] [solves]     env: GROUPS="..." (semicolon-separated counter groups)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_r03
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B=${1:-65536}; N=${2:-3}
IFS=';' read -ra GR <<< "${GROUPS_PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES;SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY}"
i=0
for C in "${GR[@]}"; do
  D=$OUT/p$i; mkdir -p "$D"; i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$D" -o pmc -- python "$REPO/scripts/r03/quick_headline.py" $B $N > "$D/log.txt" 2>&1
  echo "pmc [$C] exit $?"
done
python - "$OUT" $N <<'PY'
import csv, glob, os, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for path in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("loikb::", "")
        if not name.startswith("k_f"): continue
        tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
for k, d in sorted(tot.items()):
    print(k, {c: "%.4g" % (v / n) for c, v in sorted(d.items())})
PY
find "$OUT" -name "*.csv" -size +1M -delete
