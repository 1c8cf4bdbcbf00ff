#!/bin/bash
# counters of k_flat2 on scripts/r06/pairing_probe.py (2048 copies of one 999-iteration instance: the iteration loop and nothing else),
# the plain build against the time-sliced one -- separate --pmc passes, kernel trace only
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
export PROBE_N=${PROBE_N:-2048} LOIKB_FLAT_ORDER=0
for mode in "plain:0" "sliced:2000"; do
  name=${mode%%:*}; export LOIKB_FLAT_SLICE=${mode##*:}
  echo "=== $name"
  for C in ${PMC_SETS:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"} "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_IFETCH SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" "SQ_WAIT_ANY SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_SENDMSG"; do
    D=$R/gpurun_out/pmc_loop/$name/$(echo $C | tr ' ' '_' | cut -c1-60); rm -rf $D; mkdir -p $D
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o pmc -- python $R/scripts/r06/pairing_probe.py > $D/log.txt 2>&1
    python - "$D" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(float); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'k_flat2' not in r['Kernel_Name'] or int(r["Grid_Size"]) != int(__import__("os").environ.get("PROBE_N","2048")) * 64: continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print({c: v / n[c] for c, v in acc.items()}, 'dispatches', max(n.values()) if n else 0)
PY
  done
done
find $R/gpurun_out/pmc_loop -name "*.csv" -size +256k -delete 2>/dev/null
