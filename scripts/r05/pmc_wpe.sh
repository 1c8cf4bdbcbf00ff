#!/bin/bash
# counters of k_flat2 on a bulk-dominated batch for the two / three wavefronts-per-SIMD builds (one gpurun call)
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in ${WPES:-2 3}; do
  export LOIKB_FLAT_WPE=$w
  echo "=== LOIKB_FLAT_WPE=$w"
  bash $R/scripts/r04/pmc_valu.sh 2>&1 | grep -v "^$"
  cd /tmp; export TMPDIR=/tmp
  O=$R/gpurun_out/pmc_valu_x; mkdir -p $O
  for C in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_IFETCH SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_FLAT SQ_INSTS_VALU"; do
    D=$O/$(echo $C | tr ' ' '_' | cut -c1-50); rm -rf $D; mkdir -p $D
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o pmc -- python $R/scripts/r03/quick_headline.py ${B:-262144} 3 > $D/log.txt 2>&1
    python - "$D" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'k_flat2' not in r['Kernel_Name']: continue
        acc['k_flat2'][r['Counter_Name']] += float(r['Counter_Value']); n[('k_flat2', r['Counter_Name'])] += 1
for k in acc:
    print(k, {c: v / n[(k, c)] for c, v in acc[k].items()})
PY
  done
  find $R/gpurun_out/pmc_valu $O -name "*.csv" -size +1M -delete 2>/dev/null
done
