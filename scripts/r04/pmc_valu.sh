#!/bin/bash
# VALU busy / issue counters of k_flat2 on a bulk-dominated batch (4x the headline): is the bulk regime VALU-bound?
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_valu; mkdir -p $O
for C in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  D=$O/$(echo $C | tr ' ' '_' | cut -c1-50); mkdir -p $D
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o pmc -- python $R/scripts/r03/quick_headline.py ${B:-262144} 3 > $D/log.txt 2>&1
  python - "$D" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('<')[0].split('(')[0][-20:]
        if 'k_flat2' not in r['Kernel_Name'] and 'k_fslots' not in r['Kernel_Name']: continue
        k = 'k_flat2' if 'k_flat2' in r['Kernel_Name'] else 'k_fslots'
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in acc:
    print(k, {c: v / n[(k, c)] for c, v in acc[k].items()}, 'dispatches', max(n[(k, c)] for c in acc[k]))
PY
done
