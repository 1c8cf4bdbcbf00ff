#!/bin/bash
# LDS-pipe occupancy of the whole-body kernel (k_flat1, talos44, four tasks, 65 536 instances): SQ_LDS_IDX_ACTIVE over the cycles the dispatch took
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/wb_lds; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for C in "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  D=$O/pmc_$(echo $C | tr ' ' '_' | cut -c1-40); mkdir -p $D
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o pmc -- python $R/scripts/r03/quick_wholebody.py 65536 4 > $D/log.txt 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(O + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = 'k_flat1' if 'k_flat1' in r['Kernel_Name'] else 'k_fslots' if 'k_fslots' in r['Kernel_Name'] else None
        if k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
    if 'GRBM_GUI_ACTIVE' in d:
        cyc = sum(d['GRBM_GUI_ACTIVE']) / len(d['GRBM_GUI_ACTIVE']) / 8
        print('  cycles per dispatch', cyc, ' lds pipe busy', sum(d['SQ_LDS_IDX_ACTIVE']) / len(d['SQ_LDS_IDX_ACTIVE']) / (256 * cyc))
PY
