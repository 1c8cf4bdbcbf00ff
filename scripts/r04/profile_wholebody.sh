#!/bin/bash
# kernel stats + PMC (HBM bytes, VALU / LDS counters) of the whole-body batch (talos44, four tasks, 65 536 instances: k_fslots + k_flat1),
# as scripts/profile_round.sh does for the headline.  usage: scripts/r04/profile_wholebody.sh <tag>
TAG=${1:-r04_d}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${TAG}_wholebody; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for MODE in fresh repeat; do
  [ $MODE = fresh ] && export LOIKB_FLAT_ORDER=0 || unset LOIKB_FLAT_ORDER
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$MODE -o trace -- python $R/scripts/r03/quick_wholebody.py 65536 6 > $O/log_$MODE.txt 2>&1
  find $O/trace_$MODE -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$MODE.csv \;
  rm -rf $O/trace_$MODE
done
unset LOIKB_FLAT_ORDER
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"; do
  D=$O/pmc_$(echo $C | tr ' ' '_' | cut -c1-40); mkdir -p $D
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o pmc -- python $R/scripts/r03/quick_wholebody.py 65536 4 > $D/log.txt 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, json, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for fn in glob.glob(O + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = 'k_flat1' if 'k_flat1' in r['Kernel_Name'] else 'k_fslots' if 'k_fslots' in r['Kernel_Name'] else None
        if k: acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
out = {"workload": "talos44 whole body, four tasks, 65 536 instances (scripts/r03/quick_wholebody.py: one handle, the batch repeated -- the first launch in arrival order with time slices, the later ones ordered)",
       "units": "per dispatch, averaged over the profiled dispatches; FETCH_SIZE doubled (gfx950), KiB -> bytes", "kernels": {}}
for k, d in acc.items():
    e = {c: d[c] / n[k][c] for c in d}
    o = {"dispatches": max(n[k].values())}
    if "FETCH_SIZE" in e: o["fetch_bytes"] = 2 * 1024 * e["FETCH_SIZE"]
    if "WRITE_SIZE" in e: o["write_bytes"] = 1024 * e["WRITE_SIZE"]
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY"):
        if c in e: o[c] = e[c]
    if e.get("SQ_ACTIVE_INST_LDS"): o["lds_bank_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_ACTIVE_INST_LDS"]
    out["kernels"][k] = o
print(json.dumps(out, indent=1))
PY
