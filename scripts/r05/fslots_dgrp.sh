#!/bin/bash
# k_fslots with the decades taken through its two passes g at a time (LOIKB_FSLOT_DGRP): time, HBM bytes fetched / written per dispatch
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/dgrp; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for g in 0 1 2 3 4; do
  export LOIKB_FSLOT_DGRP=$g
  timeout 300 python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('dgrp $g: ms_per_step %.3f  k_flat2 %.3f  k_fslots %.3f ms' % (d['ms_per_step'], r['avg_launch_ms'], r['other_kernel']['avg_launch_ms']))"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/p; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p -o pmc -- python $R/bench.py --no-cpu-baseline --no-variants --steps 2 --warmup 1 > /dev/null 2>&1
    python - $O/p $c <<'PY'
import csv, glob, sys
v = [float(r["Counter_Value"]) for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(fn)) if "k_fslots" in r["Kernel_Name"]]
print("   %s per k_fslots dispatch: %.3f GB" % (sys.argv[2], sum(v) / len(v) * 1024 * (2 if sys.argv[2] == "FETCH_SIZE" else 1) / 1e9))
PY
  done
done
rm -rf $O/p
