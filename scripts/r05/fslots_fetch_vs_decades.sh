#!/bin/bash
# what k_fslots fetches: HBM bytes per dispatch against the number of decades it builds (the records do not depend on it, the table does)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/dgrp; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export LOIKB_LEAN_ADAPT=0 LOIKB_LEAN_KLO=0
for nd in 1 2 4 8; do
  export LOIKB_LEAN_DECADES=$nd
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/p; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p -o pmc -- python $R/bench.py --no-cpu-baseline --no-variants --steps 2 --warmup 1 > /dev/null 2>&1
    python - $O/p $c $nd <<'PY'
import csv, glob, sys
v = [float(r["Counter_Value"]) for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(fn)) if "k_fslots" in r["Kernel_Name"]]
print("decades %s: %s per k_fslots dispatch: %.3f GB (%d dispatches)" % (sys.argv[3], sys.argv[2], sum(v) / max(1, len(v)) * 1024 * (2 if sys.argv[2] == "FETCH_SIZE" else 1) / 1e9, len(v)))
PY
  done
done
rm -rf $O/p
