#!/bin/bash
# the decade table against the lazily populated one on the headline batch (one gpurun call)
cd ${GRAFT_REPO_ROOT:-.}
run() { TAG="[$1 arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_headline.py 65536 7 | tail -1; }
run "table (default)"
LOIKB_FLAT_BUILD=1 run "lazy table compiled in, whole table built"
for w in "0,5" "0,4" "0,3" "0,2" "-1,5"; do
  LOIKB_FLAT_BUILD=1 LOIKB_FLAT_WINDOW=$w run "lazy table, k_fslots builds $w"
done
run "table (default) again"
