#!/bin/bash
# the decade table against the in-wave builder on the headline batch (one gpurun call): full table | narrow tables + builder
cd ${GRAFT_REPO_ROOT:-.}
run() { TAG="[$1]" python scripts/r03/quick_headline.py 65536 6 | tail -1; TAG="[$1 arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_headline.py 65536 6 | tail -1; }
run "table (default)"
LOIKB_FLAT_BUILD=1 run "builder compiled in, default table"
for w in "0 4" "0 3" "0 2" "1 2"; do
  set -- $w
  LOIKB_FLAT_BUILD=1 LOIKB_LEAN_KLO=$1 LOIKB_LEAN_DECADES=$2 LOIKB_LEAN_ADAPT=0 run "builder, table $1..+$2"
done
