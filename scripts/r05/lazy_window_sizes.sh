#!/bin/bash
# the lazily populated decade table (k_flat2<.., MUR = 2>, window of five decades from mu0's) against the full table, across batch sizes and schedules
cd ${GRAFT_REPO_ROOT:-.}
for B in 16384 32768 65536 131072 262144; do
  for mode in arrival ordered; do
    [ $mode = arrival ] && export LOIKB_FLAT_ORDER=0 || unset LOIKB_FLAT_ORDER
    TAG="[table $mode]" python scripts/r03/quick_headline.py $B 5 | tail -1
    TAG="[lazy 0,5 $mode]" LOIKB_FLAT_BUILD=1 LOIKB_FLAT_WINDOW=0,5 python scripts/r03/quick_headline.py $B 5 | tail -1
  done
done
