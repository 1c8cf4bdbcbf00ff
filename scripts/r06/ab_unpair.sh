#!/bin/bash
# one long runner per SIMD while a time-sliced launch runs out (LOIKB_FLAT_UNPAIR): off / on, arrival order, several batch sizes
cd ${GRAFT_REPO_ROOT:-.}
for B in ${BATCHES:-65536 32768 131072}; do
  for u in 0 1 0 1; do
    TAG="[unpair $u]" LOIKB_FLAT_UNPAIR=$u LOIKB_FLAT_ORDER=0 timeout 120 python scripts/r03/quick_headline.py $B 7 | tail -1
  done
done
