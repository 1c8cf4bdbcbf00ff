#!/bin/bash
# first / later slice lengths of the time-sliced launch on the headline batch in arrival order (one gpurun call)
cd ${GRAFT_REPO_ROOT:-.}
for s in "288 96" "192 96" "224 64" "256 64" "288 64" "288 128" "352 96" "416 96" "288 48" "160 160"; do
  set -- $s
  TAG="[slice $1 / $2]" LOIKB_FLAT_ORDER=0 LOIKB_FLAT_SLICE=$1 LOIKB_FLAT_SLICE2=$2 python scripts/r03/quick_headline.py 65536 7 | tail -1
done
TAG="[default]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_headline.py 65536 7 | tail -1
