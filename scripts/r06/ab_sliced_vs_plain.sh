#!/bin/bash
# arrival order: the plain build (no slices) against the time-sliced one, full table / lazily populated table; lone iteration
cd ${GRAFT_REPO_ROOT:-.}
TAG="[lone]" python scripts/r04/lone.py
for B in 65536 262144; do
TAG="[plain, full table]" LOIKB_FLAT_ORDER=0 LOIKB_FLAT_SLICE=0 LOIKB_FLAT_BUILD=0 python scripts/r03/quick_headline.py $B 6 | tail -1
TAG="[sliced 288/96, full table]" LOIKB_FLAT_ORDER=0 LOIKB_FLAT_SLICE=288 LOIKB_FLAT_SLICE2=96 LOIKB_FLAT_BUILD=0 python scripts/r03/quick_headline.py $B 6 | tail -1
TAG="[sliced 2000 (never ends), full table]" LOIKB_FLAT_ORDER=0 LOIKB_FLAT_SLICE=2000 LOIKB_FLAT_BUILD=0 python scripts/r03/quick_headline.py $B 6 | tail -1
TAG="[sliced 288/96, lazy table]" LOIKB_FLAT_ORDER=0 LOIKB_FLAT_SLICE=288 LOIKB_FLAT_SLICE2=96 python scripts/r03/quick_headline.py $B 6 | tail -1
TAG="[sliced 2000 (never ends), lazy table]" LOIKB_FLAT_ORDER=0 LOIKB_FLAT_SLICE=2000 python scripts/r03/quick_headline.py $B 6 | tail -1
done
