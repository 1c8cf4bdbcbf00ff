#!/bin/bash
# the flip-only fast path of the time-sliced builds: k_flat2's / k_flat1's, on and off (arrival order: the builds that have it)
cd ${GRAFT_REPO_ROOT:-.}
for f in none "-DLOIKB_FLIP_FAST=0 -DLOIKB_FLIP_FAST1=0" none "-DLOIKB_FLIP_FAST=0 -DLOIKB_FLIP_FAST1=0"; do
  ff="$f"; [ "$f" = none ] && ff=""
  python -c "from loik_amd import _build; _build.build(force=True, flat_flags=_build.FLAT_FLAGS + '$ff'.split())" > /dev/null 2>&1 || echo "build failed: $f"
  TAG="[$f arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_headline.py 65536 8 | tail -1
  TAG="[$f arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_headline.py 131072 5 | tail -1
  TAG="[$f arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_wholebody.py 65536 8 | tail -1
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
