#!/bin/bash
# tail re-balancing (LOIKB_REBALANCE) on arrival-order, time-sliced launches: A/B inside ONE gpurun call
cd ${GRAFT_REPO_ROOT:-.}
for f in "-DLOIKB_REBALANCE=0" "-DLOIKB_REBALANCE=1"; do
  python -c "from loik_amd import _build; _build.build(force=True, extra_flags=tuple('$f'.split()))" > /dev/null 2>&1 || echo "build failed: $f"
  for B in 32768 65536 131072; do
    TAG="[$f arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_headline.py $B 7 | tail -1
  done
  TAG="[$f arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_wholebody.py 65536 5
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
