#!/bin/bash
# A/B of build flags inside ONE gpurun call: ab_flags.sh "<flags A>" "<flags B>" ...   (lone instance, headline ordered / arrival order, 4x batch)
cd ${GRAFT_REPO_ROOT:-.}
for f in "$@"; do
  python -c "from loik_amd import _build; _build.build(force=True, extra_flags=tuple('$f'.split()))" > /dev/null 2>&1 || echo "build failed: $f"
  TAG="[$f]" python scripts/r04/lone.py
  TAG="[$f ordered]" python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[$f arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[$f]" python scripts/r03/quick_headline.py 262144 4 | tail -1
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
