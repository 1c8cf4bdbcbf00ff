#!/bin/bash
# A/B of build variants inside ONE gpurun call, several scripts per build: ab_multi.sh "<flags A>" "<flags B>" ...
cd ${GRAFT_REPO_ROOT:-.}
for f in "$@"; do
  python -c "from loik_amd import _build; _build.build(force=True, extra_flags=tuple('$f'.split()))" > /dev/null 2>&1 || echo "build failed: $f"
  TAG="[$f]" python scripts/r04/lone.py
  TAG="[$f]" python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[$f arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[$f]" python scripts/r03/quick_headline.py 262144 4 | tail -1
  TAG="[$f]" python scripts/r03/quick_wholebody.py 65536 5
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
