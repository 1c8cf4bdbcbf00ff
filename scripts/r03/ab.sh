#!/bin/bash
# A/B of build variants inside ONE gpurun call (boxes differ by a few %): ab.sh "<flags A>" "<flags B>" ...
cd ${GRAFT_REPO_ROOT:-.}
for f in "$@"; do
  python -c "from loik_amd import _build; _build.build(force=True, extra_flags=tuple('$f'.split()))" > /dev/null 2>&1 || echo "build failed: $f"
  for r in 1 2; do TAG="[$f]" python ${SCRIPT:-scripts/r03/quick_headline.py} ${ARGS:-65536 6}; done
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
