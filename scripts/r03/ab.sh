#!/bin/bash
# A/B of build variants inside ONE gpurun call (boxes differ by a few %): ab.sh "<flags A>" "<flags B>" ...
cd ${GRAFT_REPO_ROOT:-.}
for f in "$@"; do
  python -c "from loik_amd import _build; _build.build(force=True, extra_flags=tuple('$f'.split()))" > /dev/null 2>&1 || echo "build failed: $f"
  for a in ${SIZES:-65536 4096}; do TAG="[$f]" python ${SCRIPT:-scripts/r03/quick_headline.py} $a 6; done
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
