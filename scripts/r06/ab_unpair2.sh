#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for f in "-DLOIKB_UNPAIR_R1=0" "-DLOIKB_UNPAIR_R2=0" "-DLOIKB_UNPAIR_R1=0 -DLOIKB_UNPAIR_R2=0"; do
  python -c "from loik_amd import _build; _build.build(force=True, flat_flags=_build.FLAT_FLAGS + '$f'.split())" > /dev/null 2>&1 || echo "build failed: $f"
  for u in 0 1; do TAG="[$f | unpair $u]" LOIKB_FLAT_UNPAIR=$u LOIKB_FLAT_ORDER=0 timeout 120 python scripts/r03/quick_headline.py 65536 6 | tail -1; done
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
