#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for f in "$@"; do
  python -c "from loik_amd import _build; _build.build(force=True, extra_flags=tuple('$f'.split()))" > /dev/null 2>&1 || echo "build failed: $f"
  export LOIKB_FLAT_ORDER=0
  LOIKB_FLAT_SLICE=100000 TAG="[$f wb never]" python scripts/r03/quick_wholebody.py 65536 5
  TAG="[$f wb default]" python scripts/r03/quick_wholebody.py 65536 5
  LOIKB_FLAT_SLICE=100000 TAG="[$f never]" python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[$f default]" python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[$f]" python scripts/r03/quick_headline.py 8192 6 | tail -1
  LOIKB_FLAT_SLICE=288 TAG="[$f sliced]" python scripts/r03/quick_headline.py 8192 6 | tail -1
  unset LOIKB_FLAT_ORDER
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
