#!/bin/bash
# A/B of code-generation switches of the flat kernels' translation unit only (added to loik_amd/_build.py::FLAT_FLAGS), inside ONE gpurun call:
#   ab_flat_flags.sh none "<flags A>" "<flags B>" ...      -- lone iteration, headline arrival / ordered, 4 x batch, whole body arrival / ordered
cd ${GRAFT_REPO_ROOT:-.}
for f in "$@"; do
  ff="$f"; [ "$f" = none ] && ff=""
  python -c "from loik_amd import _build; _build.build(force=True, flat_flags=_build.FLAT_FLAGS + '$ff'.split())" > /dev/null 2>&1 || echo "build failed: $f"
  TAG="[$f]" python scripts/r04/lone.py
  TAG="[$f ordered]" python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[$f arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[$f]" python scripts/r03/quick_headline.py 262144 4 | tail -1
  TAG="[$f ordered]" python scripts/r03/quick_wholebody.py 65536 6 | tail -1
  TAG="[$f arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_wholebody.py 65536 6 | tail -1
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
