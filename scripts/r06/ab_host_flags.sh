#!/bin/bash
# code-generation switches of the OTHER translation unit (k_fslots, k_lean, k_tail, k_solve, the record movers): ab_host_flags.sh none "<flags>" ...
cd ${GRAFT_REPO_ROOT:-.}
for f in "$@"; do
  ff="$f"; [ "$f" = none ] && ff=""
  python -c "from loik_amd import _build; _build.build(force=True, host_flags='$ff'.split())" > /dev/null 2>&1 || echo "build failed: $f"
  TAG="[$f arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[$f arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_wholebody.py 65536 6 | tail -1
  TAG="[$f]" python scripts/bench_configs.py 2>/dev/null | tail -6 | cut -c1-220
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
