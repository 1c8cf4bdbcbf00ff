#!/bin/bash
# the iteration loop's rate by build: N copies of one 999-iteration instance (scripts/r06/pairing_probe.py), N = 1 / 1024 / 2048
#   ab_loop_rate.sh none "<flags A>" ...
cd ${GRAFT_REPO_ROOT:-.}
export PROBE_N=1,1024,2048 LOIKB_FLAT_ORDER=0
for f in "$@"; do
  ff="$f"; [ "$f" = none ] && ff=""
  python -c "from loik_amd import _build; _build.build(force=True, flat_flags=_build.FLAT_FLAGS + '$ff'.split())" > /dev/null 2>&1 || echo "build failed: $f"
  TAG="[$f | plain]" LOIKB_FLAT_SLICE=0 python scripts/r06/pairing_probe.py
  TAG="[$f | sliced, lazy table]" LOIKB_FLAT_SLICE=2000 python scripts/r06/pairing_probe.py
  TAG="[$f | sliced, full table]" LOIKB_FLAT_SLICE=2000 LOIKB_FLAT_BUILD=0 python scripts/r06/pairing_probe.py
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
