#!/bin/bash
# like ab_loop_rate.sh, the time-sliced build only, N = 1024 / 2048
cd ${GRAFT_REPO_ROOT:-.}
export PROBE_N=${PROBE_N:-1024,2048} LOIKB_FLAT_ORDER=0
for f in "$@"; do
  ff="$f"; [ "$f" = none ] && ff=""
  python -c "from loik_amd import _build; _build.build(force=True, flat_flags=_build.FLAT_FLAGS + '$ff'.split())" > /dev/null 2>&1 || echo "build failed: $f"
  TAG="[$f | plain]" LOIKB_FLAT_SLICE=0 python scripts/r06/pairing_probe.py | grep "N="
  TAG="[$f | sliced, full table]" LOIKB_FLAT_SLICE=2000 LOIKB_FLAT_BUILD=0 python scripts/r06/pairing_probe.py | grep "N="
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
