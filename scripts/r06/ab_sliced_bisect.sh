#!/bin/bash
# what makes the time-sliced build's loop slower at two wavefronts per SIMD?  experiment switches, the loop-rate probe at N = 1 / 2048
cd ${GRAFT_REPO_ROOT:-.}
export PROBE_N=1,2048 LOIKB_FLAT_ORDER=0
for f in none "-DLOIKB_X_NOHELD=1" "-DLOIKB_X_NOTOP=1" "-DLOIKB_PLAIN_RECORDS" "-DLOIKB_X_NOHELD=1 -DLOIKB_X_NOTOP=1 -DLOIKB_PLAIN_RECORDS"; do
  ff="$f"; [ "$f" = none ] && ff=""
  python -c "from loik_amd import _build; _build.build(force=True, extra_flags='$ff'.split())" > /dev/null 2>&1 || echo "build failed: $f"
  TAG="[$f | plain]" LOIKB_FLAT_SLICE=0 python scripts/r06/pairing_probe.py | grep "N= 2048"
  TAG="[$f | sliced]" LOIKB_FLAT_SLICE=2000 LOIKB_FLAT_BUILD=0 python scripts/r06/pairing_probe.py | grep "N="
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
