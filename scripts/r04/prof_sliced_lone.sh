#!/bin/bash
# phase timelines of LONE instances (64-instance batch, all wavefronts summed): plain build against the SLICED build with a slice that never ends
cd ${GRAFT_REPO_ROOT:-.}
python -c "from loik_amd import _build; _build.build(force=True, extra_flags=('-DLOIKB_TAIL_PROF',))" > /dev/null 2>&1
for q in 0 100000; do echo "== LOIKB_FLAT_SLICE=$q"; LOIKB_FLAT_SLICE=$q LOIKB_FLAT_ORDER=0 python scripts/r03/flat_phase_profile.py 64 | grep -A22 "ALL 64"; done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
