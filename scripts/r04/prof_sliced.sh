#!/bin/bash
# phase timelines (all wavefronts) of the unsliced and of the SLICED build with a slice that never ends: what the SLICED build itself costs
cd ${GRAFT_REPO_ROOT:-.}
python -c "from loik_amd import _build; _build.build(force=True, extra_flags=('-DLOIKB_TAIL_PROF',))" > /dev/null 2>&1
for q in 0 100000 288; do echo "== LOIKB_FLAT_SLICE=$q"; LOIKB_FLAT_SLICE=$q LOIKB_FLAT_ORDER=0 python scripts/r03/flat_phase_profile.py 65536 | grep -A22 "ALL 2048"; done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
