#!/bin/bash
# phase timelines (all wavefronts) of the time-sliced launch: round robin against the priority classes
cd ${GRAFT_REPO_ROOT:-.}
python -c "from loik_amd import _build; _build.build(force=True, extra_flags=('-DLOIKB_TAIL_PROF',))" > /dev/null 2>&1
for p in 0 1; do echo "== LOIKB_FLAT_PRIO=$p"; LOIKB_FLAT_PRIO=$p LOIKB_FLAT_ORDER=0 python scripts/r03/flat_phase_profile.py 65536 | grep -A24 "ALL 2048"; done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
