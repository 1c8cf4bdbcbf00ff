#!/bin/bash
# phase timeline (clock64 stamps, -DLOIKB_TAIL_PROF build) of a lone long instance: the plain build against the time-sliced one
cd ${GRAFT_REPO_ROOT:-.}
python -c "from loik_amd import _build; _build.build(force=True, extra_flags=('-DLOIKB_TAIL_PROF',))" > /dev/null 2>&1
echo "=== plain";  LOIKB_FLAT_ORDER=0 LOIKB_FLAT_SLICE=0 python scripts/r03/flat_phase_profile.py 64 | head -22
echo "=== sliced (never ends)"; LOIKB_FLAT_ORDER=0 LOIKB_FLAT_SLICE=2000 LOIKB_FLAT_BUILD=0 python scripts/r03/flat_phase_profile.py 64 | head -22
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
