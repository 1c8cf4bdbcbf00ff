#!/bin/bash
# diagnostics that need a -DLOIKB_TAIL_PROF build of the library (rebuilt back afterwards)
cd ${GRAFT_REPO_ROOT:-.}
python -c "from loik_amd import _build; _build.build(force=True, extra_flags=('-DLOIKB_TAIL_PROF',))" > /dev/null 2>&1
for s in ${SLICES:-0 64}; do echo "== LOIKB_LEAN_SLICE=$s"; LOIKB_LEAN_SLICE=$s python ${SCRIPT:-scripts/tail_phase_profile.py} ${ARGS:-65536}; done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
