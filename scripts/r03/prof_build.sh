#!/bin/bash
# diagnostics that need a -DLOIKB_TAIL_PROF build of the library (rebuilt back afterwards)
cd ${GRAFT_REPO_ROOT:-.}
python -c "from loik_amd import _build; _build.build(force=True, extra_flags=('-DLOIKB_TAIL_PROF',))" > /dev/null 2>&1
python ${SCRIPT:-scripts/r03/flat_phase_profile.py} ${ARGS:-64 65536}
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
