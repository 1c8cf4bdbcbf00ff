#!/bin/bash
# diagnostics that need a -DLOIKB_TAIL_PROF build of the library (rebuilt back afterwards).
# Since round 6 the profile build keeps the library's TWO translation units -- the flat kernels with the code generation they ship with (the
# iterative scheduler cannot be used on a single unit) --: PROF_ONE_UNIT=1 gives the one-unit build of rounds 3-6 (default schedule everywhere).
cd ${GRAFT_REPO_ROOT:-.}
if [ -n "$PROF_ONE_UNIT" ]; then FL="('-DLOIKB_TAIL_PROF',)"; else FL="('-DLOIKB_TAIL_PROF', '-DLOIKB_TAIL_PROF_TWO_UNITS')"; fi
python -c "from loik_amd import _build; _build.build(force=True, extra_flags=$FL)" > /dev/null 2>&1
python ${SCRIPT:-scripts/r03/flat_phase_profile.py} ${ARGS:-64 65536}
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
