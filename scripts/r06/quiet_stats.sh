#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -c "from loik_amd import _build; _build.build(force=True, extra_flags=('-DLOIKB_TAIL_PROF', '-DLOIKB_DBG_QUIET'))" > /dev/null 2>&1
python scripts/r06/quiet_stats.py
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
