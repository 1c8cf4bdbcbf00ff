#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for v in "" "-DLOIKB_POLL_MASK=63u" "-DLOIKB_PLAIN_RECORDS" "-DLOIKB_POLL_MASK=63u -DLOIKB_PLAIN_RECORDS"; do
  python -c "from loik_amd import _build; _build.build(force=True, extra_flags=tuple('$v'.split()))" > /dev/null 2>&1
  echo "== variant [$v]"
  SLICES=0,64 python scripts/r02/slice_sweep.py c3 | cut -c1-80
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
