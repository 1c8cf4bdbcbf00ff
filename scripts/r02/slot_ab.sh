#!/bin/bash
# in-box A/B: decade slot fetched in one round trip (default) vs in two parts (-DLOIKB_SLOT_TWO_PARTS)
cd ${GRAFT_REPO_ROOT:-.}
bench3() { for i in 1 2 3; do python bench.py --no-cpu-baseline --no-variants --steps 10 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"; done; }
for round in 1 2; do
  python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1; echo "== one round trip"; bench3
  python -c "from loik_amd import _build; _build.build(force=True, extra_flags=('-DLOIKB_SLOT_TWO_PARTS',))" > /dev/null 2>&1; echo "== two parts"; bench3
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
