#!/bin/bash
# in-box A/B: default register-pressure trackers vs -mllvm -amdgpu-use-amdgpu-trackers=1 (k_lean scratch 76 -> 52 bytes)
cd ${GRAFT_REPO_ROOT:-.}
bench3() { for i in 1 2 3; do python bench.py --no-cpu-baseline --no-variants --steps 10 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3), round(d['roofline']['other_kernel']['avg_launch_ms'],3))"; done; }
for round in 1 2 3; do
  python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1; echo "== default"; bench3
  python -c "from loik_amd import _build; _build.build(force=True, extra_flags=('-mllvm', '-amdgpu-use-amdgpu-trackers=1'))" > /dev/null 2>&1; echo "== trackers"; bench3
done
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
