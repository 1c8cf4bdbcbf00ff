#!/bin/bash
# headline with the library built with extra compiler flags (on the GPU box); the default build is restored at the end
cd ${GRAFT_REPO_ROOT:-.}
run() {
  python - "$@" <<'PY'
import sys
from loik_amd import _build
_build.build(force=True, extra_flags=tuple(sys.argv[1:]))
PY
  for i in 1 2 3; do python bench.py --no-cpu-baseline --no-variants --steps 10 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"; done
}
echo "== default"; run
echo "== trackers"; run -mllvm -amdgpu-use-amdgpu-trackers=1
echo "== no-unroll"; run -fno-unroll-loops
echo "== trackers + no-unroll"; run -mllvm -amdgpu-use-amdgpu-trackers=1 -fno-unroll-loops
python -c "from loik_amd import _build; _build.build(force=True)"
