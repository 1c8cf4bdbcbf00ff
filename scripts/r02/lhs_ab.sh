cd ${GRAFT_REPO_ROOT:-.}
echo "== LHS 22 (built)"; python scripts/bench_multi_task.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['config'], round(d['ms_per_solve'], 2))"
sed -i 's/^constexpr int LHS = 22;/constexpr int LHS = 21;/' loik_amd/csrc/loik_lean.hpp
python -c "from loik_amd import _build; _build.build(force=True)" > /dev/null 2>&1
echo "== LHS 21"; python scripts/bench_multi_task.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['config'], round(d['ms_per_solve'], 2))"
