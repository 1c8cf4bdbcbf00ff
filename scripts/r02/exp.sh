cd ${GRAFT_REPO_ROOT:-.}
SLICES=0 SCRIPT=scripts/tail_phase_profile.py ARGS="65536" bash scripts/r02/prof_build.sh 2>&1 | tail -14
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-variants 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
