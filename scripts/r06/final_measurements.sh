#!/bin/bash
# the end-of-round set on the final sources (round 6): C4 line, mu rules, batch scaling (repeat + fresh), small batches / single calls, the other
# BASELINE configurations, multi-task.   usage: final_measurements.sh <tag>
TAG=${1:-r06_x}
O=gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python $R/bench.py --config c4 > $R/$O/bench_line_c4.json 2>/dev/null
cd $R
timeout 400 python scripts/r05/mu_rules.py 65536 > $O/mu_rules.jsonl 2>/dev/null
timeout 600 python scripts/bench_batch_scaling.py > $O/batch_scaling.jsonl 2>/dev/null
for B in 8192 16384 32768 65536; do LOIKB_FLAT_ORDER=0 TAG="fresh (arrival order)" timeout 120 python scripts/r06/quick_probe.py $B 6; done > $O/batch_scaling_fresh.txt 2>/dev/null
timeout 300 python scripts/r06/small_latency.py 1 8 64 256 1024 4096 > $O/small_batches.jsonl 2>/dev/null
timeout 300 python scripts/bench_configs.py > $O/configs.jsonl 2>/dev/null
timeout 300 python scripts/bench_multi_task.py > $O/multi_task.jsonl 2>/dev/null
wc -l $O/*.jsonl $O/*.txt
