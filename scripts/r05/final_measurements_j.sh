#!/bin/bash
# the end-of-round set on the final sources (round 5, last session): bench line (pmc_latest.json = r05_j_pmc.json -> pmc_stale false), C4 line, mu rules,
# batch scaling, small batches, the other BASELINE configurations, the whole body under the mu rules
O=gpurun_out/r05_j2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python $R/bench.py > $R/$O/bench_line.json 2> $R/$O/bench.err
timeout 600 python $R/bench.py --config c4 > $R/$O/bench_line_c4.json 2>/dev/null
cd $R
timeout 400 python scripts/r05/mu_rules.py 65536 > $O/mu_rules.jsonl 2>/dev/null
timeout 600 python scripts/bench_batch_scaling.py > $O/batch_scaling.jsonl 2>/dev/null
timeout 300 python scripts/bench_small_batches.py > $O/small_batches.jsonl 2>/dev/null
timeout 300 python scripts/bench_configs.py > $O/configs.jsonl 2>/dev/null
timeout 300 python scripts/bench_multi_task.py > $O/multi_task.jsonl 2>/dev/null
wc -l $O/*.jsonl
