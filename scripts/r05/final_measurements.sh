mkdir -p gpurun_out/r05_h2; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python $R/bench.py > $R/gpurun_out/r05_h2/bench_line.json 2> $R/gpurun_out/r05_h2/bench.err
timeout 600 python $R/bench.py --config c4 > $R/gpurun_out/r05_h2/bench_line_c4.json 2>/dev/null
cd $R
timeout 400 python scripts/r05/mu_rules.py 65536 > gpurun_out/r05_h2/mu_rules.jsonl 2>/dev/null
timeout 600 python scripts/bench_batch_scaling.py > gpurun_out/r05_h2/batch_scaling.jsonl 2>/dev/null
timeout 300 python scripts/bench_small_batches.py > gpurun_out/r05_h2/small_batches.jsonl 2>/dev/null
wc -l gpurun_out/r05_h2/*.jsonl
