#!/bin/bash
# the bench line with the committed PMC summary of the same sources (pmc_stale false), then the long fuzz on the final sources
TAG=${1:-r06_e}; O=gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python $R/bench.py > $R/$O/bench_line.json 2> $R/$O/bench.err
cd $R
timeout 2400 python scripts/fuzz_engines.py 3000 6006 0.7 2>&1 | grep -v " ok *$" > $O/fuzz_3000.txt
tail -1 $O/fuzz_3000.txt | cut -c1-600
