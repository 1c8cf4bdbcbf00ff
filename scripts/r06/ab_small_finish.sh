#!/bin/bash
# A/B of the short sequence's end (run_tail, small_flat): k_list_unfinished + copy + run_main_loop's event / synchronisation (LOIKB_SMALL_FINISH=0, as before)
# against k_small_finish (counters straight into the pinned host copy, one synchronisation), with and without the event between k_fslots and the engine
TAG=${1:-r06_o}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in "0 1" "1 1" "1 0"; do
    set -- $v
    TAG="finish=$1,slot_event=$2" LOIKB_SMALL_FINISH=$1 LOIKB_SMALL_SLOT_EVENT=$2 timeout 300 python scripts/r06/small_latency.py 1 8 64 1024 >> $O/small_finish_ab.jsonl 2>> $O/err.txt
  done
done
cat $O/small_finish_ab.jsonl | cut -c1-260
timeout 900 python -m pytest tests/test_engines.py tests/test_solver_info.py -x -q -m gpu 2>&1 | tail -3
