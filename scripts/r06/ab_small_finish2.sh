#!/bin/bash
# the short sequence after the fused reset + queue launch (reset_home, with_queue): old end / new end / new end + slot event, then the whole GPU suite
TAG=${1:-r06_o}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in "0 1" "1 1" "1 0"; do
    set -- $v
    TAG="fused reset; finish=$1,slot_event=$2" LOIKB_SMALL_FINISH=$1 LOIKB_SMALL_SLOT_EVENT=$2 timeout 300 python scripts/r06/small_latency.py 1 8 64 1024 >> $O/small_finish_ab2.jsonl 2>> $O/err.txt
  done
done
cut -c1-200 $O/small_finish_ab2.jsonl | grep '"batch": 1,'
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
