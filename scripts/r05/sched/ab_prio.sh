#!/bin/bash
# headline batch, fresh batches: the time-sliced launch's schedulers -- 0 round robin (round 4), 1 priority classes, 2 round robin with marks
# (an instance predicted to be long is not parked) -- and a few slice / mark settings
mkdir -p gpurun_out/prio
for cfg in "0 - -" "2 - -" "2 192 96" "2 224 128" "2 160 96" "1 - -" "0 - -" "2 - -"; do
  set -- $cfg
  export LOIKB_FLAT_PRIO=$1
  unset LOIKB_FLAT_SLICE LOIKB_FLAT_SLICE2
  [ "$2" != "-" ] && export LOIKB_FLAT_SLICE=$2 LOIKB_FLAT_SLICE2=$3
  echo "== prio $1 slice $2 marks $3"
  timeout 300 python bench.py --steps 12 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
r = d['roofline']
print(json.dumps({'ms_per_step': d['ms_per_step'], 'value': d['value'], 'tail_ms': r.get('tail', {}).get('ms'), 'kernel_ms': r.get('avg_launch_ms'), 'one_handle': d.get('ms_per_step_one_handle')}))
"
done
