#!/bin/bash
# what one Solve() of the headline batch puts on the stream: kernel trace of three fresh-handle solves (start offsets and durations, us)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/solve_tl; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
LOIKB_FLAT_ORDER=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o tl -- python $R/scripts/r03/quick_headline.py 65536 3 > $O/log.txt 2>&1
python - "$O" <<'PY'
import csv, glob, sys
fn = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r['Start_Timestamp']))
# the last solve: from the last k_fslots to the end
idx = [i for i, r in enumerate(rows) if 'k_fslots' in r['Kernel_Name']][-1]
t0 = int(rows[max(0, idx - 6)]['Start_Timestamp'])
prev_end = None
for r in rows[max(0, idx - 6):]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print('%9.1f us  +%7.1f us  (gap %6.1f)  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, r['Kernel_Name'][:60]))
    prev_end = e
PY
