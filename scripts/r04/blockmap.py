#!/usr/bin/env python3
"""Basic-block map of one kernel of the device assembly: start line, label, loop depth, instruction counts, branch targets.
   python scripts/r04/blockmap.py cur.s k_flat2ILi10ELi2ELb0ELi0E [first_line last_line]"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 10**9)
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN5loikb') and key in l and '@' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
L = lines[start:end]
blocks = []; cur = None
for i, l in enumerate(L):
    m = re.match(r'^(\.LBB\d+_\d+):', l); m2 = re.match(r'^; %bb\.(\d+):', l)
    if m or m2:
        d = re.search(r'Depth=(\d+)', l)
        cur = {'name': m.group(1) if m else 'bb.' + m2.group(1), 'start': i, 'ins': [], 'depth': d.group(1) if d else '0'}
        blocks.append(cur)
    elif cur is not None and l.startswith('\t') and not l.strip().startswith(('.', ';')):
        cur['ins'].append(l.strip())
for b in blocks:
    if lo <= b['start'] <= hi:
        n = lambda *p: sum(1 for x in b['ins'] if x.startswith(p))
        br = [x.replace('s_cbranch_', '').replace('\t', ' ') for x in b['ins'] if x.startswith(('s_cbranch', 's_branch'))]
        print(b['start'], b['name'], 'd' + b['depth'], 'n', len(b['ins']), 'valu', n('v_'), 'f64', n('v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_max_f64', 'v_min_f64'),
              'lds', n('ds_'), 'scr', n('scratch'), 'smem', n('s_load'), 'vmem', n('global_', 'buffer_'), 'rl', n('v_readlane', 'v_writelane'), ' '.join(br))
