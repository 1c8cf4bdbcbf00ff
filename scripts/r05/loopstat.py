#!/usr/bin/env python3
"""Static view of a kernel's iteration loop from the device assembly, by the compiler's own loop annotations.

  hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o k.s ;  python scripts/r05/loopstat.py k.s k_flat2ILi10ELi3ELb0 [-v]

The iteration loop is the innermost loop (`in Loop: Header=BBx_y Depth=d` on the block labels) that holds the DPP fold of the stopping test
(row_ror:8).  Counts are of the static body; -v lists the scratch accesses and scalar loads in it with their line numbers.
(scripts/r04/inloop.py looks for backward branches, which the structurizer's dispatch blocks defeat.)"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
verbose = '-v' in sys.argv
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN5loikb') and key in l.split(':')[0] and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
# blocks: (first line, header, depth)
blocks = []
cur = (0, None, 0)
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):\s*;.*', l) or re.match(r'^; %bb\.\d+:\s*;.*', l)
    if re.match(r'^(\.LBB\d+_\d+:|; %bb\.\d+:)', l):
        h = re.search(r'Header=(BB\d+_\d+) Depth=(\d+)', l)
        own = re.match(r'^\.L(BB\d+_\d+):', l)
        nxt = body[i + 1] if i + 1 < len(body) else ''
        h2 = re.search(r'Loop Header: Depth=(\d+)', l)
        if h: cur = (i, h.group(1), int(h.group(2)))
        elif h2 and own: cur = (i, own.group(1), int(h2.group(1)))
        else: cur = (i, None, 0)
        blocks.append(cur)
# per line: (header, depth)
info = [None] * len(body)
bi = 0
for i in range(len(body)):
    while bi + 1 < len(blocks) and blocks[bi + 1][0] <= i: bi += 1
    info[i] = blocks[bi] if blocks and blocks[bi][0] <= i else (0, None, 0)
fold = next(i for i, l in enumerate(body) if 'row_ror:8' in l)
hdr, depth = info[fold][1], info[fold][2]
# the loop's lines: every block with this header, plus deeper loops nested between its first and last block
mine = [i for i in range(len(body)) if info[i][1] == hdr]
lo, hi = min(mine), max(mine)
sel = [i for i in range(lo, hi + 1) if info[i][1] == hdr or info[i][2] > depth]
c = collections.Counter()
notes = []
for i in sel:
    l = body[i]
    if not l.startswith('\t') or l.strip().startswith(('.', ';')): continue
    x = l.split()[0]
    c['instructions'] += 1
    if x.startswith('v_'): c['VALU'] += 1
    if x.startswith(('v_fma_f64', 'v_fmac_f64', 'v_mul_f64', 'v_add_f64', 'v_max_f64', 'v_min_f64')): c['f64'] += 1
    if x.startswith('ds_'): c['LDS'] += 1
    if x.startswith('s_') and not x.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch')): c['SALU'] += 1
    if x.startswith('s_waitcnt'): c['waitcnt'] += 1
    if x.startswith('s_nop'): c['nop'] += 1
    if x.startswith('scratch_'): c[x] += 1; notes.append((i, l.strip()))
    if x.startswith(('global_', 'buffer_', 'flat_')): c['VMEM'] += 1
    if x.startswith(('s_load', 's_buffer_load')): c['SMEM'] += 1; notes.append((i, l.strip()))
    if x.startswith(('s_cbranch', 's_branch')): c['branch'] += 1
    if x.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')): c['lane<->scalar'] += 1
    if x.startswith(('v_mov_b32', 'v_mov_b64')) and 'row_' not in l and 'quad_perm' not in l: c['v_mov'] += 1
    if 'row_' in l or 'quad_perm' in l: c['dpp'] += 1
    if 'permlane' in x: c['permlane'] += 1
# the inline asm v_max/v_min/v_bfe are between ASMSTART/ASMEND and start with a tab too: counted above
print(f"{key}: loop header {hdr} depth {depth}, lines {lo}..{hi}; " + ', '.join(f'{k} {v}' for k, v in sorted(c.items())))
if verbose:
    for i, l in notes: print(f"   {i}: {l}   [{info[i][1]} d{info[i][2]}]")
