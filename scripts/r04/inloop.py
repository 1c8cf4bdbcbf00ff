#!/usr/bin/env python3
"""Static count of the instructions in a kernel's main loop, from the device assembly (no GPU needed).

  hipcc --offload-arch=gfx950 -O3 -ffp-contract=on -std=c++17 -I include -x hip loik_amd/csrc/loik_host.hip -S --cuda-device-only -o cur.s
  python scripts/r04/inloop.py cur.s k_flat1ILi16ELb0ELi0E

The main loop is taken as the innermost-or-not backward branch span holding the most v_fma_f64; the counts are of the static body
(conditional blocks inside it count in full)."""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN5loikb') and key in l and l.rstrip().split(':')[0].endswith(l.split(':')[0]) and ':' in l and '@' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
spans = []
for i, l in enumerate(body):
    m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|\s+s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = labels.get(m.group(1) or m.group(2))
        if t is not None and t < i: spans.append((t, i))
def instrs(a, b): return [l.split()[0] for l in body[a:b + 1] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
# the iteration loop: the shortest backward-branch span that still holds the DPP folds of the stopping test (row_ror:8)
def has_fold(a, b): return any('row_ror:8' in l for l in body[a:b + 1])
def n64(a, b): return sum(1 for x in instrs(a, b) if x.startswith(('v_fma_f64', 'v_mul_f64', 'v_add_f64')))
best = min((s for s in spans if has_fold(*s) and n64(*s) >= 100), key=lambda s: s[1] - s[0])
ins = instrs(*best)
c = collections.Counter()
for x in ins:
    if x.startswith('v_'): c['VALU'] += 1
    if x.startswith(('v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_max_f64', 'v_min_f64')): c['f64'] += 1
    if x.startswith('ds_'): c['LDS'] += 1
    if x.startswith('s_') and not x.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch')): c['SALU'] += 1
    if x.startswith('s_waitcnt'): c['waitcnt'] += 1
    if x.startswith('s_nop'): c['nop'] += 1
    if x.startswith('scratch_'): c[x] += 1
    if x.startswith(('global_', 'buffer_', 'flat_')): c['VMEM'] += 1
    if x.startswith(('s_load', 's_buffer_load')): c['SMEM'] += 1
    if x.startswith(('s_cbranch', 's_branch')): c['branch'] += 1
    if x.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')): c['lane<->scalar'] += 1
    if x.startswith('v_mov_b32') or x.startswith('v_mov_b64'): c['v_mov'] += 1
    if 'dpp' in x: c['dpp(name)'] += 1
dpp = sum(1 for l in body[best[0]:best[1] + 1] if 'row_' in l or 'quad_perm' in l)
print(f"{key}: loop lines {best[0]}..{best[1]} of {len(body)}; instructions {len(ins)}; DPP {dpp}")
print('  ' + ', '.join(f'{k} {v}' for k, v in sorted(c.items())))
print(f"  other loops (span, fma64): " + ', '.join(f'{a}-{b}:{sum(1 for x in instrs(a,b) if x.startswith("v_fma_f64"))}' for a, b in sorted(set(spans), key=lambda s: s[0] - s[1])[:6]))
