"""Bank conflicts of k_flat1's gathers for a tree, from the host-built schedule (loikb_flat_schedule) and MI355X's ds_read_b64 rule
(groups of 32 lanes; bank pair of an 8-byte word = word index mod 32; distinct addresses on one bank pair serialise, equal ones broadcast).
Prints LDS-array cycles per gather instruction (2 = conflict-free) -- where the kernel's conflict cycles come from.  CPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, loik_amd
from loik_amd import capi
name = sys.argv[1] if len(sys.argv) > 1 else "talos44"
m = loik_amd.builtin_model(name); fs = capi.flat_schedule(m.parents)
G, nb = fs["G"], len(m.parents) - 1
W = 64
def cycles(idx):            # idx: word index per lane (64 lanes; for G = 32 the caller maps both halves)
    tot = 0
    for g in (range(0, 32), range(32, 64)):
        banks = {}
        for l in g:
            banks.setdefault(int(idx[l]) % 32, set()).add(int(idx[l]))
        tot += max(len(v) for v in banks.values())
    return tot
depth, size, anc, red, part = fs["depth"], fs["size"], fs["anc"], fs["red"], fs["part"]
lanes = np.arange(W)
rows = []
if G == 64:
    none_n = W            # nbuf / pbuf zero pad index
    for t in range(8):
        idx = np.where(red[:, t] >= 0, red[:, t], 10 * 64 + 64 + 64)   # (none: a pad far away, one address)
        rows.append(("W tau share a[%d]" % t, cycles(idx)))
    for q in range(8):
        idx = np.where(part[:, q] >= 0, part[:, q], W)
        rows.append(("partials pp[%d]" % q, cycles(idx)))
    for k in range(fs["nanc"]):
        idx = np.where(anc[:, k] >= 0, anc[:, k], W)
        rows.append(("nu gather nb[%d]" % k, cycles(idx)))
    def anc_at(d):
        out = np.full(W, W)
        for l in range(nb):
            k = depth[l] - d - 1
            if k >= 0 and anc[l, k] >= 0: out[l] = anc[l, k]
        return out
    for d in (1, 2, 3, 4, 8):
        for c in range(6):
            rows.append(("path d=%d c=%d" % (d, c), cycles(c * 66 + anc_at(d))))
    src = lanes + np.where(size > 0, size - 1, 0)
    for c in range(6):
        rows.append(("prefix src c=%d" % c, cycles(c * 66 + src)))
tot = sum(c for _, c in rows)
for n_, c in rows:
    if c > 2: print("%-22s %d cycles" % (n_, c))
print("%s: %d gather instructions, %d LDS-array cycles (%d conflict-free): %.0f %% conflict cycles" % (name, len(rows), tot, 2 * len(rows), 100.0 * (tot - 2 * len(rows)) / tot))
