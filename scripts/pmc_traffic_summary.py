"""sum FETCH_SIZE / WRITE_SIZE (KiB) over the solve kernels of the profiled bench run -> JSON (per Solve() step)"""
import csv, json, sys, collections
fetch_csv, write_csv, nsolves = sys.argv[1], sys.argv[2], int(sys.argv[3])  # nsolves = warmup + steps of the profiled run
def load(path, counter):
    tot = collections.defaultdict(float); n = collections.Counter()
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"]
        key = ("k_solve" if "k_solve" in name else "k_tail" if "k_tail" in name else "k_move" if "k_move" in name else
               "k_lean" if "k_lean" in name else "k_hslots" if "k_hslots" in name else None)
        if key:
            tot[key] += float(row["Counter_Value"]); n[key] += 1
    return tot, n
f, nf = load(fetch_csv, "FETCH_SIZE")
w, nw = load(write_csv, "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950, 16 B/lane streaming reads); WRITE_SIZE uncorrected",
       "solves_profiled": nsolves, "kernels": {}}
for k in sorted(set(f) | set(w)):
    out["kernels"][k] = {"dispatches_per_step": nf[k] / nsolves,
                         "fetch_bytes_per_step": 2.0 * f[k] * 1024 / nsolves,
                         "write_bytes_per_step": w[k] * 1024 / nsolves}
tot = sum(v["fetch_bytes_per_step"] + v["write_bytes_per_step"] for v in out["kernels"].values())
out["hbm_bytes_per_step"] = tot
ks = out["kernels"].get("k_solve")
if ks:
    out["k_solve_hbm_bytes_per_step"] = ks["fetch_bytes_per_step"] + ks["write_bytes_per_step"]
print(json.dumps(out, indent=1))
