"""The two standing fuzz deviations of round 5 as small fixtures for the driver-run suite.  Runs on the GPU box:
replays the case of the recorded fuzz run (same draws), takes the instances that are off the oracle's iteration count or furthest
from its z plus their neighbours (128 in all; instances are independent, so the subset reproduces the numbers), solves the subset
again on both solvers and writes inputs + the oracle's answers + the deviations found to gpurun_out/fuzz_fixtures/*.npz.
  python scripts/r06/make_fuzz_fixtures.py            (about two minutes: the replay draws every earlier case's model)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import fuzz_engines
OUT = os.path.join(ROOT, "gpurun_out", "fuzz_fixtures")
os.makedirs(OUT, exist_ok=True)
CASES = [dict(name="r05_j_fuzz_1500_case1212", ncase=1213, seed=4242, flat_bias=0.7, only=1212),
         dict(name="r05_j_fuzz_3000_case1126", ncase=1127, seed=9191, flat_bias=0.7, only=1126),
         dict(name="r06_e_fuzz_3000_case522", ncase=523, seed=6006, flat_bias=0.7, only=522),
         # (second session: the 10 000-case fuzz of the final sources, profiles/r06_k_fuzz_10000.txt)
         dict(name="r06_k_fuzz_10000_case6985", ncase=6986, seed=60606, flat_bias=0.7, only=6985),
         dict(name="r06_k_fuzz_10000_case7785", ncase=7786, seed=60606, flat_bias=0.7, only=7785),
         dict(name="r06_m_fuzz_6000_case4656", ncase=4657, seed=31337, flat_bias=0.7, only=4656),   # (profiles/r06_m_fuzz_6000.txt)
         # (last session: the 12 000-case fuzz of the final sources, profiles/r06_q_fuzz_12000.txt)
         dict(name="r06_q_fuzz_12000_case4160", ncase=4161, seed=424242, flat_bias=0.7, only=4160),
         dict(name="r06_q_fuzz_12000_case6317", ncase=6318, seed=424242, flat_bias=0.7, only=6317),
         dict(name="r06_q_fuzz_12000_case7166", ncase=7167, seed=424242, flat_bias=0.7, only=7166),
         dict(name="r06_q_fuzz_12000_case10176", ncase=10177, seed=424242, flat_bias=0.7, only=10176),
         # (after SolveInit's small-batch path changed: profiles/r06_s_fuzz_6000.txt; bit-identical on the library of before, scripts/r06/replay_case_two_libs.py)
         dict(name="r06_s_fuzz_6000_case5896", ncase=5897, seed=90210, flat_bias=0.7, only=5896)]
if len(sys.argv) > 1:   # (only the named cases)
    CASES = [c for c in CASES if c["name"] in sys.argv[1:]]
for c in CASES:
    box = {}
    fuzz_engines.fuzz(c["ncase"], c["seed"], verbose=True, only=c["only"], flat_bias=c["flat_bias"], capture=lambda d: box.update(d))
    m, wl, out, got = box["model"], box["wl"], box["out"], box["got"]
    B = wl["q"].shape[0]
    off = np.flatnonzero(~box["same"]); worst = np.argsort(-box["dz"])[:16]
    pick = list(dict.fromkeys([int(x) for x in off] + [int(x) for x in worst]))
    for x in range(B):
        if len(pick) >= 128: break
        if x not in pick: pick.append(x)
    pick = np.array(sorted(pick[:128]))
    def sub(a, per_instance_ndim):   # a batch-shaped array -> the subset; a shared one as it is
        a = np.asarray(a)
        return a[pick] if a.ndim == per_instance_ndim and a.shape[0] == B else a
    fx = dict(parents=m.parents, jtype=m.jtype, axis=m.axis, placement=m.placement,
              pitch=(m.pitch if m.pitch is not None else np.zeros(0)), has_composite=np.array(int(bool(m.composite))),
              q=sub(wl["q"], 2), H_ref=np.asarray(wl["H_ref"]), v_ref=np.asarray(wl["v_ref"]), c_ids=np.asarray(wl["c_ids"]),
              Ais=sub(wl["Ais"], 4), bis=sub(wl["bis"], 3), lb=sub(wl["lb"], 2), ub=sub(wl["ub"], 2),
              prm=np.array(json.dumps(box["prm"])), env=np.array(json.dumps(box["env"])), kw=np.array(json.dumps(box["kw"])), engine=np.array(box["engine"]),
              spare=np.array(box["spare"]), nc=np.array(box["nc"]), pick=pick,
              ref_iters=out["iters"][pick], ref_z=out["z"][pick], ref_converged=out["converged"][pick], ref_primal_infeasible=out["primal_infeasible"][pick],
              gpu_iters_full_batch=np.asarray(got["iter"])[pick], gpu_dz_full_batch=box["dz"][pick])
    if box["refs"] is not None:
        fx["refs_H"] = box["refs"][0]; fx["refs_v"] = box["refs"][1]
    assert not m.composite, "fixture format: no composite joints expected in these cases"
    fx["B_full"] = np.array(B)
    np.savez_compressed(os.path.join(OUT, c["name"] + ".npz"), **fx)
    print(c["name"], "instances", len(pick), "off-count", len(off), "max |dz| same-iteration", float(box["dz"][box["same"]].max()), "max |dz| overall", float(box["dz"].max()),
          "bytes", os.path.getsize(os.path.join(OUT, c["name"] + ".npz")), flush=True)
