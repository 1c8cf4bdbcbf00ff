"""How far the flat engines' OSQP-rule solves sit from the oracle on test_osqp_mu_rule_matches_oracle's batch (and four more seeds): the share of
instances at the oracle's iteration count and the largest |dz| / |dnu| among them -- the margin behind that test's tolerances."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import loik_amd
from helpers import FIXTURE, feasible_batch
from oracle import ref
talos = loik_amd.builtin_model("talos32")
link = talos.getJointId("arm_left_7_joint")
prm = dict(FIXTURE, max_iter=500, tol_abs=1e-6, tol_rel=0.0, mu_update_strat=1)
for seed in (91, 92, 93, 94, 95):
    for B in (700, 4000):
        wl = feasible_batch(talos, B, link, seed, nu_scale=0.5)
        args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        s = loik_amd.BatchedLoik(talos, B, **prm)
        s.Solve(*args)
        out = ref.solve_batch(talos, *args, nthreads=8, want_nu=True, **prm)
        it, z, nu = np.asarray(s.get("iter")), np.asarray(s.get("z")), np.asarray(s.get("nu"))
        same = it == out["iters"]
        dz = np.abs(z - out["z"]).reshape(B, -1).max(1); dn = np.abs(nu - out["nu"]).reshape(B, -1).max(1)
        print("seed %d B %4d: same-iteration %.4f  max|dz| same %.2e  max|dnu| same %.2e  off: %d, max|dz| off %.2e, max iteration gap %d  plan %s" % (
            seed, B, same.mean(), dz[same].max(), dn[same].max(), int((~same).sum()), dz[~same].max() if (~same).any() else 0.0,
            int(np.abs(it - out["iters"]).max()), "flat2" if s.stats()["flat_split_launches"] else "other"), flush=True)
        s.close()
