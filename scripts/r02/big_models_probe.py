import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, time
import loik_amd
from helpers import FIXTURE, assert_end_to_end, fetch_end_to_end, multi_task_batch, random_tree
from oracle import ref
for nb, seed in ((80, 3), (120, 4), (65, 5)):
    model = random_tree(seed, nb, branch_prob=0.3)
    B = 300
    wl = multi_task_batch(model, B, [nb // 3, nb], 7, bound=0.5, nu_scale=0.3)
    prm = dict(FIXTURE, max_iter=300, tol_abs=1e-6, tol_rel=0.0, num_eq_c=2)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"], nthreads=8, want_nu=True, **prm)
    s = loik_amd.BatchedLoik(model, B, **prm)
    t = time.perf_counter()
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    dt = time.perf_counter() - t
    same = assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.95, ztol=2e-6, off_ztol=1e-5, what="nb %d" % nb, res_tol=(1e-7, 1e-5))
    print("nb", nb, "ok: same-iteration", same.mean(), "ms", round(dt * 1e3, 2), "|", s.plan())
    s.close()
