"""Randomised cross-check of every engine against the CPU oracle (not part of the test suite: minutes on the GPU box).

  python scripts/fuzz_engines.py [ncases] [seed] [flat_bias]

Each case draws: a random kinematic tree (3..44 joints, depth-first or breadth-first numbered; 1-DoF joints of every type,
optionally helical joints, a free-flyer / planar root, spherical, translation, SphericalZYX, planar, unbounded-revolute and composite joints), 0..4 task
constraints with a shared or per-instance A, shared or per-instance bounds, an identity / diagonal / full reference cost
with or without v_ref -- or per-link references (UpdateReferences) --, tolerances and max_iter, the DEFAULT or the OSQP penalty
rule, optionally a spare constraint slot (a null constraint in every engine) -- and an ENGINE configuration (default
plan, k_solve only, k_tail from the first iteration, hand-over after a few iterations, lean kernel with forced escapes, lean
kernel time-sliced, the flat engine with forced escapes / in two stages).  The comparison is tests/helpers.py::assert_end_to_end: no instance is dropped."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import loik_amd  # noqa: E402
from helpers import (FIXTURE, assert_end_to_end, composite_tree, fetch_end_to_end, helical_tree, multi_task_batch, random_tree,  # noqa: E402
                     random_tree_multidof, renumber_breadth_first)
from oracle import ref  # noqa: E402

ENGINES = {
    "default": ({}, {}),
    "solve_only": ({}, dict(tail_max_instances=-1)),
    "tail_only": ({"LOIKB_LEAN": "0"}, dict(tail_max_instances=1 << 20)),
    "handover": ({}, dict(max_launch_iters=3, tail_max_instances=1 << 20)),
    "lean": ({"LOIKB_FLAT": "0"}, {}),
    "lean_escapes": ({"LOIKB_FLAT": "0", "LOIKB_LEAN_KLO": "1", "LOIKB_LEAN_DECADES": "3"}, {}),
    "lean_sliced": ({"LOIKB_FLAT": "0", "LOIKB_LEAN_SLICE": "9"}, {}),
    "flat_escapes": ({"LOIKB_LEAN_KLO": "1", "LOIKB_LEAN_DECADES": "3"}, {}),
    # k_flat2's round-robin time slicing forced (one wavefront per CU so that instances wait: requeues from the 7th iteration)
    "flat_sliced": ({"LOIKB_FLAT_SLICE": "7", "LOIKB_LEAN_WG_PER_CU": "1"}, {}),
    "flat_one_lane": ({"LOIKB_FLAT_SPLIT": "0"}, {}),   # k_flat (one joint per lane) where k_flat2 / k_flat1 would run
    # the same solve a second time on the handle: longest-first order from the first one, no fetch of the zero state (compared: the second)
    "flat_ordered": ({"LOIKB_FLAT_ORDER_HOLDOFF": "0"}, {}),
    "lean_ordered": ({"LOIKB_FLAT": "0", "LOIKB_FLAT_ORDER_HOLDOFF": "0"}, {}),   # the same with k_lean
    # round 5: decades outside a narrow table built in-wave (k_flat2<.., MUR = 2>); the lazily populated table (k_fslots builds one decade of
    # it, the rest is built by the instances that get there, into the table), plain and with forced time slices (parked build requests)
    "flat_builds": ({"LOIKB_FLAT_BUILD": "1", "LOIKB_LEAN_KLO": "1", "LOIKB_LEAN_DECADES": "2", "LOIKB_LEAN_ADAPT": "0"}, {}),
    "flat_lazy": ({"LOIKB_FLAT_BUILD": "1", "LOIKB_FLAT_WINDOW": "0,1"}, {}),
    "flat_lazy_sliced": ({"LOIKB_FLAT_BUILD": "1", "LOIKB_FLAT_WINDOW": "1,1", "LOIKB_FLAT_SLICE": "7", "LOIKB_LEAN_WG_PER_CU": "1"}, {}),
}
ENV_KEYS = ("LOIKB_LEAN", "LOIKB_FLAT", "LOIKB_FLAT_SPLIT", "LOIKB_FLAT_SLICE", "LOIKB_LEAN_KLO", "LOIKB_LEAN_DECADES", "LOIKB_LEAN_SLICE",
            "LOIKB_LEAN_WG_PER_CU", "LOIKB_FLAT_ORDER_HOLDOFF", "LOIKB_FLAT_BUILD", "LOIKB_FLAT_WINDOW", "LOIKB_LEAN_ADAPT")


def fuzz(ncase, seed, verbose=True, max_batch=3000, only=None, flat_bias=0.0, capture=None):
    """`ncase` random cases from `seed`; returns the summary dict (cases, mismatches, unconverged_only, refused, flat_cases, ...).
    max_batch bounds the batch sizes drawn (the test suite's slice uses a smaller bound); only = a case index to replay;
    flat_bias = share of the cases drawn inside the flat engine's domain (> 16 joints numbered depth-first, H_ref = h I, DEFAULT
    penalty rule) on top of those that land there by chance."""
    rng = np.random.default_rng(seed)
    say = print if verbose else (lambda *a, **k: None)
    if only is None and os.environ.get("FUZZ_ONLY"):
        only = int(os.environ["FUZZ_ONLY"])
    saved_env = {k: os.environ[k] for k in ENV_KEYS if k in os.environ}
    summary = dict(cases=0, mismatches=0, unconverged_only=0, unconverged_cases=[], refused=0, flat_cases=0, worst_dz_same=0.0, instances=0, off_count=0, by_engine={})
    for case in range(ncase):
        for_flat = bool(rng.random() < flat_bias)
        nb = int(rng.integers(17 if for_flat else 3, 45))
        seed = int(rng.integers(1, 100000))
        kind = rng.random()
        if for_flat and kind < 0.45 and rng.random() < 0.6:
            kind = 0.9   # (mostly plain trees: a multi-DoF root chain makes the tree deeper than the engine's ancestor table)
        if kind < 0.35 and nb >= 6:
            model = random_tree_multidof(seed, nb, root_freeflyer=bool(rng.random() < 0.4), n_spherical=int(rng.integers(0, 2)),
                                         n_translation=int(rng.integers(0, 2)), n_zyx=int(rng.integers(0, 2)),
                                         n_planar=int(rng.integers(0, 2)), n_rub=int(rng.integers(0, 3)),
                                         root_planar=bool(rng.random() < 0.15))
        elif kind < 0.45 and nb >= 6 and nb <= 30:   # JointModelComposite: 1..3 joints become composites of 2..4 sub-joints
            model = composite_tree(seed, nb, [int(x) for x in rng.choice(np.arange(1, nb + 1), size=int(rng.integers(1, 4)), replace=False)])
        else:
            bp = float(rng.uniform(0.45, 0.7) if for_flat else rng.uniform(0.1, 0.6))
            if rng.random() < 0.2:   # helical joints (S = [pitch a; a]) among the revolute ones
                model = helical_tree(seed, nb, int(rng.integers(1, 5)), branch_prob=bp)
            else:
                model = random_tree(seed, nb, branch_prob=bp)
                if rng.random() < 0.3 and not for_flat:
                    model, _ = renumber_breadth_first(model)
        if model.nv > 64:
            continue
        nc = int(rng.choice([0, 1, 1, 2, 3, 4]))
        nc = min(nc, model.njoints - 1)
        B = min(int(rng.choice([70, 130, 256, 700, 3000] if for_flat else [3, 70, 130, 256, 700, 3000])), max_batch)
        links = [int(x) for x in rng.choice(np.arange(1, model.njoints), size=max(nc, 1), replace=False)]
        wl = multi_task_batch(model, B, links, seed + 1, bound=0.5, nu_scale=0.4, per_instance_A=bool(rng.random() < 0.3))
        if nc == 0:
            wl["c_ids"] = np.zeros(0, dtype=np.int32); wl["Ais"] = np.zeros((0, 6, 6)); wl["bis"] = np.zeros((B, 0, 6))
        hk = int(rng.integers(0, 3))
        if for_flat and hk == 0 and rng.random() < 0.5:   # (H_ref = h I with a target: the engine's has_hv path)
            wl["H_ref"] = float(rng.uniform(0.3, 2.0)) * np.eye(6); wl["v_ref"] = 0.2 * rng.normal(size=6)
        if hk == 1:
            wl["H_ref"] = np.diag(rng.uniform(0.3, 2.0, size=6)); wl["v_ref"] = 0.2 * rng.normal(size=6)
        elif hk == 2:
            M = rng.normal(size=(6, 6)); wl["H_ref"] = M @ M.T / 6 + 0.5 * np.eye(6); wl["v_ref"] = 0.2 * rng.normal(size=6)
        if rng.random() < 0.3:
            wl["lb"] = -0.5 * (1 + 0.2 * rng.random((B, model.nv))); wl["ub"] = 0.5 * (1 + 0.2 * rng.random((B, model.nv)))
        refs = None
        if rng.random() < 0.2:   # per-link references: one weight / target per joint of the caller's model (flat engine: HM = 3)
            Hs = np.zeros((model.njoints, 6, 6)); vs = 0.2 * rng.normal(size=(model.njoints, 6))
            for i in range(model.njoints):
                M = rng.normal(size=(6, 6)); Hs[i] = M @ M.T / 6 + rng.uniform(0.2, 1.0) * np.eye(6)
            refs = (Hs, vs)
        spare = int(rng.random() < 0.2 and nc + 1 <= model.njoints - 1)   # eq_c_capacity = num_eq_c + 1
        osqp = bool(rng.random() < 0.2)   # (round 5: also inside the flat engine's domain -- k_flat2<.., MUR = 1>)
        multidof = model.nv != model.njoints - 1
        # (a tolerance of 1e-8 is below the rounding noise of the multi-DoF chain representation and of mu ~ 1e6: the iteration at
        #  which such an instance stops is then decided by that noise)
        prm = dict(FIXTURE, num_eq_c=nc, max_iter=int(rng.choice([60, 300, 1000])),
                   tol_abs=float(rng.choice([1e-4, 1e-6] if (osqp or multidof) else [1e-4, 1e-6, 1e-8])),
                   tol_rel=float(rng.choice([0.0, 1e-6])), mu_update_strat=1 if osqp else 0)
        engine = str(rng.choice(["default", "handover", "flat_escapes", "flat_one_lane", "flat_sliced", "flat_ordered", "flat_ordered", "flat_builds", "flat_lazy",
                                 "flat_lazy_sliced"] if for_flat else list(ENGINES)))
        if engine == "flat_one_lane" and (hk != 0 or refs is not None):
            engine = "default"   # (k_flat takes H_ref = h I only: such a handle would run k_lean -- covered by "lean")
        env, kw = ENGINES[engine]
        if only is not None and case != only:   # replay one case of a run (same draws)
            continue
        for k in ENV_KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                              nthreads=8, want_nu=True, refs=refs, **prm)
        s = loik_amd.BatchedLoik(model, B, **prm, **kw, eq_c_capacity=nc + spare)
        try:
            if refs is None:
                s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
            else:
                s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
                s.UpdateReferences(*refs)
                s.Solve()
            if engine in ("flat_ordered", "lean_ordered"):
                first_ordered = s.stats()["flat_ordered"]
                if refs is None:
                    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
                else:
                    s.SolveInit(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
                    s.UpdateReferences(*refs)
                    s.Solve()
                summary["ordered_launches"] = summary.get("ordered_launches", 0) + s.stats()["flat_ordered"]
                assert first_ordered == 0
        except loik_amd.LoikError as e:   # a stated limit of the library (e.g. a tree too bushy for k_solve's LDS slots): not a mismatch
            summary["refused"] += 1
            children = np.bincount(np.asarray(model.parents[1:]), minlength=model.njoints)
            say("case %3d %-12s nb %2d REFUSED: %s (max children %d)" % (case, engine, model.njoints - 1, e, children.max()), flush=True)
            s.close()
            continue
        st = s.stats()
        # Rounding budget.  1-DoF trees under the DEFAULT rule: z to 1e-7 on identical-iteration instances and the residuals to
        # 1e-9 + 1e-6 relative.  Multi-DoF chains (a different elimination order than the oracle's nv x nv blocks) and the OSQP
        # rule (mu up to 1e6, i.e. H ~ mu_eq ~ 1e10) cancel more digits in f = H v + p: z to 1e-5, residual scalars not compared.
        loose = osqp or model.nv != model.njoints - 1
        got = fetch_end_to_end(s, residuals=not loose)
        same = got["iter"] == out["iters"]
        dz = np.abs(got["z"] - out["z"]).reshape(B, -1).max(axis=1)
        if capture is not None:   # (scripts/r06/make_fuzz_fixtures.py: a replayed case's inputs and both solvers' answers)
            capture(dict(case=case, model=model, wl=wl, prm=prm, refs=refs, engine=engine, env=env, kw=kw, spare=spare, nc=nc, osqp=osqp, out=out, got=got,
                         same=same, dz=dz, mu=np.asarray(s.get("mu"), dtype=float)))
        ok, why = True, ""
        try:
            # (loose cases: the digits lost in f = H v + p scale with mu; the answer itself is only good to tol_abs -- the budget is
            #  the larger of 1e-5 and half the solver tolerance: OSQP at mu ~ 1e6 on a 50-DoF chain model reached 1.9e-5 at tol 1e-4)
            # (OSQP: mu follows the residual ratio continuously, so an instance that has NOT converged when max_iter stops it carries
            #  the rounding history of every mu it went through: three of 3000 such instances ended 1e-5 .. 9e-5 apart at tol 1e-4 --
            #  the budget under that rule is the solver tolerance itself)
            # (round 5, from 11 000 cases: an instance both solvers flag infeasible returns the iterate its tail solve stopped at -- 2.7e-7
            #  apart after 200 iterations on a helical tree at tol 1e-8 -- : 1e-6 for those; an instance that stops one iteration earlier
            #  or later is tol / mu away where mu < 1, the dual residual being mu |z_k - z_k-1|: 1.2e-4 at mu = 3e-3 under OSQP's rule)
            mu_dev = np.asarray(s.get("mu"), dtype=float)
            off_scale = np.minimum(np.maximum(1.0, 1.0 / np.maximum(mu_dev, 1e-12)), 1e3)   # (capped: OSQP clips mu at 1e-6, and a budget of 10 is no check -- ADVICE r05)
            assert_end_to_end(got, out, prm, same_frac=0.95 if B >= 70 else 0.0,
                              ztol=(max(1e-5, (1.0 if osqp else 0.5) * prm["tol_abs"]) if loose else 1e-7),
                              off_ztol=max(1e-5 if loose else 1e-6, 10 * prm["tol_abs"]), what="case %d" % case,
                              res_tol=(1e-7, 1e-5), inf_ztol=1e-6, off_scale=off_scale)  # (several task constraints: forces ~ mu_eq ~ 1e4..1e7 cancel in the residuals)
        except AssertionError as e:
            ok, why = False, str(e)[:300]
            # An instance that max_iter stopped before it converged -- in either solver -- is not an answer: its iterate depends on
            # every rounding on the way (continuously under OSQP's rule, through near-ties of the decade rule otherwise).  If the
            # comparison holds once those are set aside, the case is counted as "unconverged only", not as a mismatch.
            stopped = ((np.asarray(got["iter"]) >= prm["max_iter"] - 1) & ~np.asarray(got["converged"]).astype(bool)) | \
                      ((out["iters"] >= prm["max_iter"] - 1) & ~out["converged"])
            keep = ~stopped
            if stopped.any() and keep.any():
                try:
                    assert_end_to_end({k: np.asarray(v)[keep] for k, v in got.items()}, {k: np.asarray(v)[keep] for k, v in out.items()}, prm,
                                      same_frac=0.0, ztol=(max(1e-5, (1.0 if osqp else 0.5) * prm["tol_abs"]) if loose else 1e-7),
                                      off_ztol=max(1e-5 if loose else 1e-6, 10 * prm["tol_abs"]), what="case %d (converged or flagged only)" % case,
                                      res_tol=(1e-7, 1e-5), inf_ztol=1e-6, off_scale=off_scale[keep])
                    ok, why = True, "unconverged-only: %d instance(s) stopped by max_iter differ" % int(stopped.sum())
                    summary["unconverged_only"] += 1
                    summary["unconverged_cases"].append("case %d: %s; first failure: %s" % (case, why, str(e)[:160]))
                except AssertionError as e2:
                    why = why + " || without the instances max_iter stopped: " + str(e2)[:200]
            if only is not None:
                bad = np.argsort(-dz)[:8]
                mu = s.get("mu")
                offi = np.flatnonzero(~same)
                print("  off the oracle's iteration count (instance, iterations here / oracle, |dz|, converged here / oracle, mu):",
                      [(int(b), int(got["iter"][b]), int(out["iters"][b]), float(dz[b]), bool(got["converged"][b]), bool(out["converged"][b]), float(mu_dev[b])) for b in offi[:16]])
                print("  worst instances:", [(int(b), float(dz[b]), int(got["iter"][b]), bool(got["converged"][b]), float(mu[b]),
                                              float(np.abs(out["z"][b]).max())) for b in bad])
        summary["cases"] += 1; summary["mismatches"] += not ok; summary["flat_cases"] += st["flat_launches"] > 0
        summary["worst_dz_same"] = max(summary["worst_dz_same"], float(dz[same].max()) if same.any() else 0.0)
        summary["instances"] += B; summary["off_count"] += int((~same).sum())
        e = summary["by_engine"].setdefault(engine, dict(cases=0, mismatches=0))
        e["cases"] += 1; e["mismatches"] += not ok
        say("case %3d %-14s nb %2d nv %2d nc %d%s B %4d %s Href %s max_iter %4d tol %.0e %s: same-iteration %.3f max|dz| %.1e off %d "
              "lean %d flat %d esc %d requeue %d built %d  %s %s" % (
                  case, engine, model.njoints - 1, model.nv, nc, "+1" if spare else "  ", B, model.name[:18], "L" if refs else str(hk),
                  prm["max_iter"], prm["tol_abs"],
                  "OSQP" if osqp else "DEF ", same.mean(), dz[same].max() if same.any() else 0.0, int((~same).sum()), st["lean_launches"], st["flat_launches"],
                  st["lean_escaped"], st["lean_requeues"], st.get("flat_built", 0), "ok" if ok else "MISMATCH", why), flush=True)
        s.close()
    for k in ENV_KEYS:   # (the engines' switches are read at loikb_create: nothing of the last case may outlive the run)
        os.environ.pop(k, None)
    os.environ.update(saved_env)
    return summary

if __name__ == "__main__":
    out = fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 2024,
               flat_bias=float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
    # (a case whose only differences are instances that max_iter stopped unconverged counts as "unconverged_only", a stated limit
    #  of the library as "refused": both are reported beside the mismatches, not hidden in them)
    print(json.dumps(out))
