"""General reference cost: H_ref any symmetric PSD 6x6 and v_ref != 0 -- `SolveInit(q, H_ref, v_ref, ...)` arguments of
the reference (/root/reference/include/loik/loik-loid-optimized.hpp:335-361; UpdateReference,
ik-id-description-optimized.hpp:78-96) that its own fixture leaves at I / 0 (tests/loik-loid.cpp:118-120).
On the device a diagonal H_ref and a full one run different kernel instantiations (k_solve<T, HDIAG>)."""
import numpy as np
import pytest

import loik_amd
from helpers import FIXTURE, assert_close, feasible_batch, problem_args, random_tree, random_tree_multidof
from oracle import dense, ref


def reference_costs(seed):
    rng = np.random.default_rng(seed)
    M = rng.normal(size=(6, 6))
    full = M @ M.T / 6 + 0.5 * np.eye(6)
    return {"diag": (np.diag(rng.uniform(0.3, 2.0, size=6)), 0.3 * rng.normal(size=6)),
            "full": (full, 0.3 * rng.normal(size=6)),
            "identity_vref": (np.eye(6), 0.2 * rng.normal(size=6))}


@pytest.mark.parametrize("kind", ["diag", "full", "identity_vref"])
def test_recursive_and_dense_oracles_agree_with_a_general_reference_cost(kind):
    model = random_tree(31, 9)
    wl = feasible_batch(model, 2, model.njoints - 1, 8)
    wl["H_ref"], wl["v_ref"] = reference_costs(3)[kind]
    prm = dict(FIXTURE, max_iter=10, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    for b in range(2):
        opt, pl = ref.RefSolver(model, **prm), dense.DenseSolver(model, **prm)
        opt.Solve(*problem_args(wl, b)); pl.Solve(*problem_args(wl, b))
        for n in ("nu", "z", "w"):
            assert_close(getattr(opt, n), getattr(pl, n), 1e-9, n)
        assert_close(opt.vis[1:], pl.vis[1:], 1e-9, "vis")
        assert_close(opt.fis[1:], pl.fis[1:], 1e-8, "fis")
        assert_close(opt.scalar("dual_residual"), pl.dual_residual, 1e-8, "dual")
        assert_close(opt.scalar("primal_residual"), pl.primal_residual, 1e-9, "primal")


FIELDS = ["nu", "z", "w", "vis", "fis", "g", "yis", "Aty", "Stf_plus_w", "dual_residual_vec", "primal_residual_vec"]
SCALARS = ["primal_residual", "dual_residual", "dual_residual_v", "dual_residual_nu", "mu", "Href_v_inf_norm",
           "g_inf_norm", "delta_fis_inf_norm", "tol_dual", "tol_primal"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["diag", "full", "identity_vref"])
@pytest.mark.parametrize("which", ["talos", "tree", "multidof"])
def test_gpu_general_reference_cost(which, kind, request):
    if which == "talos":
        model = request.getfixturevalue("talos"); link = model.getJointId("arm_left_7_joint")
    elif which == "tree":
        model = random_tree(13, 23); link = model.njoints - 1
    else:
        model = random_tree_multidof(seed=5, nb=9, root_freeflyer=True, n_spherical=1, n_translation=1)
        link = model.njoints - 1
    tol = 1e-7 if which == "multidof" else 1e-9
    B = 80
    from loik_amd import workloads
    wl = workloads.make_workload(model, B, link, 44, bound=0.5, snap_prob=0.0, nu_scale=0.4)
    wl["H_ref"], wl["v_ref"] = reference_costs(7)[kind]
    for k in (1, 3, 6):
        # tol_rel > 0: tol_dual then depends on |H_ref v|, |H_ref v_ref| (hxx:548-552)
        prm = dict(FIXTURE, max_iter=k + 1, tol_abs=0.0, tol_rel=1e-30, tol_primal_inf=0.0)
        s = loik_amd.BatchedLoik(model, B, **prm)
        s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        got = {n: s.get(n) for n in FIELDS + SCALARS}
        got["His"], pis = s.His_full(), None
        for b in range(0, B, 9):
            r = ref.RefSolver(model, **prm)
            r.Solve(*problem_args(wl, b))
            for n in FIELDS:
                want = r.field(n)
                if n in ("vis", "fis", "g"):
                    want = want[1:]
                assert_close(got[n][b], want, tol, "%s b%d k%d" % (n, b, k))
            assert_close(got["His"][b], r.His[1:], tol, "His")
            for n in SCALARS:
                assert_close(got[n][b], r.scalar(n), tol, "%s b%d k%d" % (n, b, k))
        s.close()
    prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                          nthreads=4, **prm)
    for kw in (dict(tail_max_instances=-1), dict()):
        s = loik_amd.BatchedLoik(model, B, **prm, **kw)
        s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
        same = s.get("iter") == out["iters"]
        assert same.mean() >= 0.95
        assert np.array_equal(s.get("converged").astype(bool)[same], out["converged"][same])
        assert np.max(np.abs(s.get("z") - out["z"])[same]) < (1e-6 if which == "multidof" else 1e-8)
        s.close()


@pytest.mark.gpu
def test_gpu_nonsymmetric_reference_cost_is_refused(talos):
    """upstream's SE3actOn symmetrises silently; the device stores 21 entries and says so"""
    wl = feasible_batch(talos, 4, talos.njoints - 1, 3)
    H = np.eye(6); H[0, 1] = 0.3
    s = loik_amd.BatchedLoik(talos, 4, **dict(FIXTURE, max_iter=5))
    with pytest.raises(loik_amd.LoikError) as e:
        s.Solve(wl["q"], H, wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    assert e.value.code == -23
    s.close()
