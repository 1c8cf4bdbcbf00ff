"""Multi-DoF joints: free-flyer, spherical and translation joints.

The reference gets them from Pinocchio's joint variant (visitors /root/reference/include/loik/loik-loid-optimized.hxx:
21-23, :91-93; `calc_aba` with an nv x nv `Dinv`).  The device solves an all-1-DoF tree in which such a joint is a chain
of 1-DoF joints about the axes of one frame with massless intermediate links; eliminating the joint's coordinates one
at a time is the same block elimination, so every iterate is the same up to rounding.  The CPU tests here prove that
on the oracle (true nv x nv joints vs the chain), the GPU tests compare the product with the true multi-DoF oracle.
"""
import numpy as np
import pytest

import loik_amd
from helpers import (FIXTURE, J_FREEFLYER, J_SPHERICAL, J_TRANSLATION, assert_close, expand_to_chains,
                     random_tree_multidof)
from loik_amd import workloads
from oracle import ref


def one_problem(model, seed, link=None, bound=0.5):
    link = model.njoints - 1 if link is None else link
    wl = workloads.make_workload(model, 1, link, seed, bound=bound, snap_prob=0.2, nu_scale=0.4)
    return dict(q=wl["q"][0], H_ref=np.eye(6), v_ref=np.zeros(6), c_ids=wl["c_ids"], Ais=wl["Ais"], bis=wl["bis"][0],
                lb=wl["lb"], ub=wl["ub"])


# the two eliminations are the same arithmetic only up to rounding, and f = H v + p cancels digits (H ~ mu_eq ~ 1e2..1e4)
TOL = 1e-7

CASES = [dict(seed=3, nb=6, root_freeflyer=True, n_spherical=0, n_translation=0),
         dict(seed=5, nb=9, root_freeflyer=True, n_spherical=1, n_translation=1),
         dict(seed=8, nb=12, root_freeflyer=False, n_spherical=2, n_translation=1),
         dict(seed=11, nb=14, root_freeflyer=True, n_spherical=2, n_translation=2)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_chain_of_massless_links_reproduces_the_multidof_joint(case):
    model = random_tree_multidof(**case)
    assert model.nv > model.njoints - 1
    p = one_problem(model, case["seed"])
    prm = dict(FIXTURE, max_iter=40, tol_abs=1e-9, tol_rel=0.0)
    a = ref.RefSolver(model, **prm)
    m1, q1, link_of = expand_to_chains(model, p["q"])
    assert m1.nv == model.nv and m1.njoints - 1 == model.nv
    b = ref.RefSolver(m1, **prm)
    a.SolveInit(p["q"], p["H_ref"], p["v_ref"], p["c_ids"], p["Ais"], p["bis"], p["lb"], p["ub"])
    b.SolveInit(q1, p["H_ref"], p["v_ref"], link_of[p["c_ids"]], p["Ais"], p["bis"], p["lb"], p["ub"])
    compared = 0
    for it in range(1, 30):
        # once the residuals are rounding noise the mu rule (a ratio of the two) is decided by that noise
        if it > 1 and min(a.scalar("primal_residual"), a.scalar("dual_residual")) < 1e-7:
            break
        compared += 1
        for s in (a, b):
            s.IterationBody()
            s.CheckConvergence()
            if it > 1:
                s.CheckFeasibility()
            s.UpdateMu()
        for name in ("nu", "z", "w", "Stf_plus_w"):
            assert_close(getattr(a, name), getattr(b, name), TOL, "%s @%d" % (name, it))
        for name in ("vis", "fis", "g", "pis"):
            assert_close(getattr(a, name)[1:], getattr(b, name)[link_of[1:]], TOL, "%s @%d" % (name, it))
        assert_close(a.His[1:], b.His[link_of[1:]], TOL, "His @%d" % it)
        for name in ("primal_residual", "dual_residual", "mu", "delta_y_qp_inf_norm", "A_qp_T_delta_y_qp_inf_norm",
                     "delta_fis_inf_norm", "delta_vis_inf_norm", "delta_nu_inf_norm", "Href_v_inf_norm", "g_inf_norm",
                     "Stf_plus_w_inf_norm", "nu_inf_norm", "tol_primal", "tol_dual", "ub_qp_T_delta_y_qp_plus",
                     "lb_qp_T_delta_y_qp_minus"):
            assert_close(a.scalar(name), b.scalar(name), TOL, "%s @%d" % (name, it))
    assert compared >= 3


@pytest.mark.parametrize("case", CASES[:2], ids=lambda c: "seed%d" % c["seed"])
def test_multidof_solve_satisfies_the_task(case):
    """first principles: the converged answer moves the task link as asked and respects the box"""
    model = random_tree_multidof(**case)
    p = one_problem(model, case["seed"] + 100)
    s = ref.RefSolver(model, **dict(FIXTURE, max_iter=400, tol_abs=1e-8, tol_rel=0.0))
    s.Solve(p["q"], p["H_ref"], p["v_ref"], p["c_ids"], p["Ais"], p["bis"], p["lb"], p["ub"])
    assert s.get_convergence_status()
    z = s.z
    v = workloads.link_velocity(model, p["q"][None], z[None], int(p["c_ids"][0]))[0]
    assert np.max(np.abs(v - p["bis"][0])) < 1e-6
    assert np.all(z <= p["ub"] + 1e-9) and np.all(z >= p["lb"] - 1e-9)
