"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/loik_ref.c).

The reference ships no golden vectors and cannot be built here, so these fixtures are NOT upstream
outputs: they freeze the oracle's own answers (inputs + outputs after k = 1, 2, 5 ADMM iterations and at the
stopping point) so that (a) the oracle cannot drift silently and (b) the GPU box, which has no /root/reference
and need not trust a rebuilt oracle, checks the HIP path against committed numbers.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import loik_amd  # noqa: E402
from helpers import (FIXTURE, feasible_batch, fixture_problem, multi_task_batch, problem_args, random_tree,  # noqa: E402
                     random_tree_multidof)
from loik_amd import workloads  # noqa: E402
from oracle import ref  # noqa: E402

STATE = ["nu", "z", "w", "vis", "fis", "g", "yis", "Aty", "His", "pis", "UDinv", "Dinv", "Stf_plus_w", "liMi"]
SCAL = ["iter", "converged", "primal_infeasible", "primal_residual", "dual_residual", "mu", "tol_primal", "tol_dual"]


def snapshot(model, params, args):
    s = ref.RefSolver(model, **params)
    s.Solve(*args)
    out = {k: s.field(k) for k in STATE}
    out.update({k: np.float64(s.scalar(k)) for k in SCAL})
    return out


def case(model, wl, nbatch, params, ks=(1, 2, 5)):
    data = dict(parents=model.parents, jtype=model.jtype, axis=model.axis, placement=model.placement,
                q=wl["q"][:nbatch], H_ref=wl["H_ref"], v_ref=wl["v_ref"], c_ids=wl["c_ids"], Ais=wl["Ais"],
                bis=wl["bis"][:nbatch], lb=wl["lb"], ub=wl["ub"])
    for key, val in params.items():
        data["param_" + key] = np.float64(val)
    for b in range(nbatch):
        for k in ks:  # state after exactly k iterations: stopping logic disabled
            p = dict(params, max_iter=k + 1, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
            for name, val in snapshot(model, p, problem_args(wl, b)).items():
                data["k%d_b%d_%s" % (k, b, name)] = val
        for name, val in snapshot(model, params, problem_args(wl, b)).items():
            data["end_b%d_%s" % (b, name)] = val
    return data


def main():
    prm = dict(FIXTURE, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
    panda = loik_amd.builtin_model("panda7")
    np.savez_compressed(os.path.join(HERE, "panda7_feasible.npz"),
                        **case(panda, feasible_batch(panda, 4, panda.njoints - 1, 21, nu_scale=0.5), 4, prm))
    talos = loik_amd.builtin_model("talos32")
    link = talos.getJointId("arm_left_7_joint")
    np.savez_compressed(os.path.join(HERE, "talos32_leftwrist.npz"),
                        **case(talos, feasible_batch(talos, 4, link, 22, nu_scale=0.5), 4, prm))
    tree = random_tree(9, 20)
    np.savez_compressed(os.path.join(HERE, "random_tree20.npz"),
                        **case(tree, feasible_batch(tree, 3, tree.njoints - 1, 23, nu_scale=0.5), 3, prm))
    # the reference fixture itself (head target, primal-infeasible -> certificate + tail solve)
    fx = fixture_problem(talos, bound=2.0)
    wl = dict(fx, q=fx["q"][None], bis=fx["bis"][None])
    np.savez_compressed(os.path.join(HERE, "talos32_reference_fixture.npz"),
                        **case(talos, wl, 1, dict(FIXTURE, max_iter=200)))
    # multi-DoF joints: floating-base Talos (free-flyer root_joint) and a random tree with a free-flyer root, a
    # spherical and a translation joint (the oracle's nv x nv joints; the device solves chains of 1-DoF joints)
    tff = loik_amd.builtin_model("talos32_freeflyer")
    link = tff.getJointId("arm_left_7_joint")
    np.savez_compressed(os.path.join(HERE, "talos32_freeflyer_leftwrist.npz"),
                        **case(tff, workloads.make_workload(tff, 2, link, 24, bound=0.5, snap_prob=0.0, nu_scale=0.4), 2, prm))
    md = random_tree_multidof(seed=5, nb=9, root_freeflyer=True, n_spherical=1, n_translation=1)
    np.savez_compressed(os.path.join(HERE, "random_multidof9.npz"),
                        **case(md, workloads.make_workload(md, 2, md.njoints - 1, 25, bound=0.5, snap_prob=0.2, nu_scale=0.4), 2, prm))
    # two simultaneous tasks (both wrists), full symmetric H_ref, non-zero v_ref
    links = [talos.getJointId("arm_left_7_joint"), talos.getJointId("arm_right_7_joint")]
    wl2 = multi_task_batch(talos, 2, links, 26)
    rng = np.random.default_rng(27)
    M = rng.normal(size=(6, 6))
    wl2["H_ref"], wl2["v_ref"] = M @ M.T / 6 + 0.5 * np.eye(6), 0.2 * rng.normal(size=6)
    np.savez_compressed(os.path.join(HERE, "talos32_two_wrists_full_href.npz"),
                        **case(talos, wl2, 2, dict(prm, num_eq_c=2)))


if __name__ == "__main__":
    main()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
