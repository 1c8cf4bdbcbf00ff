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
from helpers import FIXTURE, feasible_batch, fixture_problem, problem_args, random_tree  # noqa: E402
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


if __name__ == "__main__":
    main()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
