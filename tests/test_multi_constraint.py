"""Several simultaneous task constraints: `num_eq_c > 1` is a constructor argument of the reference
(/root/reference/include/loik/loik-loid-optimized.hpp:129-134; FwdPass1 / DualUpdate loop over
`active_task_constraint_ids_`, loik-loid-optimized.hxx:321-334, :410-451) although its own fixture uses one."""
import numpy as np
import pytest

import loik_amd
from helpers import (FIXTURE, assert_close, assert_end_to_end, fetch_end_to_end, multi_task_batch, problem_args,
                     random_tree)
from oracle import dense, ref


def _links(model, nc):
    """nc distinct links, leaves first (so that the constraints sit on different branches where there are any)"""
    children = {i: 0 for i in range(model.njoints)}
    for i in range(1, model.njoints):
        children[int(model.parents[i])] += 1
    leaves = [i for i in range(model.njoints - 1, 0, -1) if children[i] == 0]
    rest = [i for i in range(model.njoints - 1, 0, -1) if children[i] != 0]
    return (leaves + rest)[:nc]


@pytest.mark.parametrize("nc", [2, 3])
def test_recursive_and_dense_oracles_agree_with_several_constraints(nc):
    """the reference's relational pin (opt == plain, tests/loik-loid.cpp:305-556) with nc > 1"""
    model = random_tree(21 + nc, 11)
    links = _links(model, nc)
    wl = multi_task_batch(model, 2, links, 5 + nc)
    prm = dict(FIXTURE, num_eq_c=nc, max_iter=12, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
    for b in range(2):
        opt = ref.RefSolver(model, **prm)
        pl = dense.DenseSolver(model, **prm)
        args = problem_args(wl, b)
        opt.Solve(*args)
        pl.Solve(*args)
        assert opt.get_iter() == pl.get_iter() == 11
        assert_close(opt.nu, pl.nu, 1e-9, "nu"); assert_close(opt.z, pl.z, 1e-9, "z"); assert_close(opt.w, pl.w, 1e-9, "w")
        assert_close(opt.vis[1:], pl.vis[1:], 1e-9, "vis"); assert_close(opt.fis[1:], pl.fis[1:], 1e-8, "fis")
        for c, cid in enumerate(links):
            assert_close(opt.yis[c], pl.yis[cid], 1e-8, "yis")
        assert_close(opt.scalar("primal_residual"), pl.primal_residual, 1e-9, "primal")
        assert_close(opt.scalar("dual_residual"), pl.dual_residual, 1e-8, "dual")


# ---------------------------------------------------------------------------------------------------------------------
FIELDS = ["nu", "z", "w", "vis", "fis", "g", "yis", "Aty", "Stf_plus_w", "primal_residual_vec", "dual_residual_vec"]
SCALARS = ["primal_residual", "dual_residual", "primal_residual_task", "primal_residual_slack", "mu",
           "delta_yis_inf_norm", "Av_inf_norm", "g_inf_norm", "delta_fis_inf_norm"]


def _gpu(model, wl, prm, **kw):
    s = loik_amd.BatchedLoik(model, wl["q"].shape[0], **prm, **kw)
    s.Solve(wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("which,nc,per_instance_A", [("talos", 2, False), ("talos", 3, True), ("tree", 2, True),
                                                     ("tree", 4, False), ("panda9", 2, False)])
def test_gpu_several_constraints(which, nc, per_instance_A, request):
    model = random_tree(9, 19) if which == "tree" else request.getfixturevalue(which)
    if which == "talos":
        links = [model.getJointId(n) for n in ("arm_left_7_joint", "arm_right_7_joint", "head_2_joint")][:nc]
    else:
        links = _links(model, nc)
    B = 90
    wl = multi_task_batch(model, B, links, 3 + nc, per_instance_A=per_instance_A)
    for k in (1, 2, 5):
        prm = dict(FIXTURE, num_eq_c=nc, max_iter=k + 1, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
        s = _gpu(model, wl, prm)
        got = {n: s.get(n) for n in FIELDS + SCALARS}
        got["His"] = s.His_full()
        for b in range(0, B, 11):
            r = ref.RefSolver(model, **prm)
            r.Solve(*problem_args(wl, b))
            for n in FIELDS:
                want = r.field(n)
                if n in ("vis", "fis", "g"):
                    want = want[1:]
                assert_close(got[n][b], want, 1e-9, "%s b%d k%d" % (n, b, k))
            assert_close(got["His"][b], r.His[1:], 1e-9, "His")
            for n in SCALARS:
                assert_close(got[n][b], r.scalar(n), 1e-9, "%s b%d k%d" % (n, b, k))
        s.close()
    # end to end: stopping logic, the solve kernel alone and the tail kernel alone
    prm = dict(FIXTURE, num_eq_c=nc, max_iter=400, tol_abs=1e-6, tol_rel=0.0)
    Ais = wl["Ais"] if per_instance_A else wl["Ais"]
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], Ais, wl["bis"], wl["lb"], wl["ub"],
                          nthreads=4, want_nu=True, **prm)
    for kw in (dict(tail_max_instances=-1), dict()):
        s = _gpu(model, wl, prm, **kw)
        assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=1e-9, what="nc=%d" % nc)
        assert s.stats()["tail_instances"] == (0 if kw else B)
        s.close()


@pytest.mark.gpu
def test_gpu_tailored_update_of_one_of_several_constraints(talos):
    """Solve(q, c_id, Ai, bi) replaces ONE constraint of the active set (ik-id-description-optimized.hpp:178-218)"""
    links = [talos.getJointId(n) for n in ("arm_left_7_joint", "arm_right_7_joint")]
    B = 40
    wl = multi_task_batch(talos, B, links, 12)
    prm = dict(FIXTURE, num_eq_c=2, max_iter=300, tol_abs=1e-6, tol_rel=0.0, warm_start=True)
    s = _gpu(talos, wl, prm)
    refs = []
    for b in range(0, B, 7):
        r = ref.RefSolver(talos, **prm)
        r.Solve(*problem_args(wl, b))
        refs.append((b, r))
    b2 = 0.7 * wl["bis"][:, 1]
    s.Solve(wl["q"], links[1], wl["Ais"][1], b2)
    z, it = s.get("z"), s.get("iter")
    for b, r in refs:
        r.Solve(wl["q"][b], links[1], wl["Ais"][1], b2[b])
        assert it[b] == r.get_iter()
        assert_close(z[b], r.z, 1e-8, "z b%d" % b)
    s.close()


def _no_constraint_problem(model, B, seed):
    rng = np.random.default_rng(seed)
    nv = model.nv
    lb = rng.uniform(-0.5, 0.2, size=(B, nv))      # some boxes exclude 0: the answer is then on the boundary
    ub = lb + rng.uniform(0.1, 0.6, size=(B, nv))
    return dict(q=model.random_configurations(rng, B), H_ref=np.eye(6), v_ref=0.1 * rng.normal(size=6),
                c_ids=np.zeros(0, dtype=np.int32), Ais=np.zeros((0, 6, 6)), bis=np.zeros((B, 0, 6)), lb=lb, ub=ub)


def test_oracle_without_constraints(panda7):
    """num_eq_c = 0 (empty active set): only the reference cost and the box remain"""
    p = _no_constraint_problem(panda7, 3, 1)
    prm = dict(FIXTURE, num_eq_c=0, max_iter=500, tol_abs=1e-8, tol_rel=0.0)
    for b in range(3):
        r = ref.RefSolver(panda7, **prm)
        r.Solve(p["q"][b], p["H_ref"], p["v_ref"], p["c_ids"], p["Ais"], p["bis"][b], p["lb"][b], p["ub"][b])
        assert r.get_convergence_status()
        assert np.all(r.z >= p["lb"][b] - 1e-12) and np.all(r.z <= p["ub"][b] + 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["panda7", "talos"])
def test_gpu_without_constraints(which, request):
    model = request.getfixturevalue(which)
    B = 70
    p = _no_constraint_problem(model, B, 2)
    for max_iter, kw in [(4, {}), (400, {}), (400, dict(tail_max_instances=-1))]:
        prm = dict(FIXTURE, num_eq_c=0, max_iter=max_iter, tol_abs=1e-7, tol_rel=0.0)
        s = loik_amd.BatchedLoik(model, B, **prm, **kw)
        s.Solve(p["q"], p["H_ref"], p["v_ref"], p["c_ids"], p["Ais"], p["bis"], p["lb"], p["ub"])
        it, z, nu, conv = s.get("iter"), s.get("z"), s.get("nu"), s.get("converged")
        for b in range(0, B, 6):
            r = ref.RefSolver(model, **prm)
            r.Solve(p["q"][b], p["H_ref"], p["v_ref"], p["c_ids"], p["Ais"], p["bis"][b], p["lb"][b], p["ub"][b])
            assert it[b] == r.get_iter() and bool(conv[b]) == r.get_convergence_status()
            assert_close(z[b], r.z, 1e-9, "z"); assert_close(nu[b], r.nu, 1e-9, "nu")
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["talos32", "talos44"])
def test_whole_body_four_tasks_in_the_lean_engine(robot):
    """both wrists + both feet (num_eq_c = 4) on the Talos topology incl. the 44-DoF tree of the reference's fixture file
    (talos_full_v2.urdf, tests/loik-loid.cpp:110-111: four joints on each wrist link, depth 11): the default engine is
    the lean kernel (single-wavefront workgroups: seven per CU fit the LDS) -- k iterations and end to end vs the oracle"""
    from loik_amd import workloads
    model = loik_amd.builtin_model(robot)
    assert (model.nv == 44 and model.names[-1] == "head_2_joint") if robot == "talos44" else model.nv == 32
    B = 320
    wl = workloads.talos_wholebody(B, seed=5, model=model)
    args = (wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"])
    for k in (1, 4):
        prm = dict(wl["params"], max_iter=k + 1, tol_abs=0.0, tol_primal_inf=0.0)
        s = loik_amd.BatchedLoik(model, B, **prm)
        s.Solve(*args)
        assert s.stats()["lean_launches"] >= 1
        got = {n: s.get(n) for n in FIELDS + SCALARS}
        got["His"] = s.His_full()
        for b in range(0, B, 53):
            r = ref.RefSolver(model, **prm)
            r.Solve(*problem_args(wl, b))
            for n in FIELDS:
                want = r.field(n)
                if n in ("vis", "fis", "g"):
                    want = want[1:]
                assert_close(got[n][b], want, 1e-9, "%s b%d k%d" % (n, b, k))
            assert_close(got["His"][b], r.His[1:], 1e-9, "His")
            for n in SCALARS:
                assert_close(got[n][b], r.scalar(n), 1e-9, "%s b%d k%d" % (n, b, k))
        s.close()
    prm = dict(wl["params"], max_iter=500)
    out = ref.solve_batch(model, *args, nthreads=4, want_nu=True, **prm)
    s = loik_amd.BatchedLoik(model, B, **prm)
    s.Solve(*args)
    st = s.stats()
    assert st["lean_launches"] >= 1 and st["lean_escaped"] == 0 and st["tail_instances"] == B, st
    assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=1e-9, what="whole body " + robot)
    # every task met where the solver converged (first principles)
    conv = s.get("converged").astype(bool)
    assert conv.mean() > 0.5
    for c, link in enumerate(wl["c_ids"]):
        v = workloads.link_velocity(model, wl["q"], s.get("z"), int(link))
        assert np.max(np.abs(v - wl["bis"][:, c])[conv]) < 1e-5
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("robot,nc", [("talos32", 6), ("talos32", 10), ("talos32", 11), ("talos32", 16),
                                      ("talos44", 6), ("talos44", 10), ("talos44", 11)])
@pytest.mark.parametrize("per_instance_A", [False, True])
def test_gpu_many_constraints_and_the_flat_engines_limit(robot, nc, per_instance_A):
    """VERDICT r04 #5a / ADVICE r03 (high): k_flat2 / k_flat1 update the task constraints on lanes 6 c + k of the wavefront, ten constraints
    in one pass; the plan sends more to the engines that loop over them (loik_host.hip::plan_engines).  The gate went in with no test
    above num_eq_c = 4: here 6 and 10 (lanes 24..59: every row of the last constraint's block, the half-filled tail of the wavefront)
    on both flat engines, and 11 / 16 on the fallback -- k iterations against the oracle field by field, then end to end.
    num_eq_c is the reference's constructor argument (loik-loid-optimized.hpp:129-134); FwdPass1 / DualUpdate loop over the active
    constraints (hxx:321-334, :410-451)."""
    model = loik_amd.builtin_model(robot)
    # nc distinct links spread over the tree (wrists, feet, head first, then joints along the chains)
    links = _links(model, nc)
    B = 128
    wl = multi_task_batch(model, B, links, 40 + nc, per_instance_A=per_instance_A, nu_scale=0.3)
    on_flat = nc <= 10
    for k in (1, 2, 5):
        prm = dict(FIXTURE, num_eq_c=nc, max_iter=k + 1, tol_abs=0.0, tol_rel=0.0, tol_primal_inf=0.0)
        s = _gpu(model, wl, prm)
        st = s.stats()
        if on_flat:
            assert ("k_flat2" if robot == "talos32" else "k_flat1") in s.plan(), s.plan()
            assert st["flat_launches"] >= 1 and (st["flat_split_launches"] >= 1) == (robot == "talos32") and st["tail_instances"] == B, (s.plan(), st)
        else:
            assert "more task constraints than the flat engine" in s.plan(), s.plan()
            assert st["flat_launches"] == 0, st
        got = {n: s.get(n) for n in FIELDS + SCALARS}
        got["His"] = s.His_full()
        for b in range(0, B, 17):
            r = ref.RefSolver(model, **prm)
            r.Solve(*problem_args(wl, b))
            for n in FIELDS:
                want = r.field(n)
                if n in ("vis", "fis", "g"):
                    want = want[1:]
                assert_close(got[n][b], want, 1e-9, "%s b%d k%d" % (n, b, k))
            assert_close(got["His"][b], r.His[1:], 1e-9, "His")
            for n in SCALARS:
                assert_close(got[n][b], r.scalar(n), 1e-9, "%s b%d k%d" % (n, b, k))
        s.close()
    prm = dict(FIXTURE, num_eq_c=nc, max_iter=300, tol_abs=1e-6, tol_rel=0.0)
    out = ref.solve_batch(model, wl["q"], wl["H_ref"], wl["v_ref"], wl["c_ids"], wl["Ais"], wl["bis"], wl["lb"], wl["ub"],
                          nthreads=8, want_nu=True, **prm)
    s = _gpu(model, wl, prm)
    assert (s.stats()["flat_launches"] >= 1) == on_flat
    assert_end_to_end(fetch_end_to_end(s), out, prm, same_frac=0.99, ztol=1e-9, what="%s nc=%d" % (robot, nc))
    s.close()
