"""Synthetic IK workloads of BASELINE.json / SURVEY.md 8(d) (host-side input generation only, numpy).

The targets are built to be *feasible*: b = A * J_c(q) * nu_star with nu_star inside the box, so that "solves to
1e-6 residual" is well defined (the reference fixture's head target is primal-infeasible).
J_c(q) nu_star is evaluated by propagating link velocities down the kinematic chain,
v_i = liMi^-1 . v_parent + S_i nu_i (the same recursion as the reference's forward pass,
/root/reference/include/loik/loik-loid-optimized.hxx:125-134) -- vectorised over the batch.
"""
import numpy as np

J_RX, J_RY, J_RZ, J_PX, J_PY, J_PZ, J_RU, J_PU = 1, 2, 3, 4, 5, 6, 7, 8
J_FREEFLYER, J_SPHERICAL, J_TRANSLATION = 9, 10, 11
J_SPHERICAL_ZYX, J_PLANAR, J_RUBX, J_RUBY, J_RUBZ = 12, 13, 14, 15, 16
J_RUBU = 18
J_HX, J_HY, J_HZ, J_HU = 19, 20, 21, 22   # JointModelHelical*: M = (Rot(a, q), pitch q a), S = [pitch a; a]

# the reference fixture's solver parameters, /root/reference/tests/loik-loid.cpp:91-105
FIXTURE_PARAMS = dict(tol_primal_inf=1e-2, tol_dual_inf=1e-2, tol_tail_solve=1e-1, rho=1e-5, mu=1e-2,
                      mu_equality_scale_factor=1e4, mu_update_strat=0, num_eq_c=1, eq_c_dim=6, warm_start=False)


def _rot(jtype, axis, q):
    """batched joint rotation M(q).rotation(): [B,3,3]"""
    B = q.shape[0]
    c, s = np.cos(q), np.sin(q)
    R = np.zeros((B, 3, 3))
    if jtype in (J_RX, J_RY, J_RZ):
        a = np.zeros(3)
        a[jtype - J_RX] = 1.0
    else:
        a = np.asarray(axis, dtype=float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R[:] = np.eye(3)[None] * c[:, None, None] + (1 - c)[:, None, None] * np.outer(a, a)[None] + s[:, None, None] * K[None]
    return R


def quat_rot(qt):
    """batched Eigen::Quaternion::toRotationMatrix for (x, y, z, w) rows: [B,3,3]"""
    x, y, z, w = qt[:, 0], qt[:, 1], qt[:, 2], qt[:, 3]
    R = np.empty((qt.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


J_COMPOSITE = 17
_NQ = {9: 7, 10: 4, 11: 3, 12: 3, 13: 4, 14: 2, 15: 2, 16: 2, 18: 2}   # (free-flyer, spherical, translation, ZYX, planar, RUB*)
_NV = {9: 6, 10: 3, 11: 3, 12: 3, 13: 3}


class _Chain:
    """the same kinematics with every JointModelComposite written out as its sub-joints (one joint each, in order, with the
    composite's internal placements): link velocities do not care that the bodies in between do not exist; q / nu keep
    their layout (a composite's coordinates are its sub-joints' in order)"""

    def __init__(self, model):
        parents, jtype, axis, placement, idx_q, idx_v = [0], [0], [np.zeros(3)], [np.asarray(model.placement[0])], [0], [0]
        pitch = [0.0]   # (helical joints and helical sub-joints)
        mp = getattr(model, "pitch", None)
        self.link_of = [0]
        for i in range(1, model.njoints):
            par = self.link_of[int(model.parents[i])]
            if int(model.jtype[i]) != J_COMPOSITE:
                parents.append(par); jtype.append(int(model.jtype[i])); axis.append(np.asarray(model.axis[i]))
                placement.append(np.asarray(model.placement[i])); idx_q.append(int(model.idx_q[i])); idx_v.append(int(model.idx_v[i]))
                pitch.append(0.0 if mp is None else float(mp[i]))
            else:
                iq, iv = int(model.idx_q[i]), int(model.idx_v[i])
                Pj = np.asarray(model.placement[i]); Rj, tj = Pj[:9].reshape(3, 3), Pj[9:]
                for k, (st, a, P) in enumerate(model.composite[i]):
                    P = np.asarray(P, dtype=float)
                    if k == 0:   # jointPlacements[i] * placement of the first sub-joint
                        P = np.concatenate([(Rj @ P[:9].reshape(3, 3)).ravel(), tj + Rj @ P[9:]])
                    parents.append(par if k == 0 else len(parents) - 1)
                    jtype.append(int(st)); axis.append(np.asarray(a, dtype=float)); placement.append(P)
                    idx_q.append(iq); idx_v.append(iv)
                    pitch.append(float(model.comp_pitch[int(model.comp_first[i]) + k]))
                    iq += _NQ.get(int(st), 1)
                    iv += _NV.get(int(st), 1)
            self.link_of.append(len(parents) - 1)
        self.njoints = len(parents)
        self.parents, self.jtype = np.array(parents), np.array(jtype)
        self.axis, self.placement = np.array(axis), np.array(placement)
        self.pitch = np.array(pitch)
        self.idx_q, self.idx_v = np.array(idx_q), np.array(idx_v)
        self.composite = None


def link_velocity(model, q, nu, link):
    """spatial velocity [B,6] (Pinocchio order [linear; angular], link frame) of `link` for joint velocities nu"""
    if getattr(model, "composite", None):
        ch = getattr(model, "_chain_cache", None)
        if ch is None:
            ch = model._chain_cache = _Chain(model)
        return link_velocity(ch, q, nu, ch.link_of[int(link)])
    B = q.shape[0]
    chain = []
    i = int(link)
    while i > 0:
        chain.append(i)
        i = int(model.parents[i])
    v = np.zeros((B, 6))
    for i in reversed(chain):
        jt = int(model.jtype[i])
        P = model.placement[i]
        Rp, tp = P[:9].reshape(3, 3), P[9:]
        iq, iv = int(model.idx_q[i]), int(model.idx_v[i])
        qi, nui = q[:, iq], nu[:, iv]
        if jt in (J_SPHERICAL_ZYX, J_PLANAR):
            # JointModelSphericalZYX::calc / JointModelPlanar::calc: M(q) and the motion subspace S (q-dependent for ZYX)
            if jt == J_SPHERICAL_ZYX:
                c0, s0 = np.cos(q[:, iq]), np.sin(q[:, iq]); c1, s1 = np.cos(q[:, iq + 1]), np.sin(q[:, iq + 1])
                c2, s2 = np.cos(q[:, iq + 2]), np.sin(q[:, iq + 2])
                Rj = np.empty((B, 3, 3))
                Rj[:, 0, 0] = c0 * c1; Rj[:, 0, 1] = c0 * s1 * s2 - s0 * c2; Rj[:, 0, 2] = c0 * s1 * c2 + s0 * s2
                Rj[:, 1, 0] = s0 * c1; Rj[:, 1, 1] = s0 * s1 * s2 + c0 * c2; Rj[:, 1, 2] = s0 * s1 * c2 - c0 * s2
                Rj[:, 2, 0] = -s1; Rj[:, 2, 1] = c1 * s2; Rj[:, 2, 2] = c1 * c2
                tj = np.zeros((B, 3))
                Sa = np.zeros((B, 3, 3))  # angular subspace, columns
                Sa[:, 0, 0] = -s1; Sa[:, 1, 0] = c1 * s2; Sa[:, 2, 0] = c1 * c2
                Sa[:, 1, 1] = c2; Sa[:, 2, 1] = -s2
                Sa[:, 0, 2] = 1.0
                dv = np.concatenate([np.zeros((B, 3)), np.einsum("bij,bj->bi", Sa, nu[:, iv:iv + 3])], axis=1)
            else:
                c, s_ = q[:, iq + 2], q[:, iq + 3]
                Rj = np.zeros((B, 3, 3)); Rj[:, 0, 0] = c; Rj[:, 0, 1] = -s_; Rj[:, 1, 0] = s_; Rj[:, 1, 1] = c; Rj[:, 2, 2] = 1.0
                tj = np.concatenate([q[:, iq:iq + 2], np.zeros((B, 1))], axis=1)
                dv = np.zeros((B, 6)); dv[:, 0] = nu[:, iv]; dv[:, 1] = nu[:, iv + 1]; dv[:, 5] = nu[:, iv + 2]
            R = Rp[None] @ Rj
            t = tp[None] + tj @ Rp.T
            lin, ang = v[:, :3], v[:, 3:]
            d = lin - np.cross(t, ang)
            v = np.concatenate([np.einsum("bji,bj->bi", R, d), np.einsum("bji,bj->bi", R, ang)], axis=1) + dv
            continue
        if jt in (J_RUBX, J_RUBY, J_RUBZ, J_RUBU):  # JointModelRevoluteUnbounded(Unaligned): q = (cos, sin)
            qi = np.arctan2(q[:, iq + 1], q[:, iq])
            jt = J_RU if jt == J_RUBU else J_RX + (jt - J_RUBX)
        if jt in (J_FREEFLYER, J_SPHERICAL, J_TRANSLATION):
            # M(q) = (R(quat), t) ; S = I6 | [0; I3] | [I3; 0]: the joint velocity is added in the child frame
            Rj = quat_rot(q[:, iq + 3:iq + 7]) if jt == J_FREEFLYER else (
                quat_rot(q[:, iq:iq + 4]) if jt == J_SPHERICAL else np.broadcast_to(np.eye(3), (B, 3, 3)))
            tj = q[:, iq:iq + 3] if jt != J_SPHERICAL else np.zeros((B, 3))
            R = Rp[None] @ Rj
            t = tp[None] + tj @ Rp.T
            lin, ang = v[:, :3], v[:, 3:]
            d = lin - np.cross(t, ang)
            v = np.concatenate([np.einsum("bji,bj->bi", R, d), np.einsum("bji,bj->bi", R, ang)], axis=1)
            if jt == J_FREEFLYER:
                v = v + nu[:, iv:iv + 6]
            elif jt == J_SPHERICAL:
                v[:, 3:] += nu[:, iv:iv + 3]
            else:
                v[:, :3] += nu[:, iv:iv + 3]
            continue
        pitch = 0.0
        if jt in (J_HX, J_HY, J_HZ, J_HU):   # the revolute joint about the same axis + the translation / velocity along it
            pitch = float(model.pitch[i])
            jt = J_RU if jt == J_HU else J_RX + (jt - J_HX)
        if jt in (J_RX, J_RY, J_RZ, J_RU):
            R = Rp[None] @ _rot(jt, model.axis[i], qi)
            t = np.broadcast_to(tp, (B, 3))
            if pitch:
                a = np.asarray(model.axis[i], dtype=float) if jt == J_RU else np.eye(3)[jt - J_RX]
                t = tp[None] + (pitch * qi)[:, None] * (Rp @ a)[None]
        else:
            a = np.zeros(3)
            if jt == J_PU:
                a = np.asarray(model.axis[i], dtype=float)
            else:
                a[jt - J_PX] = 1.0
            R = np.broadcast_to(Rp, (B, 3, 3))
            t = tp[None] + (Rp @ a)[None] * qi[:, None]
        lin, ang = v[:, :3], v[:, 3:]
        d = lin - np.cross(t, ang)
        vl = np.einsum("bji,bj->bi", R, d)
        va = np.einsum("bji,bj->bi", R, ang)
        v = np.concatenate([vl, va], axis=1)
        S = np.zeros(6)
        if jt in (J_RX, J_RY, J_RZ):
            S[3 + jt - J_RX] = 1
        elif jt in (J_PX, J_PY, J_PZ):
            S[jt - J_PX] = 1
        elif jt == J_RU:
            S[3:] = model.axis[i]
        else:
            S[:3] = model.axis[i]
        if pitch:
            S[:3] = pitch * S[3:]
        v = v + S[None] * nui[:, None]
    return v


def make_workload(model, batch, link, seed, bound=0.5, snap_prob=0.25, nu_scale=None):
    """Feasible single-task workload: q ~ U(q_lo,q_hi); nu_star ~ U(-bound,bound), each component snapped to
    +-bound w.p. snap_prob (so that joint-velocity limits are active); A = I6; b = J_link(q) nu_star; box = +-bound."""
    rng = np.random.default_rng(seed)
    nq, nv = model.nq, model.nv
    q = model.random_configurations(rng, batch)
    s = bound if nu_scale is None else nu_scale
    nu_star = rng.uniform(-s, s, size=(batch, nv))
    snap = rng.random((batch, nv)) < snap_prob
    nu_star = np.where(snap, np.sign(nu_star) * s, nu_star)
    b = link_velocity(model, q, nu_star, link)
    return dict(q=q, H_ref=np.eye(6), v_ref=np.zeros(6), c_ids=np.array([link], dtype=np.int32),
                Ais=np.eye(6).reshape(1, 6, 6), bis=b.reshape(batch, 1, 6), lb=-bound * np.ones(nv),
                ub=bound * np.ones(nv), nu_star=nu_star)


def talos_c3(batch, seed=0x101C + 3, model=None):
    """BASELINE config "Talos humanoid with joint-limit inequality constraints, batch=65536, adaptive rho, fp64":
    task on the left wrist (arm_left_7_joint; support chain torso_1-2 + arm_left_1-7 = 9 joints), SURVEY.md 8(d) C3."""
    if model is None:
        from . import builtin_model
        model = builtin_model("talos32")
    link = model.getJointId("arm_left_7_joint")
    wl = make_workload(model, batch, link, seed, bound=0.5, snap_prob=0.0)
    wl["params"] = dict(FIXTURE_PARAMS, max_iter=1000, tol_abs=1e-6, tol_rel=0.0)
    wl["model"] = model
    wl["name"] = "talos32_leftwrist_B%d_tol1e-6_adaptive_mu_fp64" % batch
    return wl


def panda_c2(batch=4096, seed=0x101C + 2, model=None):
    """BASELINE config "Panda 7-DoF, batch=4096, fixed 50 ADMM iters, fp64": mu frozen, no stopping logic."""
    if model is None:
        from . import builtin_model
        model = builtin_model("panda7")
    link = model.njoints - 1
    wl = make_workload(model, batch, link, seed, bound=4.0, snap_prob=0.0, nu_scale=1.0)
    wl["params"] = dict(FIXTURE_PARAMS, max_iter=51, tol_abs=0.0, tol_rel=0.0)
    wl["model"] = model
    wl["name"] = "panda7_B%d_fixed50_fp64" % batch
    return wl


def talos_c4(batch, T=4, seed=0x101C + 4, model=None):
    """BASELINE config "Talos batch=1,048,576 sharded across 8xMI355X (sampling-planner workload; per-instance early stop
    + lane compaction)": `batch` is ONE GPU's share (131072).  T successive targets per instance through the tailored
    warm-started entry Solve(q, c_id, Ai, bi) (loik-loid-optimized.hpp:596-695): step t has its own configuration and
    its own feasible wrist twist.  Returns the C3-style workload of step 0 plus `steps` = [(q_t, bis_t)], SURVEY.md 8(d) C4."""
    wl = talos_c3(batch, seed=seed, model=model)
    model = wl["model"]
    link = int(wl["c_ids"][0])
    steps = [(wl["q"], wl["bis"])]
    for t in range(1, T):
        w = make_workload(model, batch, link, seed + 1000 * t, bound=0.5, snap_prob=0.0)
        steps.append((w["q"], w["bis"]))
    wl["steps"] = steps
    wl["params"] = dict(wl["params"], warm_start=True)
    wl["name"] = "talos32_leftwrist_B%d_T%d_tailored_warm_start_fp64" % (batch, T)
    return wl


def panda_c5(batch=65536, seed=0x101C + 5, tol=1e-3, model=None):
    """BASELINE config "fp32 path: Panda batch=65536, fp32 vs fp64 tolerance/throughput trade-off": the C2 generator with
    the stopping logic on, at a tolerance fp32 can reach."""
    if model is None:
        from . import builtin_model
        model = builtin_model("panda7")
    link = model.njoints - 1
    wl = make_workload(model, batch, link, seed, bound=4.0, snap_prob=0.0, nu_scale=1.0)
    wl["params"] = dict(FIXTURE_PARAMS, max_iter=200, tol_abs=tol, tol_rel=0.0)
    wl["model"] = model
    wl["name"] = "panda7_B%d_tol%g" % (batch, tol)
    return wl


def talos_wholebody(batch, seed=0x101C + 6, model=None, links=("arm_left_7_joint", "arm_right_7_joint", "leg_left_6_joint",
                                                               "leg_right_6_joint")):
    """Whole-body variant of the headline workload: FOUR simultaneous 6-D tasks (both wrists, both feet; num_eq_c = 4, the
    constructor argument of the reference, loik-loid-optimized.hpp:129-134) on the Talos topology -- by default the 44-DoF
    tree of talos_full_v2.urdf, the file the reference's fixture loads (tests/loik-loid.cpp:110-111).  b_c = J_c(q) nu* for ONE
    common nu* ~ U(-0.5, 0.5)^nv -> jointly feasible; A_c = I, box +-0.5, the C3 stopping rule.  Every joint of the robot is
    in some task's support chain except the head."""
    if model is None:
        from . import builtin_model
        model = builtin_model("talos44")
    ids = [model.getJointId(n) for n in links]
    wl = make_workload(model, batch, ids[0], seed, bound=0.5, snap_prob=0.0)
    b = np.empty((batch, len(ids), 6))
    for c, link in enumerate(ids):
        b[:, c] = link_velocity(model, wl["q"], wl["nu_star"], link)
    wl["c_ids"] = np.array(ids, dtype=np.int32)
    wl["Ais"] = np.tile(np.eye(6), (len(ids), 1, 1))
    wl["bis"] = b
    wl["params"] = dict(FIXTURE_PARAMS, max_iter=1000, tol_abs=1e-6, tol_rel=0.0, num_eq_c=len(ids))
    wl["model"] = model
    wl["name"] = "%s_wholebody_%dtasks_B%d_tol1e-6_adaptive_mu_fp64" % (model.name, len(ids), batch)
    return wl
