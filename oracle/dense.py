"""Dense-QP CPU oracle (numpy) -- TEST INFRASTRUCTURE ONLY.

Restates the reference's *plain* solver `FirstOrderLoikTpl<double>` and its explicit standard-QP
formulation `IkProblemStandardQPFormulation` (x=[v;nu], y=[f;y;w], z=[0;b;z]):

  * passes ............ include/loik/loik-loid.hxx:16-189   (general action matrices, no symmetry assumption)
  * QP assembly ....... include/loik/ik-id-description.hpp:411-491
  * per-iteration QP .. include/loik/ik-id-description.hpp:499-539
  * residuals ......... include/loik/loik-loid.hxx:207-295  (dual residual = P x + q + A^T y, line 280)
  * tolerances ........ include/loik/loik-loid.hxx:302-324
  * infeasibility ..... include/loik/loik-loid.hxx:331-367  (primal AND dual certificates)
  * loops ............. include/loik/loik-loid.hpp:150-183 (ResetSolver), :251-347 (tail), :362-460 (Solve)

This is the ground truth the reference's own tests compare the optimized solver against
(tests/loik-loid.cpp:305-556, :559-671).  It shares no code with oracle/loik_ref.c: forward kinematics,
action matrices and every pass are written independently here, so agreement of the two at 1e-10
abs-or-rel reproduces the reference's relational pin.  PARITY UNPINNED w.r.t. upstream binaries.
"""
import numpy as np

J_RX, J_RY, J_RZ, J_PX, J_PY, J_PZ, J_RU, J_PU = 1, 2, 3, 4, 5, 6, 7, 8
J_FREEFLYER, J_SPHERICAL, J_TRANSLATION = 9, 10, 11   # JointModelFreeFlyer / Spherical / Translation


def skew(t):
    return np.array([[0.0, -t[2], t[1]], [t[2], 0.0, -t[0]], [-t[1], t[0], 0.0]])


def quat_matrix(x, y, z, w):
    """Eigen::Quaternion::toRotationMatrix"""
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def joint_transform(jtype, axis, q):
    """JointModel::calc -> (R, t) of M(q); q = the joint's own segment of the configuration"""
    q = np.atleast_1d(np.asarray(q, dtype=float))
    if jtype == J_FREEFLYER:
        return quat_matrix(*q[3:7]), q[:3].copy()
    if jtype == J_SPHERICAL:
        return quat_matrix(*q[:4]), np.zeros(3)
    if jtype == J_TRANSLATION:
        return np.eye(3), q[:3].copy()
    q = q[0]
    R = np.eye(3)
    t = np.zeros(3)
    c, s = np.cos(q), np.sin(q)
    if jtype == J_RX:
        R = np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=float)
    elif jtype == J_RY:
        R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=float)
    elif jtype == J_RZ:
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=float)
    elif jtype == J_RU:
        a = np.asarray(axis, dtype=float)
        R = c * np.eye(3) + (1.0 - c) * np.outer(a, a) + s * skew(a)
    elif jtype in (J_PX, J_PY, J_PZ):
        t[jtype - J_PX] = q
    elif jtype == J_PU:
        t = np.asarray(axis, dtype=float) * q
    return R, t


def joint_S(jtype, axis):
    """motion subspace, 6 x nv_i"""
    if jtype == J_FREEFLYER:
        return np.eye(6)
    if jtype == J_SPHERICAL:
        return np.eye(6)[:, 3:]
    if jtype == J_TRANSLATION:
        return np.eye(6)[:, :3]
    return joint_S1(jtype, axis).reshape(6, 1)


def joint_S1(jtype, axis):
    S = np.zeros(6)
    if jtype in (J_PX, J_PY, J_PZ):
        S[jtype - J_PX] = 1.0
    elif jtype in (J_RX, J_RY, J_RZ):
        S[3 + jtype - J_RX] = 1.0
    elif jtype == J_PU:
        S[:3] = axis
    elif jtype == J_RU:
        S[3:] = axis
    return S


def action_matrix(R, t):
    """SE3::toActionMatrix: [[R, [t]x R],[0, R]]"""
    X = np.zeros((6, 6))
    X[:3, :3] = R
    X[:3, 3:] = skew(t) @ R
    X[3:, 3:] = R
    return X


def dual_action_matrix(R, t):
    """SE3::toDualActionMatrix: [[R, 0],[[t]x R, R]]"""
    X = np.zeros((6, 6))
    X[:3, :3] = R
    X[3:, :3] = skew(t) @ R
    X[3:, 3:] = R
    return X


def action_matrix_inverse(R, t):
    """SE3::toActionMatrixInverse: [[R^T, -R^T [t]x],[0, R^T]]"""
    X = np.zeros((6, 6))
    X[:3, :3] = R.T
    X[:3, 3:] = -R.T @ skew(t)
    X[3:, 3:] = R.T
    return X


class DenseSolver:
    """Plain solver of the reference with dense-QP residual / feasibility logic."""

    def __init__(self, model, max_iter=200, tol_abs=1e-3, tol_rel=1e-3, tol_primal_inf=1e-2, tol_dual_inf=1e-2,
                 rho=1e-5, mu=1e-2, mu_equality_scale_factor=1e4, mu_update_strat=0, num_eq_c=1, eq_c_dim=6,
                 warm_start=False, tol_tail_solve=1e-1):
        if eq_c_dim != 6:
            raise RuntimeError("equality constraint dimension is not 6")
        m = self.model = model
        self.nj, self.nb, self.nv = m.njoints, m.njoints - 1, m.nv
        self.max_iter, self.tol_abs, self.tol_rel = max_iter, tol_abs, tol_rel
        self.tol_primal_inf, self.tol_dual_inf = tol_primal_inf, tol_dual_inf
        self.rho, self.mu0, self.mu, self.scale = rho, mu, mu, mu_equality_scale_factor
        self.mu_update_strat = mu_update_strat
        self.nc, self.m = num_eq_c, eq_c_dim
        self.warm_start, self.tol_tail_solve = warm_start, tol_tail_solve
        nj, nv = self.nj, self.nv
        self.S = [joint_S(int(m.jtype[i]), m.axis[i]) for i in range(nj)]      # 6 x nv_i
        self.nvs = [0] + [self.S[i].shape[1] for i in range(1, nj)]
        self.nqs = [0] + [{J_FREEFLYER: 7, J_SPHERICAL: 4, J_TRANSLATION: 3}.get(int(m.jtype[i]), 1) for i in range(1, nj)]
        self.nu = np.zeros(nv); self.w = np.zeros(nv); self.z = np.zeros(nv)
        self.vis = np.zeros((nj, 6)); self.vis_prev = np.zeros((nj, 6)); self.fis = np.zeros((nj, 6))
        self.yis = np.zeros((nj, 6))
        self.His = np.zeros((nj, 6, 6)); self.pis = np.zeros((nj, 6))
        self.Ris = [np.zeros(n) for n in self.nvs]; self.ris = [np.zeros(n) for n in self.nvs]
        self.Di_invs = [np.zeros((n, n)) for n in self.nvs]
        self.Pis = np.zeros((nj, 6, 6))
        self.liMi = [(np.eye(3), np.zeros(3)) for _ in range(nj)]
        self.oMi = [(np.eye(3), np.zeros(3)) for _ in range(nj)]
        self.primal_residual = self.dual_residual = np.inf
        self.tol_primal = self.tol_dual = 0.0
        self.ResetSolver()

    # ---- loik-loid.hpp:150-183 -------------------------------------------------------------------
    def ResetSolver(self):
        nb, nv, m = self.nb, self.nv, self.m
        self.qp_constraint_dim = 6 * nb + m * nb + nv
        self.qp_var_dim = 6 * nb + nv
        cd, vd = self.qp_constraint_dim, self.qp_var_dim
        # problem_.Reset(), ik-id-description.hpp:372-403
        self.A_qp = np.zeros((cd, vd)); self.P_qp = np.zeros((vd, vd)); self.q_qp = np.zeros(vd)
        self.x_qp = np.zeros(vd); self.z_qp = np.zeros(cd); self.y_qp = np.zeros(cd)
        self.lb_qp = np.zeros(cd); self.ub_qp = np.zeros(cd)
        self.x_qp_prev = np.zeros(vd); self.z_qp_prev = np.zeros(cd); self.y_qp_prev = np.zeros(cd)
        self.delta_x_qp = np.zeros(vd); self.delta_z_qp = np.zeros(cd); self.delta_y_qp = np.zeros(cd)
        self.delta_y_qp_plus = np.zeros(cd); self.delta_y_qp_minus = np.zeros(cd)
        # Base::Reset(), task-solver-base.hpp:73-84
        self.iter = 0; self.converged = False; self.primal_infeasible = False; self.dual_infeasible = False
        self.mu = self.mu0
        # ik_id_data_.Reset(warm_start), loik-loid-data.hxx:99-150
        if not self.warm_start:
            self.nu[:] = 0; self.w[:] = 0; self.z[:] = 0
            self.vis[:] = 0; self.vis_prev[:] = 0; self.fis[:] = 0; self.yis[:] = 0
        else:
            self.fis[:] = 0
            self.vis_prev[:] = self.vis
        self.His[:] = 0; self.pis[:] = 0
        self.tail_solve_iter = 0
        self.primal_residual_vec = np.zeros(self.m * nb + nv)
        self.dual_residual_vec = np.zeros(6 * nb + nv)
        self.mu_eq = self.scale * self.mu
        self.mu_ineq = self.mu

    # ---- loik-loid.hxx:16-33 ---------------------------------------------------------------------
    def FwdPassInit(self, q):
        m = self.model
        for idx in range(1, self.nj):
            iq = int(m.idx_q[idx])
            Rj, tj = joint_transform(int(m.jtype[idx]), m.axis[idx], q[iq:iq + self.nqs[idx]])
            P = np.asarray(m.placement[idx], dtype=float)
            Rp, tp = P[:9].reshape(3, 3), P[9:]
            R, t = Rp @ Rj, tp + Rp @ tj
            self.liMi[idx] = (R, t)
            Ro, to = self.oMi[int(m.parents[idx])]
            self.oMi[idx] = (Ro @ R, to + Ro @ t)

    # ---- ik-id-description.hpp:411-491 -----------------------------------------------------------
    def UpdateQPADMMSolveInit(self, H_ref, v_ref, c_ids, Ais, bis, lb, ub):
        nb, nv, m6 = self.nb, self.nv, self.m
        H_ref = np.asarray(H_ref, dtype=float).reshape(6, 6); v_ref = np.asarray(v_ref, dtype=float).reshape(6)
        Ais = np.asarray(Ais, dtype=float).reshape(-1, 6, 6); bis = np.asarray(bis, dtype=float).reshape(-1, 6)
        if len(c_ids) != self.nc or Ais.shape[0] != self.nc or bis.shape[0] != self.nc:
            raise RuntimeError("number of equality constraints doesn't match initialization")
        if len(lb) != nv or len(ub) != nv:
            raise RuntimeError("inequality constraint dimension has changed")
        self.H_refs = [H_ref.copy() for _ in range(self.nj)]; self.v_refs = [v_ref.copy() for _ in range(self.nj)]
        self.c_ids = [int(c) for c in c_ids]; self.Ais = Ais; self.bis = bis
        self.lb = np.asarray(lb, dtype=float).copy(); self.ub = np.asarray(ub, dtype=float).copy()
        off = 6 * nb + m6 * nb
        self.lb_qp[off:off + nv] = self.lb
        self.ub_qp[off:off + nv] = self.ub
        self.A_qp[:6 * nb, :6 * nb] = -np.eye(6 * nb)
        self.A_qp[off:off + nv, 6 * nb:6 * nb + nv] = np.eye(nv)
        mdl = self.model
        for idx in range(1, self.nj):
            r0 = (idx - 1) * 6
            self.P_qp[r0:r0 + 6, r0:r0 + 6] = H_ref
            self.q_qp[r0:r0 + 6] = -H_ref.T @ v_ref   # (rewritten per link by UpdateReferences)
            iv = int(mdl.idx_v[idx])
            self.A_qp[r0:r0 + 6, 6 * nb + iv:6 * nb + iv + self.nvs[idx]] = self.S[idx]
            parent = int(mdl.parents[idx])
            if parent > 0:
                cp = (parent - 1) * 6
                Ri, ti = self.oMi[idx]
                Rp, tp = self.oMi[parent]
                # iMo.toActionMatrix() * oMp.toActionMatrix()
                self.A_qp[r0:r0 + 6, cp:cp + 6] = action_matrix(Ri.T, -Ri.T @ ti) @ action_matrix(Rp, tp)
            self.A_qp[r0:r0 + 6, r0:r0 + 6] = -np.eye(6)
        for c, c_idx in enumerate(self.c_ids):
            r0 = 6 * nb + (c_idx - 1) * m6
            c0 = (c_idx - 1) * 6
            self.A_qp[r0:r0 + m6, c0:c0 + 6] = Ais[c]
            self.lb_qp[r0:r0 + m6] = bis[c]
            self.ub_qp[r0:r0 + m6] = bis[c]
        self.z_qp[6 * nb:6 * nb + m6 * nb] = self.ub_qp[6 * nb:6 * nb + m6 * nb]

    # ---- per-link weights and targets: ik-id-description.hpp's UpdateReferences (same text as the optimized class',
    #      ik-id-description-optimized.hpp:103-121) + the P_qp / q_qp blocks UpdateQPADMMSolveInit fills from them
    def UpdateReferences(self, H_refs, v_refs):
        H_refs = np.asarray(H_refs, dtype=float).reshape(-1, 6, 6); v_refs = np.asarray(v_refs, dtype=float).reshape(-1, 6)
        if H_refs.shape[0] != self.nj or v_refs.shape[0] != self.nj:
            raise RuntimeError("input arguments 'H_refs', 'v_refs' have wrong size")
        self.H_refs = [H.copy() for H in H_refs]; self.v_refs = [v.copy() for v in v_refs]
        for idx in range(1, self.nj):
            r0 = (idx - 1) * 6
            self.P_qp[r0:r0 + 6, r0:r0 + 6] = self.H_refs[idx]
            self.q_qp[r0:r0 + 6] = -self.H_refs[idx].T @ self.v_refs[idx]

    # ---- loik-loid.hpp:362-377 -------------------------------------------------------------------
    def SolveInit(self, q, H_ref, v_ref, c_ids, Ais, bis, lb, ub):
        self.ResetSolver()
        self.FwdPassInit(np.asarray(q, dtype=float))
        self.UpdateQPADMMSolveInit(H_ref, v_ref, c_ids, Ais, bis, lb, ub)

    # ---- loik-loid.hxx:39-76 ---------------------------------------------------------------------
    def FwdPass1(self):
        for idx in range(1, self.nj):
            iv, n = int(self.model.idx_v[idx]), self.nvs[idx]
            self.Ris[idx] = self.mu_ineq * np.ones(n)
            self.ris[idx] = self.w[iv:iv + n] - self.mu_ineq * self.z[iv:iv + n]
            self.His[idx] = self.rho * np.eye(6) + self.H_refs[idx]
            self.pis[idx] = -self.rho * self.vis_prev[idx] - self.H_refs[idx].T @ self.v_refs[idx]
        for c, c_id in enumerate(self.c_ids):
            Ai, bi = self.Ais[c], self.bis[c]
            self.His[c_id] += self.mu_eq * Ai.T @ Ai
            self.pis[c_id] += Ai.T @ self.yis[c_id] - self.mu_eq * Ai.T @ bi

    # ---- loik-loid.hxx:82-113 --------------------------------------------------------------------
    def BwdPass(self):
        for idx in range(self.nj - 1, 0, -1):
            parent = int(self.model.parents[idx])
            R, t = self.liMi[idx]
            Hi, pi, Si = self.His[idx], self.pis[idx], self.S[idx]
            Di = np.diag(self.Ris[idx]) + Si.T @ Hi @ Si
            Dinv = np.linalg.inv(Di)
            self.Di_invs[idx] = Dinv
            Pi = np.eye(6) - Hi @ Si @ Dinv @ Si.T
            self.Pis[idx] = Pi
            Xd = dual_action_matrix(R, t)
            self.His[parent] = self.His[parent] + Xd @ (Pi @ Hi) @ action_matrix_inverse(R, t)
            self.pis[parent] = self.pis[parent] + Xd @ (Pi @ pi - Hi @ Si @ Dinv @ self.ris[idx])

    # ---- loik-loid.hxx:120-151 -------------------------------------------------------------------
    def FwdPass2(self):
        for idx in range(1, self.nj):
            parent = int(self.model.parents[idx])
            iv, n = int(self.model.idx_v[idx]), self.nvs[idx]
            R, t = self.liMi[idx]
            Hi, pi, Si = self.His[idx], self.pis[idx], self.S[idx]
            vi_parent = action_matrix_inverse(R, t) @ self.vis[parent]
            self.nu[iv:iv + n] = -self.Di_invs[idx] @ (Si.T @ (Hi @ vi_parent + pi) + self.ris[idx])
            self.vis[idx] = vi_parent + Si @ self.nu[iv:iv + n]
            self.fis[idx] = Hi @ self.vis[idx] + pi

    # ---- loik-loid.hxx:158-164 -------------------------------------------------------------------
    def BoxProj(self):
        self.z = np.minimum(self.ub, np.maximum(self.lb, self.nu + (1.0 / self.mu_ineq) * self.w))

    # ---- loik-loid.hxx:171-189 -------------------------------------------------------------------
    def DualUpdate(self):
        for c, c_id in enumerate(self.c_ids):
            self.yis[c_id] = self.yis[c_id] + self.mu_eq * (self.Ais[c] @ self.vis[c_id] - self.bis[c])
        self.w = self.w + self.mu_ineq * (self.nu - self.z)

    # ---- ik-id-description.hpp:499-539 -----------------------------------------------------------
    def UpdateQPADMMSolveLoopUtility(self):
        nb, nv, m6 = self.nb, self.nv, self.m
        self.x_qp_prev = self.x_qp.copy(); self.z_qp_prev = self.z_qp.copy(); self.y_qp_prev = self.y_qp.copy()
        for idx in range(1, self.nj):
            r = (idx - 1) * 6
            self.x_qp[r:r + 6] = self.vis[idx]
            self.y_qp[r:r + 6] = self.fis[idx]
            ry = 6 * nb + (idx - 1) * m6
            self.y_qp[ry:ry + m6] = self.yis[idx]
        self.x_qp[6 * nb:6 * nb + nv] = self.nu
        self.y_qp[6 * nb + nb * m6:6 * nb + nb * m6 + nv] = self.w
        self.z_qp[6 * nb + nb * m6:6 * nb + nb * m6 + nv] = self.z
        self.delta_x_qp = self.x_qp - self.x_qp_prev
        self.delta_y_qp = self.y_qp - self.y_qp_prev
        self.delta_z_qp = self.z_qp - self.z_qp_prev
        self.delta_y_qp_plus = np.maximum(self.delta_y_qp, 0)
        self.delta_y_qp_minus = np.minimum(self.delta_y_qp, 0)

    # ---- loik-loid.hxx:207-295 -------------------------------------------------------------------
    def ComputeResiduals(self):
        nb, nv, m6 = self.nb, self.nv, self.m
        for c, c_id in enumerate(self.c_ids):
            self.primal_residual_vec[m6 * (c_id - 1):m6 * c_id] = self.Ais[c] @ self.vis[c_id] - self.bis[c]
        self.primal_residual_vec[m6 * nb:] = self.nu - self.z
        self.primal_residual = np.max(np.abs(self.primal_residual_vec))
        self.primal_residual_task = np.max(np.abs(self.primal_residual_vec[:m6 * nb]))
        self.primal_residual_slack = np.max(np.abs(self.primal_residual_vec[m6 * nb:]))
        self.dual_residual_vec = self.P_qp @ self.x_qp + self.q_qp + self.A_qp.T @ self.y_qp
        self.dual_residual = np.max(np.abs(self.dual_residual_vec))
        self.dual_residual_v = np.max(np.abs(self.dual_residual_vec[:6 * nb]))
        self.dual_residual_nu = np.max(np.abs(self.dual_residual_vec[6 * nb:]))

    # ---- loik-loid.hxx:302-324 -------------------------------------------------------------------
    def CheckConvergence(self):
        ninf = lambda x: np.max(np.abs(x))
        self.tol_primal = self.tol_abs + self.tol_rel * max(ninf(self.A_qp @ self.x_qp), ninf(self.z_qp))
        self.tol_dual = self.tol_abs + self.tol_rel * max(max(ninf(self.P_qp @ self.x_qp),
                                                              ninf(self.A_qp.T @ self.y_qp)), ninf(self.q_qp))
        if self.primal_residual < self.tol_primal and self.dual_residual < self.tol_dual:
            self.converged = True

    # ---- loik-loid.hxx:331-367 -------------------------------------------------------------------
    def get_primal_infeasibility_cond_1(self):
        return bool(np.max(np.abs(self.A_qp.T @ self.delta_y_qp)) <= self.tol_primal_inf * np.max(np.abs(self.delta_y_qp)))

    def get_primal_infeasibility_cond_2(self):
        return bool((self.ub_qp @ self.delta_y_qp_plus + self.lb_qp @ self.delta_y_qp_minus)
                    <= self.tol_primal_inf * np.max(np.abs(self.delta_y_qp)))

    def CheckFeasibility(self):
        if self.get_primal_infeasibility_cond_1() and self.get_primal_infeasibility_cond_2():
            self.primal_infeasible = True
        dx = np.max(np.abs(self.delta_x_qp))
        c1 = np.max(np.abs(self.P_qp @ self.delta_x_qp)) <= self.tol_dual_inf * dx
        c2 = (self.q_qp @ self.delta_x_qp) <= self.tol_dual_inf * dx
        if c1 and c2:
            Adx = self.A_qp @ self.delta_x_qp
            if np.all(Adx >= -self.tol_dual_inf * dx) and np.all(Adx <= self.tol_dual_inf * dx):
                self.dual_infeasible = True

    # ---- loik-loid.hxx:374-402 -------------------------------------------------------------------
    def UpdateMu(self):
        if self.mu_update_strat != 0:
            raise RuntimeError("mu update strategy not supported")
        if self.primal_residual > 10 * self.dual_residual:
            self.mu *= 10
        elif self.dual_residual > 10 * self.primal_residual:
            self.mu *= 0.1
        else:
            return
        self.mu_eq = self.scale * self.mu
        self.mu_ineq = self.mu

    def _body(self):
        self.vis_prev = self.vis.copy()  # UpdatePrev
        self.FwdPass1(); self.BwdPass(); self.FwdPass2(); self.BoxProj(); self.DualUpdate()
        self.UpdateQPADMMSolveLoopUtility()
        self.ComputeResiduals()

    # ---- loik-loid.hpp:251-347 -------------------------------------------------------------------
    def InfeasibilityTailSolve(self):
        self.tail_solve_iter = 0
        while (np.max(np.abs(self.delta_x_qp)) >= self.tol_tail_solve
               or np.max(np.abs(self.delta_z_qp)) >= self.tol_tail_solve):
            if self.iter >= self.max_iter:
                return
            self.iter += 1
            self.tail_solve_iter += 1
            self._body()

    # ---- loik-loid.hpp:383-460 / :476-... --------------------------------------------------------
    def Solve(self, *a):
        if len(a) == 8:
            self.SolveInit(*a)
        elif len(a) != 0:
            raise TypeError("Solve() takes 0 or 8 arguments")
        for i in range(1, self.max_iter):
            self.iter = i
            self._body()
            self.CheckConvergence()
            if self.iter > 1:
                self.CheckFeasibility()
            if self.converged:
                break
            elif self.primal_infeasible or self.dual_infeasible:
                self.InfeasibilityTailSolve()
                break
            self.UpdateMu()

    # ---- getters used by the reference's tests (loik-loid.hpp:594-614) ------------------------------
    def get_iter(self): return self.iter
    def get_delta_x_qp_inf_norm(self): return np.max(np.abs(self.delta_x_qp))
    def get_delta_z_qp_inf_norm(self): return np.max(np.abs(self.delta_z_qp))
    def get_delta_y_qp_inf_norm(self): return np.max(np.abs(self.delta_y_qp))
    def get_A_qp_T_delta_y_qp_inf_norm(self): return np.max(np.abs(self.A_qp.T @ self.delta_y_qp))
    def get_ub_qp_T_delta_y_qp_plus(self): return float(self.ub_qp @ self.delta_y_qp_plus)
    def get_lb_qp_T_delta_y_qp_minus(self): return float(self.lb_qp @ self.delta_y_qp_minus)
