"""numpy restatement of what the flat engine computes (loik_amd/csrc/loik_flat.hpp) -- TEST INFRASTRUCTURE, like oracle/.

The two recursions of an ADMM iteration of the reference (FwdPass1 + BwdPass leaf -> root, FwdPass2 root -> leaf;
/root/reference/include/loik/loik-loid-optimized.hxx:290-338, :31-81, :102-163) are the sparse LDL^T solve of
(J^T H J + mu I) nu = -(J^T p^base + w - mu z) in tree order.  With everything expressed at the world origin:

    tau_a  = (w_a - mu z_a) + S^w_a . sum_{d in subtree(a)} p^base,w_d          (subtree sum of 6-vectors)
    r'     = W tau,          W = (I + L)^-1,   L_{a,d} = S^w_a . UDinv^w_d   (d a descendant of a)
    nu     = -W^T (Dinv r')
    v^w_i  = sum_{a in ancestors*(i)} S^w_a nu_a                                (path sum of 6-vectors)
    f^w_i  = sum_{d in subtree(i)} phi^w_d ,  phi_d = H^base_d v_d + p^base_d   (force balance; subtree sum)
    g_i    = A^T y_i - phi_i                                                     (BwdPass2 in closed form)

tests/test_flat_arithmetic.py checks that this reorganised arithmetic reproduces the CPU oracle's iteration counts, flags and
answers -- the claim the device engine rests on -- without a GPU; scripts/r03/flat_proto.py runs it on large batches."""
import numpy as np


def skew(t):
    B = t.shape[0]
    T = np.zeros((B, 3, 3))
    T[:, 0, 1] = -t[:, 2]; T[:, 0, 2] = t[:, 1]
    T[:, 1, 0] = t[:, 2]; T[:, 1, 2] = -t[:, 0]
    T[:, 2, 0] = -t[:, 1]; T[:, 2, 1] = t[:, 0]
    return T


def rot_axis(a, q):
    c, s = np.cos(q), np.sin(q)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3)[None] * c[:, None, None] + (1 - c)[:, None, None] * np.outer(a, a)[None] + s[:, None, None] * K[None]


def xstar(R, t):
    """force transform child -> parent as a 6x6: [[R, 0], [T R, R]]"""
    B = R.shape[0]
    X = np.zeros((B, 6, 6))
    X[:, :3, :3] = R
    X[:, 3:, 3:] = R
    X[:, 3:, :3] = skew(t) @ R
    return X


class Flat:
    def __init__(self, wl):
        m = wl["model"]
        self.m = m
        prm = wl["params"]
        self.prm = prm
        n = m.njoints - 1
        self.n = n
        q = wl["q"]
        B = q.shape[0]
        self.B = B
        par = np.asarray(m.parents)
        self.par = par
        # ---- kinematics: liMi, oMi, S (local), S^w
        S = np.zeros((n + 1, 6))
        Rl = np.zeros((B, n + 1, 3, 3)); tl = np.zeros((B, n + 1, 3))
        R0 = np.zeros((B, n + 1, 3, 3)); t0 = np.zeros((B, n + 1, 3))
        R0[:, 0] = np.eye(3)
        for i in range(1, n + 1):
            jt = int(m.jtype[i])
            P = m.placement[i]; Rp, tp = P[:9].reshape(3, 3), P[9:]
            qi = q[:, int(m.idx_q[i])]
            if jt in (1, 2, 3, 7):
                a = np.zeros(3)
                if jt == 7: a = np.asarray(m.axis[i], float)
                else: a[jt - 1] = 1
                Rl[:, i] = Rp[None] @ rot_axis(a, qi); tl[:, i] = tp[None]
                S[i, 3:] = a
            else:
                a = np.zeros(3)
                if jt == 8: a = np.asarray(m.axis[i], float)
                else: a[jt - 4] = 1
                Rl[:, i] = Rp[None]; tl[:, i] = tp[None] + (Rp @ a)[None] * qi[:, None]
                S[i, :3] = a
            p = par[i]
            R0[:, i] = R0[:, p] @ Rl[:, i]
            t0[:, i] = t0[:, p] + np.einsum("bij,bj->bi", R0[:, p], tl[:, i])
        self.S, self.Rl, self.tl, self.R0, self.t0 = S, Rl, tl, R0, t0
        # S^w = motion local -> world
        Sw = np.zeros((B, n + 1, 6))
        for i in range(1, n + 1):
            ang = np.einsum("bij,j->bi", R0[:, i], S[i, 3:]); lin = np.einsum("bij,j->bi", R0[:, i], S[i, :3])
            Sw[:, i, 3:] = ang; Sw[:, i, :3] = lin + np.cross(t0[:, i], ang)
        self.Sw = Sw
        # tree masks
        anc = np.zeros((n + 1, n + 1), bool)  # anc[i, a]: a is an ancestor-or-self of i (a >= 1)
        for i in range(1, n + 1):
            a = i
            while a > 0:
                anc[i, a] = True
                a = par[a]
        self.anc = anc                      # path sums:   x_i = sum_a anc[i,a] y_a
        self.sub = anc.T.copy()             # subtree sums: X_a = sum_d sub[a,d] y_d
        self.depth = anc.sum(1)
        # problem
        self.Href = np.asarray(wl["H_ref"], float)
        self.Hv = self.Href @ np.asarray(wl["v_ref"], float)
        self.c_ids = [int(c) for c in wl["c_ids"]]
        A = np.asarray(wl["Ais"], float)
        self.A = np.broadcast_to(A, (B,) + A.shape[-3:]) if A.ndim == 3 else A
        self.b = np.asarray(wl["bis"], float)
        self.lb = np.broadcast_to(np.asarray(wl["lb"], float), (B, n)); self.ub = np.broadcast_to(np.asarray(wl["ub"], float), (B, n))
        self.AtA = np.einsum("bcki,bckj->bcij", self.A, self.A)
        self.Atb = np.einsum("bcki,bck->bci", self.A, self.b)
        self.bnorm = np.abs(self.b).reshape(B, -1).max(1)
        self.Hv_inf = np.abs(self.Hv).max()
        self.factors = {}

    # ---- world <-> local
    def f_to_world(self, f):  # [B, n+1, 6] forces
        lin = np.einsum("bnij,bnj->bni", self.R0, f[..., :3]); ang = np.einsum("bnij,bnj->bni", self.R0, f[..., 3:])
        return np.concatenate([lin, ang + np.cross(self.t0, lin)], -1)

    def f_to_local(self, F):
        lin = np.einsum("bnji,bnj->bni", self.R0, F[..., :3])
        ang = np.einsum("bnji,bnj->bni", self.R0, F[..., 3:] - np.cross(self.t0, F[..., :3]))
        return np.concatenate([lin, ang], -1)

    def m_to_local(self, V):
        lin = np.einsum("bnji,bnj->bni", self.R0, V[..., :3] - np.cross(self.t0, V[..., 3:]))
        ang = np.einsum("bnji,bnj->bni", self.R0, V[..., 3:])
        return np.concatenate([lin, ang], -1)

    # ---- H recursion for mu = mu0 * 10^k for the instances idx: UD^w, dinv, W
    def factor(self, kexp, idx):
        n, par = self.n, self.par
        prm = self.prm
        mu = prm["mu"]
        for _ in range(abs(kexp)):
            mu = mu * 10 if kexp > 0 else mu * 0.1
        mu_eq = prm["mu_equality_scale_factor"] * mu
        nb = idx.size
        H = np.zeros((nb, n + 1, 6, 6))
        H[:, 1:] = prm["rho"] * np.eye(6) + self.Href
        for c, cid in enumerate(self.c_ids):
            H[:, cid] += mu_eq * self.AtA[idx, c]
        UD = np.zeros((nb, n + 1, 6)); dinv = np.zeros((nb, n + 1))
        for i in range(n, 0, -1):
            U = np.einsum("bij,j->bi", H[:, i], self.S[i])
            d = 1.0 / (U @ self.S[i] + mu)
            dinv[:, i] = d
            UD[:, i] = U * d[:, None]
            p = par[i]
            if p > 0:
                Hp = H[:, i] - UD[:, i, :, None] * U[:, None, :]
                X = xstar(self.Rl[idx, i], self.tl[idx, i])
                H[:, p] += X @ Hp @ X.transpose(0, 2, 1)
        # L_{a,d} = S^w_a . UD^w_d (a strict ancestor of d); W = (I + L)^-1 column by column, nearest ancestor first
        lin = np.einsum("bnij,bnj->bni", self.R0[idx], UD[..., :3]); ang = np.einsum("bnij,bnj->bni", self.R0[idx], UD[..., 3:])
        UDw = np.concatenate([lin, ang + np.cross(self.t0[idx], lin)], -1)
        L = np.einsum("bak,bdk->bad", self.Sw[idx], UDw)  # [b, a, d]
        W = np.zeros((nb, n + 1, n + 1))
        for d in range(1, n + 1):
            W[:, d, d] = 1.0
            path = []
            a = par[d]
            while a > 0:
                acc = L[:, a, d].copy()
                for e in path:
                    acc += L[:, a, e] * W[:, e, d]
                W[:, a, d] = -acc
                path.append(a)
                a = par[a]
        return dinv, W

    def get_factor(self, kexp_arr, act):
        """dinv [B, n+1], W [B, n+1, n+1] of every active instance's current decade (cached per decade)"""
        for k in np.unique(kexp_arr[act]):
            need = act & (kexp_arr == k) & (self.fk != k)
            if need.any():
                idx = np.nonzero(need)[0]
                d, W = self.factor(int(k), idx)
                self.dinv[idx] = d; self.W[idx] = W; self.fk[idx] = k

    def solve(self):
        prm, n, B = self.prm, self.n, self.B
        rho, mu0, scale = prm["rho"], prm["mu"], prm["mu_equality_scale_factor"]
        max_iter, tol_abs, tol_rel = prm["max_iter"], prm["tol_abs"], prm["tol_rel"]
        tol_pinf, tol_tail = prm["tol_primal_inf"], prm["tol_tail_solve"]
        nc = len(self.c_ids)
        self.dinv = np.zeros((B, n + 1)); self.W = np.zeros((B, n + 1, n + 1)); self.fk = np.full(B, -99)
        v = np.zeros((B, n + 1, 6)); f = np.zeros((B, n + 1, 6)); g = np.zeros((B, n + 1, 6))
        w = np.zeros((B, n + 1)); z = np.zeros((B, n + 1)); nu = np.zeros((B, n + 1)); s = np.zeros((B, n + 1))
        y = np.zeros((B, nc, 6)); Aty = np.zeros((B, nc, 6))
        mu = np.full(B, mu0); kexp = np.zeros(B, int)
        it = np.zeros(B, int); done = np.zeros(B, bool); conv = np.zeros(B, bool); pinf = np.zeros(B, bool)
        tail = np.zeros(B, bool)
        lb = np.concatenate([np.zeros((B, 1)), self.lb], 1); ub = np.concatenate([np.zeros((B, 1)), self.ub], 1)
        Sw, S = self.Sw, self.S
        anc, sub = self.anc.astype(float), self.sub.astype(float)
        res = np.zeros((B, 2))
        while True:
            act = ~done
            # loop bound (hpp:377): an instance not in tail mode stops once iter + 1 >= max_iter
            stop = act & ~tail & (it + 1 >= max_iter)
            done |= stop; act &= ~stop
            if not act.any():
                break
            self.get_factor(kexp, act)
            A = np.nonzero(act)[0]
            mu_a = mu[A]; mu_eq = scale * mu_a
            va, wa, za = v[A], w[A], z[A]
            # p^base
            pb = -rho * va - self.Hv
            pb[:, 0] = 0
            for c, cid in enumerate(self.c_ids):
                pb[:, cid] += Aty[A, c] - mu_eq[:, None] * self.Atb[A, c]
            sel = type("x", (), {})()
            sel.R0, sel.t0 = self.R0[A], self.t0[A]
            lin = np.einsum("bnij,bnj->bni", sel.R0, pb[..., :3]); ang = np.einsum("bnij,bnj->bni", sel.R0, pb[..., 3:])
            pbw = np.concatenate([lin, ang + np.cross(sel.t0, lin)], -1)
            PB = np.einsum("ad,bdk->bak", sub, pbw)
            tau = (wa - mu_a[:, None] * za) + np.einsum("bak,bak->ba", Sw[A], PB)
            rp = np.einsum("bad,bd->ba", self.W[A], tau)
            nut = self.dinv[A] * rp
            nun = -np.einsum("bad,ba->bd", self.W[A], nut)
            nun[:, 0] = 0
            vw = np.einsum("ia,bak->bik", anc, Sw[A] * nun[..., None])
            lin = np.einsum("bnji,bnj->bni", sel.R0, vw[..., :3] - np.cross(sel.t0, vw[..., 3:]))
            ang = np.einsum("bnji,bnj->bni", sel.R0, vw[..., 3:])
            vn = np.concatenate([lin, ang], -1)
            vn[:, 0] = 0
            dv = vn - va
            l_dvis = np.abs(dv[:, 1:]).max((1, 2)); l_nu = np.abs(nun).max(1); l_dnu = np.abs(nun - nu[A]).max(1)
            hrv = vn @ self.Href.T
            l_hrefv = np.abs(hrv[:, 1:]).max((1, 2))
            x = nun + (1.0 / mu_a)[:, None] * wa
            zn = np.minimum(ub[A], np.maximum(lb[A], x))
            l_dz = np.abs(zn - za).max(1); l_prs = np.abs(nun - zn).max(1)
            dw = mu_a[:, None] * (nun - zn)
            l_dw = np.abs(dw).max(1)
            up = (ub[A] * np.maximum(dw, 0)).sum(1); lm = (lb[A] * np.minimum(dw, 0)).sum(1)
            wn = wa + dw
            l_dy = np.zeros(A.size); l_prt = np.zeros(A.size); l_av = np.zeros(A.size)
            phi = rho * dv + hrv - self.Hv
            phi[:, 0] = 0
            gn = -phi.copy()
            for c, cid in enumerate(self.c_ids):
                Av = np.einsum("bij,bj->bi", self.A[A, c], vn[:, cid])
                e = Av - self.b[A, c]
                dy = mu_eq[:, None] * e
                y[A, c] += dy
                Aty[A, c] = np.einsum("bji,bj->bi", self.A[A, c], y[A, c])
                l_dy = np.maximum(l_dy, np.abs(dy).max(1)); l_prt = np.maximum(l_prt, np.abs(e).max(1))
                l_av = np.maximum(l_av, np.abs(Av).max(1))
                up += (self.b[A, c] * np.maximum(dy, 0)).sum(1); lm += (self.b[A, c] * np.minimum(dy, 0)).sum(1)
                phi[:, cid] += Aty[A, c]
            lin = np.einsum("bnij,bnj->bni", sel.R0, phi[..., :3]); ang = np.einsum("bnij,bnj->bni", sel.R0, phi[..., 3:])
            phw = np.concatenate([lin, ang + np.cross(sel.t0, lin)], -1)
            Fw = np.einsum("ad,bdk->bak", sub, phw)
            lin = np.einsum("bnji,bnj->bni", sel.R0, Fw[..., :3])
            ang = np.einsum("bnji,bnj->bni", sel.R0, Fw[..., 3:] - np.cross(sel.t0, Fw[..., :3]))
            fn = np.concatenate([lin, ang], -1)
            fn[:, 0] = 0
            l_dfis = np.abs(fn - f[A])[:, 1:].max((1, 2))
            l_dg = np.abs(gn - g[A])[:, 1:].max((1, 2)); l_g = np.abs(gn[:, 1:]).max((1, 2))
            dvr = hrv - self.Hv + gn
            l_dualv = np.abs(dvr[:, 1:]).max((1, 2))
            sn = np.einsum("nk,bnk->bn", S, fn) + wn
            sn[:, 0] = 0
            l_stf = np.abs(sn).max(1); l_dstf = np.abs(sn - s[A]).max(1)
            v[A], f[A], g[A], w[A], z[A], nu[A], s[A] = vn, fn, gn, wn, zn, nun, sn
            # ---- epilogue (hpp:377-454, :271-319)
            it[A] += 1
            iter_ = it[A]
            primal = np.maximum(l_prt, l_prs); dual = np.maximum(l_dualv, l_stf)
            res[A, 0], res[A, 1] = primal, dual
            dx = np.maximum(l_dvis, l_dnu)
            tl_ = tail[A]
            tol_p = tol_abs + tol_rel * np.maximum(np.maximum(l_av, l_nu), self.bnorm[A])
            tol_d = tol_abs + tol_rel * np.maximum(np.maximum(l_hrefv, np.maximum(l_g, l_stf)), self.Hv_inf)
            cv = (primal < tol_p) & (dual < tol_d) & ~tl_
            dyqp = np.maximum(l_dfis, np.maximum(l_dy, l_dw)); atdy = np.maximum(l_dg, l_dstf)
            c1 = atdy <= tol_pinf * dyqp; c2 = (up + lm) <= tol_pinf * dyqp
            infe = c1 & c2 & (iter_ > 1) & ~tl_
            tail_stop = ~((dx >= tol_tail) | (l_dz >= tol_tail)) | (iter_ >= max_iter)
            # not in tail mode
            new_done = np.zeros(A.size, bool)
            new_done |= cv
            conv[A] |= cv
            pinf[A] |= infe
            enter_tail = infe & ~cv
            tail[A] |= enter_tail
            new_done |= enter_tail & tail_stop
            # mu update for the rest
            rest = ~tl_ & ~cv & ~infe
            upm = rest & (primal > 10 * dual); dnm = rest & ~upm & (dual > 10 * primal)
            mu[A[upm]] *= 10; kexp[A[upm]] += 1
            mu[A[dnm]] *= 0.1; kexp[A[dnm]] -= 1
            # already in tail mode
            new_done |= tl_ & tail_stop
            done[A[new_done]] = True
        return dict(z=z[:, 1:], iters=it, converged=conv, primal_infeasible=pinf, res=res, nu=nu[:, 1:])


