// loik_passes.hpp -- the reference's PASS-LEVEL public methods on the device, one pass per kernel launch.
//
// FirstOrderLoikOptimizedTpl exposes its passes (loik-loid-optimized.hpp:192-264: FwdPass1, BwdPassOptimizedVisitor,
// FwdPass2OptimizedVisitor, BoxProj, DualUpdate, ComputeResiduals, CheckConvergence, CheckFeasibility, UpdateMu) and its
// own component-wise test calls them one by one, reading the data object in between (tests/loik-loid.cpp:305-556).  The
// production kernels (k_lean / k_tail / k_solve) fuse these passes and never materialise the state in between, so the
// pass-level entry points of the C-ABI (loikb_pass) run on a SEPARATE, deliberately plain implementation: one instance per
// thread, the data object of the reference restated field by field (His AND His_aba, pis AND pis_aba, R, r, vis_prev ...:
// loik-loid-data-optimized.hpp:62-379) in an instance-major block of doubles, every pass a straightforward loop over the
// joints in the reference's order.  Nothing here is tuned; it is a debug path -- and, like the reference's plain solver
// next to its optimized one, a second implementation on the same GPU against which the fused engines are checked
// (tests/test_pass_level.py: N iterations composed of passes == Solve() of every engine).
// The state is kept per DEVICE joint: a multi-DoF joint of the caller's model is the chain of 1-DoF joints the engines use
// (JF_MASSLESS links carry no rho I + H_ref, no reference term and are left out of the norms over the links -- which reproduces
// the reference's nv x nv elimination); the getters select the body-carrying link.  The arithmetic is fp64 whatever the
// handle's precision (an fp32 handle's state is widened at load and rounded back by k_pass_store).
#pragma once

#include "loik_device.hpp"

namespace loikb {

// scalars of one instance in the pass state (doubles; integers stored exactly)
enum : int {
  PS_MU = 0, PS_MU_EQ, PS_MU_IN, PS_ITER, PS_CONVERGED, PS_PRIMAL_INF, PS_BIS_INF, PS_PRIMAL, PS_DUAL, PS_PR_TASK, PS_PR_SLACK,
  PS_DUAL_V, PS_DUAL_NU, PS_TOL_P, PS_TOL_D, PS_NU_INF, PS_DFIS_INF, PS_HREFV_INF, PS_DVIS_INF, PS_DNU_INF, PS_DZ_INF,
  PS_DYIS_INF, PS_AV_INF, PS_BTDY_PLUS, PS_BTDY_MINUS, PS_DW_INF, PS_G_INF, PS_DG_INF, PS_STF_INF, PS_DSTF_INF, PS_DX, PS_DYQP,
  PS_ATDY, PS_UBP, PS_LBM, PS_C1, PS_C2, PS_TAIL_IT, PS_COUNT = 40
};

struct PassLayout {
  int nj, nv, nc, B;
  // offsets (doubles) inside an instance's block
  int liMi, vis, vis_prev, fis, His, His_aba, pis, pis_aba, g, gnew, Href_v, UDinv, Dinv;     // per joint (0..nj-1)
  int R, r, nu, nu_prev, z, z_prev, w, delta_w, Stf, lb, ub;                                   // per DoF
  int yis, Aty, A, AtA, b, Atb;                                                               // per constraint
  int scal, stride;
};

inline PassLayout make_pass_layout(int nj, int nv, int nc, int B)
{
  PassLayout L{};
  L.nj = nj; L.nv = nv; L.nc = nc; L.B = B;
  int o = 0;
  auto take = [&](int n) { const int at = o; o += n; return at; };
  L.liMi = take(nj * 12); L.vis = take(nj * 6); L.vis_prev = take(nj * 6); L.fis = take(nj * 6);
  L.His = take(nj * 36); L.His_aba = take(nj * 36); L.pis = take(nj * 6); L.pis_aba = take(nj * 6);
  L.g = take(nj * 6); L.gnew = take(nj * 6); L.Href_v = take(nj * 6); L.UDinv = take(nj * 6); L.Dinv = take(nj);
  L.R = take(nv); L.r = take(nv); L.nu = take(nv); L.nu_prev = take(nv); L.z = take(nv); L.z_prev = take(nv);
  L.w = take(nv); L.delta_w = take(nv); L.Stf = take(nv); L.lb = take(nv); L.ub = take(nv);
  L.yis = take(nc * 6); L.Aty = take(nc * 6); L.A = take(nc * 36); L.AtA = take(nc * 36); L.b = take(nc * 6); L.Atb = take(nc * 6);
  L.scal = take(PS_COUNT);
  L.stride = o;
  return L;
}

struct PassParams {
  const double* href_tab;  // [nj][HREF_ROW]: (H_ref_i, H_ref_i v_ref_i) per joint (broadcast or UpdateReferences)
  double Hv_inf_norm;
  double rho, mu0, mu_scale, tol_abs, tol_rel, tol_primal_inf, tol_tail_solve;
  int max_iter, mu_osqp, a_shared, bnd_shared;
};

// ---- 6-D helpers on plain arrays (Pinocchio conventions) ---------------------------------------------
__device__ inline void p_act_force(const double* M, const double* f, double* o)  // (R f_l, R f_a + t x R f_l)
{
  const double* R = M; const double* t = M + 9;
  double l[3], a[3];
  for (int i = 0; i < 3; ++i) {
    l[i] = R[3 * i] * f[0] + R[3 * i + 1] * f[1] + R[3 * i + 2] * f[2];
    a[i] = R[3 * i] * f[3] + R[3 * i + 1] * f[4] + R[3 * i + 2] * f[5];
  }
  o[0] = l[0]; o[1] = l[1]; o[2] = l[2];
  o[3] = a[0] + t[1] * l[2] - t[2] * l[1];
  o[4] = a[1] + t[2] * l[0] - t[0] * l[2];
  o[5] = a[2] + t[0] * l[1] - t[1] * l[0];
}
__device__ inline void p_actinv_motion(const double* M, const double* v, double* o)  // (R^T (v_l - t x v_a), R^T v_a)
{
  const double* R = M; const double* t = M + 9;
  const double d[3] = {v[0] - (t[1] * v[5] - t[2] * v[4]), v[1] - (t[2] * v[3] - t[0] * v[5]), v[2] - (t[0] * v[4] - t[1] * v[3])};
  for (int i = 0; i < 3; ++i) {
    o[i] = R[i] * d[0] + R[3 + i] * d[1] + R[6 + i] * d[2];
    o[3 + i] = R[i] * v[3] + R[3 + i] * v[4] + R[6 + i] * v[5];
  }
}
// X* H X*^T with X* = [[R, 0], [T R, R]], T = [t]x (SE3actOn; the reference's version reads the blocks A, B, D of a symmetric H)
__device__ inline void p_congruence(const double* M, const double* H, double* out)
{
  const double* R = M; const double* t = M + 9;
  double X[36];
  const double T[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      X[6 * i + j] = R[3 * i + j];
      X[6 * i + 3 + j] = 0.0;
      double a = 0.0;
      for (int k = 0; k < 3; ++k) a += T[3 * i + k] * R[3 * k + j];
      X[6 * (3 + i) + j] = a;
      X[6 * (3 + i) + 3 + j] = R[3 * i + j];
    }
  double XH[36];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double a = 0.0;
      for (int k = 0; k < 6; ++k) a += X[6 * i + k] * 0.5 * (H[6 * k + j] + H[6 * j + k]);
      XH[6 * i + j] = a;
    }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double a = 0.0;
      for (int k = 0; k < 6; ++k) a += XH[6 * i + k] * X[6 * j + k];
      out[6 * i + j] = a;
    }
}
__device__ inline void p_S(const JointDesc& d, double* S)
{
  const bool rev = d.flags & JF_REVOLUTE;
  for (int k = 0; k < 3; ++k) { S[k] = rev ? 0.0 : d.axis[k]; S[3 + k] = rev ? d.axis[k] : 0.0; }
  if (d.flags & JF_HELICAL)   // JointModelHelical*: S = [pitch a; a]
    for (int k = 0; k < 3; ++k) S[k] = d.pitch * d.axis[k];
}
__device__ inline double p_inf6(const double* x)
{
  double m = 0.0;
  for (int k = 0; k < 6; ++k) m = fmax(m, fabs(x[k]));
  return m;
}

// ---- the pass state of an instance is loaded from the solver's tiles (after SolveInit / a solve) -----------------------------
template <typename T>
__global__ void k_pass_load(char* tiles, Layout TL, const JointDesc* __restrict__ jd, const T* __restrict__ uni, PassLayout L,
                            PassParams P, double* __restrict__ st)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= L.B) return;
  double* s = st + (size_t)b * L.stride;
  char* lp = lane_ptr<T>(tiles, TL, b);
  for (int k = 0; k < L.stride; ++k) s[k] = 0.0;
  // universe: liMi[0] = identity, His[0] = identity
  s[L.liMi + 0] = s[L.liMi + 4] = s[L.liMi + 8] = 1.0;
  for (int k = 0; k < 6; ++k) { s[L.His + 7 * k] = 1.0; s[L.His_aba + 7 * k] = 1.0; }
  for (int i = 1; i < L.nj; ++i) {
    const char* rec = lp + (size_t)(i - 1) * JREC * pair_bytes<T>();
    T R[9], t[3];
    const typename Vec2<T>::type cs = ldp<T>(rec, JP_CS);
    joint_xform<T>(jd[i], rec, cs.x, cs.y, R, t);
    for (int k = 0; k < 9; ++k) s[L.liMi + 12 * i + k] = (double)R[k];
    for (int k = 0; k < 3; ++k) s[L.liMi + 12 * i + 9 + k] = (double)t[k];
    T v[6], f[6], g[6];
    ld6<T>(rec, JP_V, v); ld6<T>(rec, JP_F, f); ld6<T>(rec, JP_G, g);
    for (int k = 0; k < 6; ++k) { s[L.vis + 6 * i + k] = (double)v[k]; s[L.fis + 6 * i + k] = (double)f[k]; s[L.g + 6 * i + k] = (double)g[k]; }
    const typename Vec2<T>::type wz = ldp<T>(rec, JP_WZ), nus = ldp<T>(rec, JP_NUS);
    const int j = i - 1;
    s[L.w + j] = (double)wz.x; s[L.z + j] = (double)wz.y; s[L.nu + j] = (double)nus.x; s[L.Stf + j] = (double)nus.y;
    if (P.bnd_shared) { s[L.lb + j] = (double)uni[TL.nc * 57 + j]; s[L.ub + j] = (double)uni[TL.nc * 57 + TL.nb + j]; }
    else { const typename Vec2<T>::type lu = ldp<T>(rec, JP_LBUB); s[L.lb + j] = (double)lu.x; s[L.ub + j] = (double)lu.y; }
  }
  for (int c = 0; c < L.nc; ++c) {
    char* crec = lp + (size_t)(TL.off_c + c * TL.crec) * pair_bytes<T>();
    T y[6], aty[6], bb[6], atb[6];
    ld6<T>(crec, CP_Y, y); ld6<T>(crec, CP_ATY, aty); ld6<T>(crec, CP_B, bb); ld6<T>(crec, CP_ATB, atb);
    for (int k = 0; k < 6; ++k) {
      s[L.yis + 6 * c + k] = (double)y[k]; s[L.Aty + 6 * c + k] = (double)aty[k];
      s[L.b + 6 * c + k] = (double)bb[k]; s[L.Atb + 6 * c + k] = (double)atb[k];
    }
    for (int q = 0; q < 36; ++q)
      s[L.A + 36 * c + q] = P.a_shared ? (double)uni[c * 36 + q] : (double)*elem_ptr<T>(crec, CP_A + q / 2, q & 1);
    for (int r = 0; r < 6; ++r)
      for (int cc = 0; cc < 6; ++cc) {
        const int k = sym(r, cc);
        s[L.AtA + 36 * c + 6 * r + cc] = P.a_shared ? (double)uni[TL.nc * 36 + c * 21 + k] : (double)*elem_ptr<T>(crec, CP_ATA + k / 2, k & 1);
      }
  }
  const char* srec = lp + (size_t)TL.off_s * pair_bytes<T>();
  const typename Vec2<T>::type mu2 = ldp<T>(srec, SP_MU), bi2 = ldp<T>(srec, SP_BI), st2 = ldp<T>(srec, SP_ST);
  double* sc = s + L.scal;
  sc[PS_MU] = (double)mu2.x; sc[PS_MU_EQ] = P.mu_scale * (double)mu2.x; sc[PS_MU_IN] = (double)mu2.x;
  sc[PS_ITER] = (double)bi2.y; sc[PS_BIS_INF] = (double)bi2.x;
  const int status = (int)st2.x;
  sc[PS_CONVERGED] = (status & ST_CONVERGED) ? 1.0 : 0.0;
  sc[PS_PRIMAL_INF] = (status & ST_PRIMAL_INF) ? 1.0 : 0.0;
}

// ---- ... and written back to them after a logged solve (k_pass_solve): the persistent part of the data object and the
// solver's scalars, so that what follows in the tiles -- a warm-started tailored Solve (Reset(true) keeps w, z, nu, the duals:
// loik-loid-data-optimized.hxx:114-127), loikb_integrate (reads z), the getters of the fused engines -- continues from the
// logged solve's result, not from the state before it.
template <typename T>
__global__ void k_pass_store(char* tiles, Layout TL, PassLayout L, PassParams P, const double* __restrict__ st)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= L.B) return;
  const double* s = st + (size_t)b * L.stride;
  char* lp = lane_ptr<T>(tiles, TL, b);
  for (int i = 1; i < L.nj; ++i) {
    char* rec = lp + (size_t)(i - 1) * JREC * pair_bytes<T>();
    T v[6], f[6], g[6];
    for (int k = 0; k < 6; ++k) { v[k] = (T)s[L.vis + 6 * i + k]; f[k] = (T)s[L.fis + 6 * i + k]; g[k] = (T)s[L.g + 6 * i + k]; }
    st6<T>(rec, JP_V, v); st6<T>(rec, JP_F, f); st6<T>(rec, JP_G, g);
    const int j = i - 1;
    stp<T>(rec, JP_WZ, (T)s[L.w + j], (T)s[L.z + j]);
    stp<T>(rec, JP_NUS, (T)s[L.nu + j], (T)s[L.Stf + j]);
  }
  for (int c = 0; c < L.nc; ++c) {
    char* crec = lp + (size_t)(TL.off_c + c * TL.crec) * pair_bytes<T>();
    T y[6], aty[6];
    for (int k = 0; k < 6; ++k) { y[k] = (T)s[L.yis + 6 * c + k]; aty[k] = (T)s[L.Aty + 6 * c + k]; }
    st6<T>(crec, CP_Y, y); st6<T>(crec, CP_ATY, aty);
  }
  const double* sc = s + L.scal;
  char* srec = lp + (size_t)TL.off_s * pair_bytes<T>();
  const double mu = sc[PS_MU];
  const int kexp = (int)floor(log10(mu / P.mu0) + 0.5);  // mu = mu0 * 10^k under the DEFAULT rule (OSQP: the engines that read k do not run)
  const bool conv = sc[PS_CONVERGED] != 0.0, pinf = sc[PS_PRIMAL_INF] != 0.0;
  const int status = (conv ? ST_CONVERGED : 0) | (pinf ? ST_PRIMAL_INF : 0) | ((pinf && !conv) ? ST_TAIL : 0) | ST_DONE;
  stp<T>(srec, SP_MU, (T)mu, (T)kexp);
  st_hi<T>(srec, SP_BI, (T)sc[PS_ITER]);
  stp<T>(srec, SP_ST, (T)status, (T)sc[PS_MU_IN]);
  stp<T>(srec, SP_TAG, T(-1), T(0));   // no UDinv / Dinv of this solve in the tiles: the next engine rebuilds its cache
  const int map[NSCAL] = {PS_PRIMAL, PS_DUAL, PS_PR_TASK, PS_PR_SLACK, PS_DUAL_V, PS_DUAL_NU, PS_TOL_P, PS_TOL_D, PS_MU, PS_MU_EQ, PS_MU_IN,
                          PS_DX, PS_DZ_INF, PS_DYQP, PS_ATDY, PS_UBP, PS_LBM, PS_DFIS_INF, PS_DYIS_INF, PS_DW_INF, PS_DVIS_INF, PS_DNU_INF,
                          PS_AV_INF, PS_NU_INF, PS_HREFV_INF, PS_G_INF, PS_STF_INF, PS_C1, PS_C2, PS_TAIL_IT};
  for (int k = 0; k < NSCAL; ++k) st_scal<T>(srec, k, (T)sc[map[k]]);
}

enum : int {  // loikb_pass ids (include/loik_amd.h)
  PASS_BEGIN_ITERATION = 0, PASS_FWD1, PASS_BWD, PASS_FWD2, PASS_BOXPROJ, PASS_DUAL, PASS_RESIDUALS, PASS_CHECK_CONV,
  PASS_CHECK_FEAS, PASS_UPDATE_MU
};

// one pass of one instance (its block `s` of the pass state); `cslot_of[i]` = constraint slot of joint i or -1
__device__ inline void pass_one(int pass, const PassLayout& L, const PassParams& P, const JointDesc* __restrict__ jd,
                                const int* __restrict__ cslot_of, double* __restrict__ s)
{
  double* sc = s + L.scal;
  const int nj = L.nj;
  switch (pass) {
  case PASS_BEGIN_ITERATION: {
    // iter_ = i (hpp:381); UpdatePrev (data-optimized.hxx:192-197); ResetInfNorms (:165-182)
    sc[PS_ITER] += 1.0;
    for (int k = 0; k < nj * 6; ++k) s[L.vis_prev + k] = s[L.vis + k];
    for (int k = 0; k < L.nv; ++k) { s[L.nu_prev + k] = s[L.nu + k]; s[L.z_prev + k] = s[L.z + k]; }
    for (int q : {PS_NU_INF, PS_DFIS_INF, PS_HREFV_INF, PS_DVIS_INF, PS_DNU_INF, PS_DZ_INF, PS_DYIS_INF, PS_AV_INF, PS_BTDY_PLUS,
                  PS_BTDY_MINUS, PS_DW_INF, PS_G_INF, PS_DG_INF, PS_STF_INF, PS_DSTF_INF})
      sc[q] = 0.0;
  } break;
  case PASS_FWD1: {
    // hxx:290-338.  (vis_prev is the iterate of the previous iteration: right after SolveInit it equals vis)
    const double mu_eq = sc[PS_MU_EQ], mu_in = sc[PS_MU_IN];
    for (int k = 0; k < L.nv; ++k) { s[L.R + k] = mu_in; s[L.r + k] = s[L.w + k] - mu_in * s[L.z + k]; }
    for (int i = 1; i < nj; ++i) {
      double* H = s + L.His + 36 * i; double* p = s + L.pis + 6 * i;
      const double* vp = s + L.vis_prev + 6 * i;
      const double rho = (jd[i].flags & JF_MASSLESS) ? 0.0 : P.rho;  // (the table's row of a massless link is zero)
      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) H[6 * r + c] = (r == c ? rho : 0.0) + P.href_tab[i * HREF_ROW + 6 * r + c];
        p[r] = -rho * vp[r] - P.href_tab[i * HREF_ROW + 36 + r];
      }
      const int cs = cslot_of[i];
      if (cs >= 0) {
        for (int q = 0; q < 36; ++q) H[q] += mu_eq * s[L.AtA + 36 * cs + q];
        for (int r = 0; r < 6; ++r) p[r] += s[L.Aty + 6 * cs + r] - mu_eq * s[L.Atb + 6 * cs + r];
      }
      for (int q = 0; q < 36; ++q) s[L.His_aba + 36 * i + q] = H[q];
      for (int r = 0; r < 6; ++r) s[L.pis_aba + 6 * i + r] = p[r];
    }
  } break;
  case PASS_BWD: {
    // hxx:345-354 + LoikBackwardStepVisitor::algo :31-81, joints nb .. 1
    for (int i = nj - 1; i >= 1; --i) {
      const JointDesc& d = jd[i];
      const int par = d.parent, j = i - 1;
      double S[6];
      p_S(d, S);
      double* Ha = s + L.His_aba + 36 * i;
      double U[6], dinv, sus = 0.0;
      for (int r = 0; r < 6; ++r) {
        double a = 0.0;
        for (int c = 0; c < 6; ++c) a += Ha[6 * r + c] * S[c];
        U[r] = a;
      }
      for (int r = 0; r < 6; ++r) sus += S[r] * U[r];
      dinv = 1.0 / (sus + s[L.R + j]);
      double* UD = s + L.UDinv + 6 * i;
      for (int r = 0; r < 6; ++r) UD[r] = U[r] * dinv;
      s[L.Dinv + i] = dinv;
      if (par > 0)  // calc_aba(..., update_I = parent > 0)
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c) Ha[6 * r + c] -= UD[r] * U[c];
      double tr[36];
      p_congruence(s + L.liMi + 12 * i, Ha, tr);
      for (int q = 0; q < 36; ++q) { s[L.His_aba + 36 * par + q] += tr[q]; s[L.His + 36 * par + q] = s[L.His_aba + 36 * par + q]; }
      double sp = 0.0;
      for (int r = 0; r < 6; ++r) sp += S[r] * s[L.pis + 6 * i + r];
      s[L.r + j] += sp;
      double pa[6], tp[6];
      for (int r = 0; r < 6; ++r) { s[L.pis_aba + 6 * i + r] -= UD[r] * s[L.r + j]; pa[r] = s[L.pis_aba + 6 * i + r]; }
      p_act_force(s + L.liMi + 12 * i, pa, tp);
      for (int r = 0; r < 6; ++r) { s[L.pis + 6 * par + r] += tp[r]; s[L.pis_aba + 6 * par + r] = s[L.pis + 6 * par + r]; }
    }
  } break;
  case PASS_FWD2: {
    // hxx:361-377 + LoikForwardStep2Visitor::algo :102-163
    for (int i = 1; i < nj; ++i) {
      const JointDesc& d = jd[i];
      const int par = d.parent, j = i - 1;
      double S[6], vp[6];
      p_S(d, S);
      p_actinv_motion(s + L.liMi + 12 * i, s + L.vis + 6 * par, vp);
      double udv = 0.0;
      for (int r = 0; r < 6; ++r) udv += s[L.UDinv + 6 * i + r] * vp[r];
      const double nu = -udv - s[L.Dinv + i] * s[L.r + j];
      s[L.nu + j] = nu;
      sc[PS_NU_INF] = fmax(sc[PS_NU_INF], fabs(nu));
      double v[6], f[6], df[6], hv[6], dv[6];
      for (int r = 0; r < 6; ++r) v[r] = vp[r] + S[r] * nu;
      for (int r = 0; r < 6; ++r) {
        double a = s[L.pis + 6 * i + r], h = 0.0;
        for (int c = 0; c < 6; ++c) { a += s[L.His + 36 * i + 6 * r + c] * v[c]; h += P.href_tab[i * HREF_ROW + 6 * r + c] * v[c]; }
        f[r] = a; hv[r] = h;
        df[r] = f[r] - s[L.fis + 6 * i + r];
        dv[r] = v[r] - s[L.vis_prev + 6 * i + r];
      }
      if (!(d.flags & JF_MASSLESS)) {  // norms over the LINKS of the caller's model
        sc[PS_DFIS_INF] = fmax(sc[PS_DFIS_INF], p_inf6(df));
        sc[PS_HREFV_INF] = fmax(sc[PS_HREFV_INF], p_inf6(hv));
        sc[PS_DVIS_INF] = fmax(sc[PS_DVIS_INF], p_inf6(dv));
      }
      for (int r = 0; r < 6; ++r) { s[L.vis + 6 * i + r] = v[r]; s[L.fis + 6 * i + r] = f[r]; s[L.Href_v + 6 * i + r] = hv[r]; }
    }
    double dn = 0.0;
    for (int k = 0; k < L.nv; ++k) dn = fmax(dn, fabs(s[L.nu + k] - s[L.nu_prev + k]));
    sc[PS_DNU_INF] = dn;
  } break;
  case PASS_BOXPROJ: {
    // hxx:384-397
    const double mu_in = sc[PS_MU_IN];
    double dz = 0.0, prs = 0.0;
    for (int k = 0; k < L.nv; ++k) {
      const double x = s[L.nu + k] + (1.0 / mu_in) * s[L.w + k];
      const double z = fmin(s[L.ub + k], fmax(s[L.lb + k], x));
      s[L.z + k] = z;
      dz = fmax(dz, fabs(z - s[L.z_prev + k]));
      prs = fmax(prs, fabs(s[L.nu + k] - z));
    }
    sc[PS_DZ_INF] = dz;
    sc[PS_PR_SLACK] = prs;
  } break;
  case PASS_DUAL: {
    // hxx:404-461
    const double mu_eq = sc[PS_MU_EQ], mu_in = sc[PS_MU_IN];
    double prt = 0.0;
    for (int i = 1; i < nj; ++i) {
      const int c = cslot_of[i];
      if (c < 0) continue;
      const double* A = s + L.A + 36 * c;
      double av[6], dy[6];
      for (int r = 0; r < 6; ++r) {
        double a = 0.0;
        for (int k = 0; k < 6; ++k) a += A[6 * r + k] * s[L.vis + 6 * i + k];
        av[r] = a;
        const double e = a - s[L.b + 6 * c + r];
        dy[r] = mu_eq * e;
        s[L.yis + 6 * c + r] += dy[r];
        prt = fmax(prt, fabs(e));
        sc[PS_BTDY_PLUS] += s[L.b + 6 * c + r] * fmax(dy[r], 0.0);
        sc[PS_BTDY_MINUS] += s[L.b + 6 * c + r] * fmin(dy[r], 0.0);
      }
      sc[PS_DYIS_INF] = fmax(sc[PS_DYIS_INF], p_inf6(dy));
      sc[PS_AV_INF] = fmax(sc[PS_AV_INF], p_inf6(av));
      for (int r = 0; r < 6; ++r) {
        double a = 0.0;
        for (int k = 0; k < 6; ++k) a += A[6 * k + r] * s[L.yis + 6 * c + k];
        s[L.Aty + 6 * c + r] = a;
      }
    }
    sc[PS_PR_TASK] = prt;
    double dwm = 0.0;
    for (int k = 0; k < L.nv; ++k) {
      const double dw = mu_in * (s[L.nu + k] - s[L.z + k]);
      s[L.delta_w + k] = dw;
      s[L.w + k] += dw;
      dwm = fmax(dwm, fabs(dw));
    }
    sc[PS_DW_INF] = dwm;
  } break;
  case PASS_RESIDUALS: {
    // ComputePrimalResiduals hxx:494-503; ComputeDualResiduals :510-522 -> BwdPass2 :468-487, :185-241
    sc[PS_PRIMAL] = fmax(sc[PS_PR_TASK], sc[PS_PR_SLACK]);
    // g_i = A^T y (constrained links, seeded by DualUpdate hxx:438-439) - f_i + sum over children of act(f_child); the sums
    // are built in `gnew` while the old g_i is still needed for delta g (hxx:215)
    double dualv = 0.0;
    for (int i = 0; i < nj; ++i)
      for (int r = 0; r < 6; ++r) {
        const int c = i > 0 ? cslot_of[i] : -1;
        s[L.gnew + 6 * i + r] = c >= 0 ? s[L.Aty + 6 * c + r] : 0.0;
      }
    double stfm = 0.0, dstf = 0.0;
    for (int i = nj - 1; i >= 1; --i) {
      const JointDesc& d = jd[i];
      const int par = d.parent, j = i - 1;
      double S[6], tf[6], gi[6], dg[6], dvr[6];
      p_S(d, S);
      for (int r = 0; r < 6; ++r) { gi[r] = s[L.gnew + 6 * i + r] - s[L.fis + 6 * i + r]; dg[r] = gi[r] - s[L.g + 6 * i + r]; }
      p_act_force(s + L.liMi + 12 * i, s + L.fis + 6 * i, tf);
      for (int r = 0; r < 6; ++r) s[L.gnew + 6 * par + r] += tf[r];
      for (int r = 0; r < 6; ++r) { dvr[r] = s[L.Href_v + 6 * i + r] - P.href_tab[i * HREF_ROW + 36 + r] + gi[r]; s[L.g + 6 * i + r] = gi[r]; }
      if (!(d.flags & JF_MASSLESS)) {
        sc[PS_G_INF] = fmax(sc[PS_G_INF], p_inf6(gi));
        sc[PS_DG_INF] = fmax(sc[PS_DG_INF], p_inf6(dg));
        dualv = fmax(dualv, p_inf6(dvr));
      }
      double sf = 0.0;
      for (int r = 0; r < 6; ++r) sf += S[r] * s[L.fis + 6 * i + r];
      const double si = sf + s[L.w + j];
      stfm = fmax(stfm, fabs(si));
      dstf = fmax(dstf, fabs(si - s[L.Stf + j]));
      s[L.Stf + j] = si;
    }
    sc[PS_STF_INF] = stfm; sc[PS_DSTF_INF] = dstf;
    sc[PS_DUAL_V] = dualv; sc[PS_DUAL_NU] = stfm;
    sc[PS_DUAL] = fmax(dualv, stfm);
  } break;
  case PASS_CHECK_CONV: {
    // hxx:540-565 (nu_inf_norm twice)
    sc[PS_TOL_P] = P.tol_abs + P.tol_rel * fmax(fmax(sc[PS_AV_INF], sc[PS_NU_INF]), fmax(sc[PS_BIS_INF], sc[PS_NU_INF]));
    sc[PS_TOL_D] = P.tol_abs + P.tol_rel * fmax(fmax(sc[PS_HREFV_INF], fmax(sc[PS_G_INF], sc[PS_STF_INF])), P.Hv_inf_norm);
    if (sc[PS_PRIMAL] < sc[PS_TOL_P] && sc[PS_DUAL] < sc[PS_TOL_D]) sc[PS_CONVERGED] = 1.0;
  } break;
  case PASS_CHECK_FEAS: {
    // hxx:572-606
    sc[PS_DYQP] = fmax(sc[PS_DFIS_INF], fmax(sc[PS_DYIS_INF], sc[PS_DW_INF]));
    sc[PS_ATDY] = fmax(sc[PS_DG_INF], sc[PS_DSTF_INF]);
    sc[PS_C1] = sc[PS_ATDY] <= P.tol_primal_inf * sc[PS_DYQP] ? 1.0 : 0.0;
    double up = sc[PS_BTDY_PLUS], lm = sc[PS_BTDY_MINUS], a = 0.0, c = 0.0;
    for (int k = 0; k < L.nv; ++k) { a += s[L.ub + k] * fmax(s[L.delta_w + k], 0.0); c += s[L.lb + k] * fmin(s[L.delta_w + k], 0.0); }
    up += a; lm += c;
    sc[PS_UBP] = up; sc[PS_LBM] = lm;
    sc[PS_C2] = (up + lm) <= P.tol_primal_inf * sc[PS_DYQP] ? 1.0 : 0.0;
    if (sc[PS_C1] != 0.0 && sc[PS_C2] != 0.0) sc[PS_PRIMAL_INF] = 1.0;
    sc[PS_DX] = fmax(sc[PS_DVIS_INF], sc[PS_DNU_INF]);
  } break;
  case PASS_UPDATE_MU: {
    // hxx:613-641 (DEFAULT); OSQP: the extension, see update_mu() in loik_device.hpp
    double mu = sc[PS_MU];
    int kexp = 0;
    update_mu<double>(P.mu_osqp ? MODE_MU_OSQP : 0, sc[PS_PRIMAL], sc[PS_DUAL], fmax(fmax(sc[PS_AV_INF], sc[PS_NU_INF]), sc[PS_BIS_INF]),
                      fmax(fmax(sc[PS_HREFV_INF], fmax(sc[PS_G_INF], sc[PS_STF_INF])), P.Hv_inf_norm), mu, kexp);
    sc[PS_MU] = mu; sc[PS_MU_EQ] = P.mu_scale * mu; sc[PS_MU_IN] = mu;
  } break;
  default: break;
  }
}

__global__ void k_pass(int pass, PassLayout L, PassParams P, const JointDesc* __restrict__ jd, const int* __restrict__ cslot_of,
                       double* __restrict__ st)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= L.B) return;
  pass_one(pass, L, P, jd, cslot_of, st + (size_t)b * L.stride);
}

// ---- Solve() with logging_ = true (loik-loid-optimized.hpp:375-455, :271-319): the main loop and the infeasibility tail solve
// of every instance written out pass by pass, one thread per instance, with the lists of LoikSolverInfo (hpp:47-127) filled the
// way upstream fills them -- after ComputeResiduals, before the stopping tests; mu_list_ therefore holds the mu the iteration
// RAN with.  "logging residuals, should be disabled for speed" (hpp:408): this is the plain implementation, not the engines.
//   log[(list * B + b) * rows_cap + k], k = iteration - 1 (zero beyond rows[b]);  rows[b] = entries of the residual / mu lists
// (the tail solve appends to iter_list_ / tail_solve_iter_list_ only: its length is the instance's tail_solve_iter).
enum : int { LOG_PR_TASK = 0, LOG_PR_SLACK, LOG_PRIMAL, LOG_DUAL_NU, LOG_DUAL_V, LOG_DUAL, LOG_MU, LOG_MU_EQ, LOG_MU_INEQ, LOG_NLIST };

__global__ void k_pass_solve(PassLayout L, PassParams P, const JointDesc* __restrict__ jd, const int* __restrict__ cslot_of,
                             double* __restrict__ st, double* __restrict__ log, int rows_cap, int* __restrict__ rows)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= L.B) return;
  double* s = st + (size_t)b * L.stride;
  double* sc = s + L.scal;
  auto body = [&]() {  // UpdatePrev ... ComputeResiduals (hpp:381-404)
    for (int pass = PASS_BEGIN_ITERATION; pass <= PASS_RESIDUALS; ++pass) pass_one(pass, L, P, jd, cslot_of, s);
  };
  int n = 0;
  sc[PS_TAIL_IT] = 0.0;
  for (int i = 1; i < P.max_iter; ++i) {
    body();  // (PASS_BEGIN_ITERATION: iter_ = i)
    if (log != nullptr && n < rows_cap) {
      const double v[LOG_NLIST] = {sc[PS_PR_TASK], sc[PS_PR_SLACK], sc[PS_PRIMAL], sc[PS_DUAL_NU], sc[PS_DUAL_V], sc[PS_DUAL],
                                   sc[PS_MU], sc[PS_MU_EQ], sc[PS_MU_IN]};
      for (int l = 0; l < LOG_NLIST; ++l) log[((size_t)l * L.B + b) * rows_cap + n] = v[l];
      ++n;
    }
    pass_one(PASS_CHECK_CONV, L, P, jd, cslot_of, s);
    if (i > 1) pass_one(PASS_CHECK_FEAS, L, P, jd, cslot_of, s);
    if (sc[PS_CONVERGED] != 0.0) break;
    if (sc[PS_PRIMAL_INF] != 0.0) {
      // InfeasibilityTailSolve (hpp:271-319)
      while (sc[PS_DX] >= P.tol_tail_solve || sc[PS_DZ_INF] >= P.tol_tail_solve) {
        if ((int)sc[PS_ITER] >= P.max_iter) break;
        sc[PS_TAIL_IT] += 1.0;
        body();  // iter_++
        sc[PS_DX] = fmax(sc[PS_DVIS_INF], sc[PS_DNU_INF]);
      }
      break;
    }
    pass_one(PASS_UPDATE_MU, L, P, jd, cslot_of, s);
  }
  if (rows != nullptr) rows[b] = n;   // (null: a solve without SolverInfo lists -- the engine of last resort)
}

// out[b][...] of one field of the pass state (instance-major, the layouts of loikb_get)
__global__ void k_pass_get(PassLayout L, const double* __restrict__ st, int off, int n, int skip, double* __restrict__ out)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= L.B) return;
  const double* s = st + (size_t)b * L.stride + off + skip;
  for (int k = 0; k < n; ++k) out[(size_t)b * n + k] = s[k];
}
// a per-link field: link l of the caller's model (l = 0 .. nl-1) is device joint sel[l] + 1 (the body-carrying link of its chain)
__global__ void k_pass_get_links(PassLayout L, const double* __restrict__ st, int off, int width, const int* __restrict__ sel, int nl,
                                 double* __restrict__ out)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= L.B) return;
  const double* s = st + (size_t)b * L.stride + off;
  for (int l = 0; l < nl; ++l)
    for (int k = 0; k < width; ++k) out[((size_t)b * nl + l) * width + k] = s[width * (sel[l] + 1) + k];
}
__global__ void k_pass_get_his(PassLayout L, const double* __restrict__ st, const int* __restrict__ sel, int nl,
                               double* __restrict__ out)  // [B][nl][21]
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= L.B) return;
  const double* s = st + (size_t)b * L.stride + L.His;
  for (int l = 0; l < nl; ++l) {
    const int i = sel[l] + 1;
    int k = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c, ++k) out[((size_t)b * nl + l) * 21 + k] = s[36 * i + 6 * r + c];
  }
}

}  // namespace loikb
