// loik_device.hpp -- CDNA4 (gfx950) device code of the batched LoIK ADMM solve.
//
// One problem instance per lane, 64-lane wavefronts, one wavefront per workgroup.  All per-instance
// arrays are struct-of-arrays with the batch index innermost ([field][joint][component][B]) so every
// vector memory instruction of a wavefront is one contiguous 512-byte (fp64) / 256-byte (fp32) run.
// The kinematic tree (parents[], joint type/axis, jointPlacements) is baked on the host into a
// per-joint `JointDesc` schedule that is uniform across lanes (scalar loads, SGPR operands).
//
// What each sweep computes, and the reference function it replaces (paths under /root/reference/):
//   sweep_bwd   : FwdPass1 + BwdPassOptimizedVisitor/LoikBackwardStepVisitor
//                 (include/loik/loik-loid-optimized.hxx:290-338, :345-354, :31-81)
//   sweep_fwd   : FwdPass2OptimizedVisitor/LoikForwardStep2Visitor + BoxProj + DualUpdate
//                 (hxx:361-377, :102-163, :384-397, :404-461)
//   sweep_bwd2  : BwdPass2OptimizedVisitor/LoikBackwardStep2Visitor + Compute{Primal,Dual}Residuals
//                 (hxx:468-487, :185-241, :494-522)
//   epilogue    : CheckConvergence, CheckFeasibility, UpdateMu, InfeasibilityTailSolve control
//                 (hxx:540-641, include/loik/loik-loid-optimized.hpp:271-319, :377-454)
//   k_fk_init   : FwdPassInit (hxx:253-283)
//
// No MFMA: the largest contraction is 6x6.  The path is HBM-bound; see DESIGN.md for the byte model.
#pragma once

#include <hip/hip_runtime.h>

namespace loikb {

constexpr int WAVE = 64;

// joint-schedule flags (uniform per joint)
enum : int {
  JF_LEAF = 1,            // no children: children-accumulator starts at 0
  JF_PARENT_ROOT = 2,     // parent is the universe: contribution discarded (update_I = parent > 0, hxx:63)
  JF_LAST_CHILD = 4,      // largest-index child of its parent: first one visited in a leaf->root sweep
  JF_NEXT_IS_PARENT = 8,  // i-1 == parent: partial sum stays in registers, else it is pushed to the LDS stack
  JF_REVOLUTE = 16,       // S = [0; axis], else prismatic S = [axis; 0]
};

// rotation generator selector for M(q)
enum : int { ROT_X = 0, ROT_Y = 1, ROT_Z = 2, ROT_U = 3, ROT_NONE = 4 };

struct JointDesc {
  double Rp[9];   // jointPlacements[i].rotation(), row-major
  double tp[3];   // jointPlacements[i].translation()
  double axis[3]; // joint axis in the joint frame
  int parent;
  int flags;
  int cslot;      // index into the active constraint list, or -1
  int rot;        // ROT_*
};

// status bits per instance
enum : int { ST_CONVERGED = 1, ST_PRIMAL_INF = 2, ST_TAIL = 4, ST_DONE = 8 };

// solver mode flags (uniform)
enum : int {
  MODE_FIXED_ITERS = 1,  // no convergence / feasibility / mu logic: exactly max_iter-1 iterations
  MODE_CACHE_H = 2,      // skip the H-recursion of the leaf->root sweep while no live lane changed mu
  MODE_A_SHARED = 4,     // one A per constraint for the whole batch
  MODE_BND_SHARED = 8,   // one lb/ub for the whole batch
};

// per-iteration scalars dumped for the getters / parity tests: rows of `scal[NSCAL][ld]`
enum : int {
  SC_PRIMAL_RES = 0, SC_DUAL_RES, SC_PRIMAL_RES_TASK, SC_PRIMAL_RES_SLACK, SC_DUAL_RES_V, SC_DUAL_RES_NU,
  SC_TOL_PRIMAL, SC_TOL_DUAL, SC_MU, SC_MU_EQ, SC_MU_INEQ,
  SC_DELTA_X_QP, SC_DELTA_Z_QP, SC_DELTA_Y_QP, SC_AT_DELTA_Y_QP, SC_UB_DY_PLUS, SC_LB_DY_MINUS,
  SC_DELTA_FIS, SC_DELTA_YIS, SC_DELTA_W, SC_DELTA_VIS, SC_DELTA_NU,
  SC_AV_INF, SC_NU_INF, SC_HREF_V_INF, SC_G_INF, SC_STF_PLUS_W_INF,
  SC_COND1, SC_COND2, SC_TAIL_ITER,
  NSCAL
};

template <typename T>
struct Params {
  // uniform problem data (UpdateReference broadcasts ONE H_ref,v_ref: ik-id-description-optimized.hpp:78-97)
  T Href[36];  // full matrix (used as-is for Href*v, hxx:149)
  T Hv[6];     // H_ref * v_ref
  T Hv_inf_norm;
  T rho, mu0, mu_scale;
  T tol_abs, tol_rel, tol_primal_inf, tol_tail_solve;
  int max_iter;
  int mode;
  int nb;   // joints 1..nb
  int nc;   // active constraints
  int B;    // instances in this launch (slots 0..B-1)
  int ld;   // leading dimension (padded batch) of every SoA array
  int max_launch_iters;
  int stack_levels;
};

template <typename T>
struct Bufs {
  // configuration: cos/sin (revolute) or q,0 (prismatic) per joint: [nb][2][ld]
  const T* cs;
  // persistent ADMM state
  T* v;    // [nb][6][ld]  vis
  T* f;    // [nb][6][ld]  fis
  T* g;    // [nb][6][ld]  fis_diff_plus_Aty
  T* nu;   // [nb][ld]
  T* z;    // [nb][ld]
  T* w;    // [nb][ld]
  T* s;    // [nb][ld]     Stf_plus_w
  T* y;    // [nc][6][ld]  yis
  T* aty;  // [nc][6][ld]  Aty
  // inter-sweep temporaries (within one iteration)
  T* H;    // [nb][21][ld] His (pre-projection, accumulated), symmetric packed
  T* p;    // [nb][6][ld]  pis
  T* ud;   // [nb][6][ld]  UDinv
  T* dinv; // [nb][ld]
  T* rr;   // [nb][ld]     r after += S^T p
  // inputs
  const T* A;    // [nc][36] (shared) or [nc][36][ld]
  const T* AtA;  // [nc][21] or [nc][21][ld]
  const T* b;    // [nc][6][ld]
  const T* Atb;  // [nc][6][ld]
  const T* lb;   // [nb] or [nb][ld]
  const T* ub;
  const T* bnorm;  // [ld] bis_inf_norm_ per instance
  // per-instance solver scalars
  T* mu;         // [ld]
  T* mu_h;       // [ld] mu the cached H/UDinv/Dinv were computed with (MODE_CACHE_H)
  int* iter;     // [ld]
  int* status;   // [ld]
  T* scal;       // [NSCAL][ld]
  unsigned int* counters;  // [0] live instances at exit, [1] instance-iterations executed
  int* wave_live;          // [ld/64] live lanes of each wavefront at exit (feeds the host-side compaction scan)
};

// ------------------------------------------------------------------------------------------------
// small fixed-size algebra, everything fully unrolled so all register arrays are statically indexed
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int sym(int i, int j)
{
  // upper-triangular row-major packing of a symmetric 6x6
  return (i <= j) ? (i * 6 - (i * (i - 1)) / 2 + (j - i)) : (j * 6 - (j * (j - 1)) / 2 + (i - j));
}

template <typename T>
__device__ __forceinline__ T tabs(T x) { return x < T(0) ? -x : x; }
template <typename T>
__device__ __forceinline__ T tmax(T a, T b) { return a > b ? a : b; }
template <typename T>
__device__ __forceinline__ T tmin(T a, T b) { return a < b ? a : b; }

template <typename T>
__device__ __forceinline__ T inf6(const T* x)
{
  T m = tabs(x[0]);
#pragma unroll
  for (int k = 1; k < 6; ++k) m = tmax(m, tabs(x[k]));
  return m;
}

template <typename T>
__device__ __forceinline__ void cross3(const T* a, const T* b, T* o)
{
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

template <typename T>
__device__ __forceinline__ void mat3_vec(const T* A, const T* x, T* y)
{
#pragma unroll
  for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}

template <typename T>
__device__ __forceinline__ void mat3t_vec(const T* A, const T* x, T* y)
{
#pragma unroll
  for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}

// liMi = jointPlacement * M(q)  (hxx:263-264).  `c`,`s` = cos q, sin q for revolute joints; for prismatic
// joints `c` carries q.
template <typename T>
__device__ __forceinline__ void make_liMi(const JointDesc& d, T c, T s, T* R, T* t)
{
  T Rp[9], tp[3], ax[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rp[k] = (T)d.Rp[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { tp[k] = (T)d.tp[k]; ax[k] = (T)d.axis[k]; }
  if (d.flags & JF_REVOLUTE) {
    T M[9];
    if (d.rot == ROT_X) {
      M[0] = T(1); M[1] = T(0); M[2] = T(0); M[3] = T(0); M[4] = c; M[5] = -s; M[6] = T(0); M[7] = s; M[8] = c;
    } else if (d.rot == ROT_Y) {
      M[0] = c; M[1] = T(0); M[2] = s; M[3] = T(0); M[4] = T(1); M[5] = T(0); M[6] = -s; M[7] = T(0); M[8] = c;
    } else if (d.rot == ROT_Z) {
      M[0] = c; M[1] = -s; M[2] = T(0); M[3] = s; M[4] = c; M[5] = T(0); M[6] = T(0); M[7] = T(0); M[8] = T(1);
    } else {  // Rodrigues: c I + (1-c) a a^T + s [a]x
      const T c1 = T(1) - c;
      T tmp;
      tmp = c1 * ax[0] * ax[1]; M[1] = tmp - s * ax[2]; M[3] = tmp + s * ax[2];
      tmp = c1 * ax[0] * ax[2]; M[2] = tmp + s * ax[1]; M[6] = tmp - s * ax[1];
      tmp = c1 * ax[1] * ax[2]; M[5] = tmp - s * ax[0]; M[7] = tmp + s * ax[0];
      M[0] = c1 * ax[0] * ax[0] + c; M[4] = c1 * ax[1] * ax[1] + c; M[8] = c1 * ax[2] * ax[2] + c;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        R[3 * i + j] = Rp[3 * i] * M[j] + Rp[3 * i + 1] * M[3 + j] + Rp[3 * i + 2] * M[6 + j];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = tp[k];
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = Rp[k];
    T tq[3] = {ax[0] * c, ax[1] * c, ax[2] * c}, rt[3];
    mat3_vec(Rp, tq, rt);
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = tp[k] + rt[k];
  }
}

// SE3::act(Force): (R f_l, R f_a + t x R f_l)           [Pinocchio; call sites hxx:74, :212]
template <typename T>
__device__ __forceinline__ void act_force(const T* R, const T* t, const T* f, T* o)
{
  T a[3], c[3];
  mat3_vec(R, f, o);
  mat3_vec(R, f + 3, a);
  cross3(t, o, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) o[3 + k] = a[k] + c[k];
}

// SE3::actInv(Motion): (R^T (v_l - t x v_a), R^T v_a)     [Pinocchio; call site hxx:125]
template <typename T>
__device__ __forceinline__ void actinv_motion(const T* R, const T* t, const T* v, T* o)
{
  T c[3], d[3];
  cross3(t, v + 3, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = v[k] - c[k];
  mat3t_vec(R, d, o);
  mat3t_vec(R, v + 3, o + 3);
}

// R X R^T for a full 3x3 X
template <typename T>
__device__ __forceinline__ void rot_congr(const T* R, const T* X, T* o)
{
  T tmp[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) tmp[3 * i + j] = R[3 * i] * X[j] + R[3 * i + 1] * X[3 + j] + R[3 * i + 2] * X[6 + j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      o[3 * i + j] = tmp[3 * i] * R[3 * j] + tmp[3 * i + 1] * R[3 * j + 1] + tmp[3 * i + 2] * R[3 * j + 2];
}

// pinocchio::impl::internal::SE3actOn (call site hxx:66): X*(M) H X(M)^-1 for symmetric H given as 21
// packed entries; result symmetric, 21 packed entries.  With A=H[0:3,0:3], B=H[0:3,3:6], D=H[3:6,3:6],
// At=R A R^T, Bt=R B R^T, Dt=R D R^T, T=[t]x:
//   A' = At ; B' = Bt + At T^T ; D' = Dt + (T Bt)^T + T B'
template <typename T>
__device__ __forceinline__ void congr_sym(const T* R, const T* t, const T* h, T* o)
{
  T A[9], Bm[9], D[9], At[9], Bt[9], Dt[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      A[3 * i + j] = h[sym(i, j)];
      Bm[3 * i + j] = h[sym(i, 3 + j)];
      D[3 * i + j] = h[sym(3 + i, 3 + j)];
    }
  rot_congr(R, A, At);
  rot_congr(R, Bm, Bt);
  rot_congr(R, D, Dt);
  // TA[j][k] = (t x At.col(k))[j];  (At T^T)[i][j] = TA[j][i] because At is symmetric
  T Bo[9], TBt[9], TBo[9], col[3], cr[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    col[0] = At[k]; col[1] = At[3 + k]; col[2] = At[6 + k];
    cross3(t, col, cr);
#pragma unroll
    for (int j = 0; j < 3; ++j) Bo[3 * k + j] = Bt[3 * k + j] + cr[j];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    col[0] = Bt[k]; col[1] = Bt[3 + k]; col[2] = Bt[6 + k];
    cross3(t, col, cr);
#pragma unroll
    for (int j = 0; j < 3; ++j) TBt[3 * j + k] = cr[j];
    col[0] = Bo[k]; col[1] = Bo[3 + k]; col[2] = Bo[6 + k];
    cross3(t, col, cr);
#pragma unroll
    for (int j = 0; j < 3; ++j) TBo[3 * j + k] = cr[j];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (i <= j) {
        o[sym(i, j)] = At[3 * i + j];
        o[sym(3 + i, 3 + j)] = Dt[3 * i + j] + TBt[3 * j + i] + TBo[3 * i + j];
      }
      o[sym(i, 3 + j)] = Bo[3 * i + j];
    }
}

// y = H x for symmetric packed H
template <typename T>
__device__ __forceinline__ void symv(const T* h, const T* x, T* y)
{
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    T a = h[sym(i, 0)] * x[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) a += h[sym(i, k)] * x[k];
    y[i] = a;
  }
}

// coalesced SoA access: element (row, lane-slot)
template <typename T>
__device__ __forceinline__ T ld(const T* base, int row, int ldim, int b) { return base[(size_t)row * ldim + b]; }
template <typename T>
__device__ __forceinline__ void st(T* base, int row, int ldim, int b, T x) { base[(size_t)row * ldim + b] = x; }

// per-lane LDS stack of pending branch accumulators: [level][entry][lane]
template <typename T>
__device__ __forceinline__ void stack_push(T* stk, int level, int nent, const T* x, int n, int lane)
{
#pragma unroll
  for (int k = 0; k < 27; ++k)
    if (k < n) stk[(level * nent + k) * WAVE + lane] = x[k];
}
template <typename T>
__device__ __forceinline__ void stack_pop_add(const T* stk, int level, int nent, T* x, int n, int lane)
{
#pragma unroll
  for (int k = 0; k < 27; ++k)
    if (k < n) x[k] += stk[(level * nent + k) * WAVE + lane];
}

// ------------------------------------------------------------------------------------------------
// per-lane running scalars of one ADMM iteration (the reference's 13 inf-norms + 2 dot products,
// loik-loid-data-optimized.hpp:259-329, plus the residual splits)
// ------------------------------------------------------------------------------------------------
template <typename T>
struct Norms {
  T nu_inf, dfis, href_v, dvis, dnu, dz, dw, dyis, av_inf, pr_task, pr_slack;
  T bTdy_plus, bTdy_minus, ub_dw_plus, lb_dw_minus;
  T g_inf, dg, stf_w_inf, dstf_w, dual_v;
  __device__ __forceinline__ void reset()
  {
    nu_inf = dfis = href_v = dvis = dnu = dz = dw = dyis = av_inf = pr_task = pr_slack = T(0);
    bTdy_plus = bTdy_minus = ub_dw_plus = lb_dw_minus = T(0);
    g_inf = dg = stf_w_inf = dstf_w = dual_v = T(0);
  }
};

// ------------------------------------------------------------------------------------------------
// leaf -> root sweep: FwdPass1 + BwdPass.  WITH_H=false re-uses the cached H/UDinv/Dinv (valid while
// mu is unchanged: they depend only on rho, mu, liMi, H_ref, AtA -- never on the iterates).
// ------------------------------------------------------------------------------------------------
template <typename T, bool WITH_H>
__device__ __forceinline__ void sweep_bwd(const Params<T>& P, const Bufs<T>& Bf, const JointDesc* __restrict__ jd,
                                          T* stk, int b, int lane, bool live, T mu_eq, T mu_in)
{
  constexpr int NENT = 27;
  const int ldm = P.ld;
  T accH[21], accp[6];
#pragma unroll
  for (int k = 0; k < 21; ++k) accH[k] = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) accp[k] = T(0);
  int level = 0;
  const int a_bs = (P.mode & MODE_A_SHARED) ? 0 : 1;
  const int a_es = (P.mode & MODE_A_SHARED) ? 1 : ldm;

  for (int i = P.nb; i >= 1; --i) {
    const JointDesc d = jd[i];
    const int j = i - 1;  // storage row of joint i
    if (live) {
      T vprev[6], hh[21], pp[6], U[6], UD[6];
      const T c = ld(Bf.cs, 2 * j, ldm, b), s = ld(Bf.cs, 2 * j + 1, ldm, b);
      const T wi = ld(Bf.w, j, ldm, b), zi = ld(Bf.z, j, ldm, b);
#pragma unroll
      for (int k = 0; k < 6; ++k) vprev[k] = ld(Bf.v, 6 * j + k, ldm, b);
      if (!WITH_H) {
#pragma unroll
        for (int k = 0; k < 6; ++k) UD[k] = ld(Bf.ud, 6 * j + k, ldm, b);
      }
      // FwdPass1 (hxx:304-315): H_i = rho I + H_ref ; p_i = -rho v_prev - Hv
      if (WITH_H) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int cc = r; cc < 6; ++cc) hh[sym(r, cc)] = (r == cc ? P.rho : T(0)) + P.Href[6 * r + cc];
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) pp[k] = -P.rho * vprev[k] - P.Hv[k];
      // constraint terms (hxx:321-334)
      if (d.cslot >= 0) {
        const int cs_ = d.cslot;
        if (WITH_H) {
#pragma unroll
          for (int k = 0; k < 21; ++k) hh[k] += mu_eq * Bf.AtA[(size_t)(cs_ * 21 + k) * a_es + (size_t)b * a_bs];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k)
          pp[k] += ld(Bf.aty, 6 * cs_ + k, ldm, b) - mu_eq * ld(Bf.Atb, 6 * cs_ + k, ldm, b);
      }
      // children contributions accumulated so far (hxx:66-67, :74-75)
      if (!(d.flags & JF_LEAF)) {
        if (WITH_H) {
#pragma unroll
          for (int k = 0; k < 21; ++k) hh[k] += accH[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) pp[k] += accp[k];
      }
      if (WITH_H) {
#pragma unroll
        for (int k = 0; k < 21; ++k) st(Bf.H, 21 * j + k, ldm, b, hh[k]);
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) st(Bf.p, 6 * j + k, ldm, b, pp[k]);

      // calc_aba (hxx:60-63): U = H S ; Dinv = 1/(S^T U + R) ; UDinv = U Dinv
      const T ax0 = (T)d.axis[0], ax1 = (T)d.axis[1], ax2 = (T)d.axis[2];
      T Stp;
      if (d.flags & JF_REVOLUTE) {
        if (WITH_H) {
#pragma unroll
          for (int k = 0; k < 6; ++k) U[k] = hh[sym(k, 3)] * ax0 + hh[sym(k, 4)] * ax1 + hh[sym(k, 5)] * ax2;
          const T dd = T(1) / ((ax0 * U[3] + ax1 * U[4] + ax2 * U[5]) + mu_in);
#pragma unroll
          for (int k = 0; k < 6; ++k) UD[k] = U[k] * dd;
          st(Bf.dinv, j, ldm, b, dd);
        }
        Stp = ax0 * pp[3] + ax1 * pp[4] + ax2 * pp[5];
      } else {
        if (WITH_H) {
#pragma unroll
          for (int k = 0; k < 6; ++k) U[k] = hh[sym(k, 0)] * ax0 + hh[sym(k, 1)] * ax1 + hh[sym(k, 2)] * ax2;
          const T dd = T(1) / ((ax0 * U[0] + ax1 * U[1] + ax2 * U[2]) + mu_in);
#pragma unroll
          for (int k = 0; k < 6; ++k) UD[k] = U[k] * dd;
          st(Bf.dinv, j, ldm, b, dd);
        }
        Stp = ax0 * pp[0] + ax1 * pp[1] + ax2 * pp[2];
      }
      if (WITH_H) {
#pragma unroll
        for (int k = 0; k < 6; ++k) st(Bf.ud, 6 * j + k, ldm, b, UD[k]);
      }
      // r_i = (w_i - mu_in z_i) + S^T p_i   (hxx:296, :70)
      const T ri = (wi - mu_in * zi) + Stp;
      st(Bf.rr, j, ldm, b, ri);

      if (!(d.flags & JF_PARENT_ROOT)) {
        T R[9], t[3], part[27], pa[6];
        make_liMi(d, c, s, R, t);
        if (WITH_H) {
          // H_aba = H - UDinv U^T (hxx:60-63), then SE3actOn (hxx:66)
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int cc = r; cc < 6; ++cc) hh[sym(r, cc)] -= UD[r] * U[cc];
          congr_sym(R, t, hh, part);
        }
        // p_aba = p - UDinv r (hxx:71-73), parent += liMi.act(p_aba) (hxx:74)
#pragma unroll
        for (int k = 0; k < 6; ++k) pa[k] = pp[k] - UD[k] * ri;
        act_force(R, t, pa, part + 21);
        if (!(d.flags & JF_LAST_CHILD)) {
          --level;
          if (WITH_H) stack_pop_add(stk, level, NENT, part, 21, lane);
          stack_pop_add(stk + 21 * WAVE, level, NENT, part + 21, 6, lane);
        }
        if (d.flags & JF_NEXT_IS_PARENT) {
          if (WITH_H) {
#pragma unroll
            for (int k = 0; k < 21; ++k) accH[k] = part[k];
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) accp[k] = part[21 + k];
        } else {
          if (WITH_H) stack_push(stk, level, NENT, part, 21, lane);
          stack_push(stk + 21 * WAVE, level, NENT, part + 21, 6, lane);
          ++level;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// root -> leaf sweep: FwdPass2 + BoxProj + DualUpdate
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void sweep_fwd(const Params<T>& P, const Bufs<T>& Bf, const JointDesc* __restrict__ jd,
                                          int b, bool live, T mu_eq, T mu_in, Norms<T>& N)
{
  const int ldm = P.ld;
  const int a_bs = (P.mode & MODE_A_SHARED) ? 0 : 1;
  const int a_es = (P.mode & MODE_A_SHARED) ? 1 : ldm;
  const int bd_bs = (P.mode & MODE_BND_SHARED) ? 0 : 1;
  const int bd_es = (P.mode & MODE_BND_SHARED) ? 1 : ldm;
  T vcur[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) vcur[k] = T(0);

  for (int i = 1; i <= P.nb; ++i) {
    const JointDesc d = jd[i];
    const int j = i - 1;
    if (live) {
      T hh[21], pp[6], UD[6], vprev[6], fold[6], vpar[6], vp[6], vi[6], fi[6], R[9], t[3];
      const T c = ld(Bf.cs, 2 * j, ldm, b), s = ld(Bf.cs, 2 * j + 1, ldm, b);
#pragma unroll
      for (int k = 0; k < 21; ++k) hh[k] = ld(Bf.H, 21 * j + k, ldm, b);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        pp[k] = ld(Bf.p, 6 * j + k, ldm, b);
        UD[k] = ld(Bf.ud, 6 * j + k, ldm, b);
        vprev[k] = ld(Bf.v, 6 * j + k, ldm, b);
        fold[k] = ld(Bf.f, 6 * j + k, ldm, b);
      }
      const T dd = ld(Bf.dinv, j, ldm, b), ri = ld(Bf.rr, j, ldm, b);
      const T wi = ld(Bf.w, j, ldm, b), nuprev = ld(Bf.nu, j, ldm, b), zprev = ld(Bf.z, j, ldm, b);
      const T lbi = Bf.lb[(size_t)j * bd_es + (size_t)b * bd_bs], ubi = Bf.ub[(size_t)j * bd_es + (size_t)b * bd_bs];
      // parent velocity: universe = 0, chain = registers, branch point = re-read (written earlier by this lane)
      if (d.parent == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) vpar[k] = T(0);
      } else if (d.flags & JF_NEXT_IS_PARENT) {
#pragma unroll
        for (int k = 0; k < 6; ++k) vpar[k] = vcur[k];
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) vpar[k] = ld(Bf.v, 6 * (d.parent - 1) + k, ldm, b);
      }
      make_liMi(d, c, s, R, t);
      actinv_motion(R, t, vpar, vp);  // hxx:125
      // nu_i = -UDinv^T v' - Dinv r_i  (hxx:127)
      T udv = UD[0] * vp[0];
#pragma unroll
      for (int k = 1; k < 6; ++k) udv += UD[k] * vp[k];
      const T nui = -udv - dd * ri;
      N.nu_inf = tmax(N.nu_inf, tabs(nui));
      // v_i = v' + S nu_i (hxx:133-134)
      const T ax0 = (T)d.axis[0], ax1 = (T)d.axis[1], ax2 = (T)d.axis[2];
#pragma unroll
      for (int k = 0; k < 6; ++k) vi[k] = vp[k];
      if (d.flags & JF_REVOLUTE) {
        vi[3] += ax0 * nui; vi[4] += ax1 * nui; vi[5] += ax2 * nui;
      } else {
        vi[0] += ax0 * nui; vi[1] += ax1 * nui; vi[2] += ax2 * nui;
      }
      // f_i = H_i v_i + p_i (hxx:139-140), delta_fis (hxx:137-146)
      symv(hh, vi, fi);
      T df[6], dv6[6], hrv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        fi[k] += pp[k];
        df[k] = fi[k] - fold[k];
        dv6[k] = vi[k] - vprev[k];
      }
      N.dfis = tmax(N.dfis, inf6(df));
      // Href_v (hxx:149-153)
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        T a = P.Href[6 * r] * vi[0];
#pragma unroll
        for (int k = 1; k < 6; ++k) a += P.Href[6 * r + k] * vi[k];
        hrv[r] = a;
      }
      N.href_v = tmax(N.href_v, inf6(hrv));
      N.dvis = tmax(N.dvis, inf6(dv6));  // hxx:156-158
      N.dnu = tmax(N.dnu, tabs(nui - nuprev));  // hxx:375
      // BoxProj (hxx:388-394)
      const T x = nui + (T(1) / mu_in) * wi;
      const T zi = tmin(ubi, tmax(lbi, x));
      N.dz = tmax(N.dz, tabs(zi - zprev));
      N.pr_slack = tmax(N.pr_slack, tabs(nui - zi));
      // DualUpdate, slack part (hxx:454-458) and the dot products of CheckFeasibility (hxx:587-590)
      const T dwi = mu_in * (nui - zi);
      N.dw = tmax(N.dw, tabs(dwi));
      N.ub_dw_plus += ubi * tmax(dwi, T(0));
      N.lb_dw_minus += lbi * tmin(dwi, T(0));
      st(Bf.w, j, ldm, b, wi + dwi);
      st(Bf.nu, j, ldm, b, nui);
      st(Bf.z, j, ldm, b, zi);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        st(Bf.v, 6 * j + k, ldm, b, vi[k]);
        st(Bf.f, 6 * j + k, ldm, b, fi[k]);
        vcur[k] = vi[k];
      }
      // DualUpdate, task part (hxx:410-451)
      if (d.cslot >= 0) {
        const int cs_ = d.cslot;
        T Av[6], e[6], yy[6], aty[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          T a = T(0);
#pragma unroll
          for (int k = 0; k < 6; ++k) a += Bf.A[(size_t)(cs_ * 36 + 6 * r + k) * a_es + (size_t)b * a_bs] * vi[k];
          Av[r] = a;
        }
        T plus = T(0), minus = T(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const T bk = ld(Bf.b, 6 * cs_ + k, ldm, b);
          e[k] = Av[k] - bk;
          const T dy = mu_eq * e[k];
          yy[k] = ld(Bf.y, 6 * cs_ + k, ldm, b) + dy;
          st(Bf.y, 6 * cs_ + k, ldm, b, yy[k]);
          N.dyis = tmax(N.dyis, tabs(dy));
          plus += bk * tmax(dy, T(0));
          minus += bk * tmin(dy, T(0));
        }
        N.bTdy_plus += plus;
        N.bTdy_minus += minus;
        N.pr_task = tmax(N.pr_task, inf6(e));
        N.av_inf = tmax(N.av_inf, inf6(Av));
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          T a = T(0);
#pragma unroll
          for (int k = 0; k < 6; ++k) a += Bf.A[(size_t)(cs_ * 36 + 6 * k + r) * a_es + (size_t)b * a_bs] * yy[k];
          aty[r] = a;
          st(Bf.aty, 6 * cs_ + r, ldm, b, a);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// leaf -> root residual sweep: BwdPass2 + dual residual
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void sweep_bwd2(const Params<T>& P, const Bufs<T>& Bf, const JointDesc* __restrict__ jd,
                                           T* stk, int b, int lane, bool live, Norms<T>& N)
{
  constexpr int NENT = 27;
  const int ldm = P.ld;
  T acc[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[k] = T(0);
  int level = 0;
  for (int i = P.nb; i >= 1; --i) {
    const JointDesc d = jd[i];
    const int j = i - 1;
    if (live) {
      T fi[6], vi[6], gold[6], gi[6];
      const T c = ld(Bf.cs, 2 * j, ldm, b), s = ld(Bf.cs, 2 * j + 1, ldm, b);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        fi[k] = ld(Bf.f, 6 * j + k, ldm, b);
        vi[k] = ld(Bf.v, 6 * j + k, ldm, b);
        gold[k] = ld(Bf.g, 6 * j + k, ldm, b);
      }
      const T wi = ld(Bf.w, j, ldm, b), sold = ld(Bf.s, j, ldm, b);
      // g_i = (Aty_c | 0) + sum_children act(f_j) - f_i   (hxx:438-439, :210-212)
      if (d.cslot >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) gi[k] = ld(Bf.aty, 6 * d.cslot + k, ldm, b);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) gi[k] = T(0);
      }
      if (!(d.flags & JF_LEAF)) {
#pragma unroll
        for (int k = 0; k < 6; ++k) gi[k] += acc[k];
      }
      T dg[6], dvr[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        gi[k] += -fi[k];
        dg[k] = gi[k] - gold[k];
        st(Bf.g, 6 * j + k, ldm, b, gi[k]);
      }
      N.dg = tmax(N.dg, inf6(dg));      // hxx:215-220
      N.g_inf = tmax(N.g_inf, inf6(gi));  // hxx:223-225
      // dual residual, v block (hxx:228): Href v_i - Hv + g_i
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        T a = P.Href[6 * r] * vi[0];
#pragma unroll
        for (int k = 1; k < 6; ++k) a += P.Href[6 * r + k] * vi[k];
        dvr[r] = a - P.Hv[r] + gi[r];
      }
      N.dual_v = tmax(N.dual_v, inf6(dvr));
      // Stf_plus_w (hxx:231-236, :482-484)
      const T ax0 = (T)d.axis[0], ax1 = (T)d.axis[1], ax2 = (T)d.axis[2];
      T stf;
      if (d.flags & JF_REVOLUTE) stf = ax0 * fi[3] + ax1 * fi[4] + ax2 * fi[5];
      else stf = ax0 * fi[0] + ax1 * fi[1] + ax2 * fi[2];
      const T si = stf + wi;
      st(Bf.s, j, ldm, b, si);
      N.stf_w_inf = tmax(N.stf_w_inf, tabs(si));
      N.dstf_w = tmax(N.dstf_w, tabs(si - sold));
      if (!(d.flags & JF_PARENT_ROOT)) {
        T R[9], t[3], part[6];
        make_liMi(d, c, s, R, t);
        act_force(R, t, fi, part);  // hxx:212
        if (!(d.flags & JF_LAST_CHILD)) {
          --level;
          stack_pop_add(stk, level, NENT, part, 6, lane);
        }
        if (d.flags & JF_NEXT_IS_PARENT) {
#pragma unroll
          for (int k = 0; k < 6; ++k) acc[k] = part[k];
        } else {
          stack_push(stk, level, NENT, part, 6, lane);
          ++level;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// persistent solve kernel: each wavefront iterates its 64 instances until all are done (or the launch
// iteration budget is spent).  No inter-wavefront communication: instances are independent.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(WAVE)
k_solve(const Params<T> P, const Bufs<T> Bf, const JointDesc* __restrict__ jd)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* stk = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  const int b = blockIdx.x * WAVE + lane;
  const bool inb = b < P.B;
  const int bb = inb ? b : 0;

  int status = inb ? Bf.status[bb] : ST_DONE;
  int iter = inb ? Bf.iter[bb] : 0;
  T mu = inb ? Bf.mu[bb] : P.mu0;
  T mu_h = inb ? Bf.mu_h[bb] : T(-1);
  const T bnorm = inb ? Bf.bnorm[bb] : T(0);
  bool live = inb && !(status & ST_DONE);
  // main-loop bound `for (i = 1; i < max_iter; ++i)` (hpp:377): nothing to do when max_iter <= 1
  if (live && !(status & ST_TAIL) && iter + 1 >= P.max_iter) { live = false; status |= ST_DONE; }
  unsigned int my_iters = 0;

  for (int k = 0; k < P.max_launch_iters; ++k) {
    if (!__any(live)) break;
    const T mu_eq = P.mu_scale * mu;  // hpp:183-184, hxx:620-621
    const T mu_in = mu;
    Norms<T> N;
    N.reset();
    if (live) { ++iter; ++my_iters; }

    const bool need_h = !(P.mode & MODE_CACHE_H) || __any(live && (mu_h != mu));
    if (need_h) {
      sweep_bwd<T, true>(P, Bf, jd, stk, bb, lane, live, mu_eq, mu_in);
      if (live) mu_h = mu;
    } else {
      sweep_bwd<T, false>(P, Bf, jd, stk, bb, lane, live, mu_eq, mu_in);
    }
    sweep_fwd<T>(P, Bf, jd, bb, live, mu_eq, mu_in, N);
    sweep_bwd2<T>(P, Bf, jd, stk, bb, lane, live, N);

    if (live) {
      // ComputePrimalResiduals / ComputeDualResiduals (hxx:494-522)
      const T primal = tmax(N.pr_task, N.pr_slack);
      const T dual = tmax(N.dual_v, N.stf_w_inf);
      T tol_p = ld(Bf.scal, SC_TOL_PRIMAL, P.ld, bb), tol_d = ld(Bf.scal, SC_TOL_DUAL, P.ld, bb);
      T dx = tmax(N.dvis, N.dnu);
      T dyqp = T(0), atdy = T(0), ubp = T(0), lbm = T(0);
      int c1 = 0, c2 = 0;
      int tail_iter = 0;
      bool ran_feas = false;
      if (P.mode & MODE_FIXED_ITERS) {
        if (iter + 1 >= P.max_iter) { status |= ST_DONE; live = false; }
      } else if (!(status & ST_TAIL)) {
        // CheckConvergence (hxx:544-555): nu_inf_norm appears twice, as upstream
        tol_p = P.tol_abs + P.tol_rel * tmax(tmax(N.av_inf, N.nu_inf), tmax(bnorm, N.nu_inf));
        tol_d = P.tol_abs + P.tol_rel * tmax(tmax(N.href_v, tmax(N.g_inf, N.stf_w_inf)), P.Hv_inf_norm);
        const bool conv = (primal < tol_p) && (dual < tol_d);
        bool infeas = false;
        if (iter > 1) {
          // CheckFeasibility (hxx:576-602)
          dyqp = tmax(N.dfis, tmax(N.dyis, N.dw));
          atdy = tmax(N.dg, N.dstf_w);
          c1 = atdy <= P.tol_primal_inf * dyqp;
          ubp = N.bTdy_plus + N.ub_dw_plus;
          lbm = N.bTdy_minus + N.lb_dw_minus;
          c2 = (ubp + lbm) <= P.tol_primal_inf * dyqp;
          infeas = c1 && c2;
          ran_feas = true;
        }
        if (conv) {
          status |= ST_CONVERGED | ST_DONE;
          if (infeas) status |= ST_PRIMAL_INF;  // flag is set before the `converged_` test (hpp:425-431)
          live = false;
        } else if (infeas) {
          // InfeasibilityTailSolve entry (hpp:271-285)
          status |= ST_PRIMAL_INF | ST_TAIL;
          if (!(dx >= P.tol_tail_solve || N.dz >= P.tol_tail_solve) || iter >= P.max_iter) {
            status |= ST_DONE;
            live = false;
          }
        } else {
          // UpdateMu (hxx:617-631)
          if (primal > T(10) * dual) mu *= T(10);
          else if (dual > T(10) * primal) mu *= T(0.1);
          if (iter + 1 >= P.max_iter) { status |= ST_DONE; live = false; }
        }
      } else {
        // tail-solve iteration (hpp:286-308)
        tail_iter = (int)ld(Bf.scal, SC_TAIL_ITER, P.ld, bb) + 1;
        if (!(dx >= P.tol_tail_solve || N.dz >= P.tol_tail_solve) || iter >= P.max_iter) {
          status |= ST_DONE;
          live = false;
        }
      }
      T* sc = Bf.scal;
      const int l = P.ld;
      st(sc, SC_PRIMAL_RES, l, bb, primal); st(sc, SC_DUAL_RES, l, bb, dual);
      st(sc, SC_PRIMAL_RES_TASK, l, bb, N.pr_task); st(sc, SC_PRIMAL_RES_SLACK, l, bb, N.pr_slack);
      st(sc, SC_DUAL_RES_V, l, bb, N.dual_v); st(sc, SC_DUAL_RES_NU, l, bb, N.stf_w_inf);
      st(sc, SC_TOL_PRIMAL, l, bb, tol_p); st(sc, SC_TOL_DUAL, l, bb, tol_d);
      st(sc, SC_MU, l, bb, mu); st(sc, SC_MU_EQ, l, bb, P.mu_scale * mu); st(sc, SC_MU_INEQ, l, bb, mu);
      st(sc, SC_DELTA_X_QP, l, bb, dx); st(sc, SC_DELTA_Z_QP, l, bb, N.dz);
      if (ran_feas) {
        st(sc, SC_DELTA_Y_QP, l, bb, dyqp); st(sc, SC_AT_DELTA_Y_QP, l, bb, atdy);
        st(sc, SC_UB_DY_PLUS, l, bb, ubp); st(sc, SC_LB_DY_MINUS, l, bb, lbm);
        st(sc, SC_COND1, l, bb, (T)c1); st(sc, SC_COND2, l, bb, (T)c2);
      }
      st(sc, SC_DELTA_FIS, l, bb, N.dfis); st(sc, SC_DELTA_YIS, l, bb, N.dyis); st(sc, SC_DELTA_W, l, bb, N.dw);
      st(sc, SC_DELTA_VIS, l, bb, N.dvis); st(sc, SC_DELTA_NU, l, bb, N.dnu);
      st(sc, SC_AV_INF, l, bb, N.av_inf); st(sc, SC_NU_INF, l, bb, N.nu_inf); st(sc, SC_HREF_V_INF, l, bb, N.href_v);
      st(sc, SC_G_INF, l, bb, N.g_inf); st(sc, SC_STF_PLUS_W_INF, l, bb, N.stf_w_inf);
      if (status & ST_TAIL) st(sc, SC_TAIL_ITER, l, bb, (T)tail_iter);
    }
  }

  if (inb) {
    Bf.status[bb] = status;
    Bf.iter[bb] = iter;
    Bf.mu[bb] = mu;
    Bf.mu_h[bb] = mu_h;
  }
  const unsigned long long live_mask = __ballot(live);
  unsigned int it_sum = my_iters;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) it_sum += __shfl_down(it_sum, off);
  if (lane == 0) {
    const unsigned int nlive = __popcll(live_mask);
    Bf.wave_live[blockIdx.x] = (int)nlive;
    if (nlive) atomicAdd(&Bf.counters[0], nlive);
    if (it_sum) atomicAdd(&Bf.counters[1], it_sum);
  }
}

// ------------------------------------------------------------------------------------------------
// FwdPassInit (hxx:253-283): joint configuration -> per-joint (cos q, sin q) | (q, 0).  liMi itself is
// never stored: every sweep rebuilds R = Rp*Rot(q), t from these two scalars + uniform constants.
// q is instance-major [B][nq] (the caller's layout).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_fk_init(const double* __restrict__ q, int nq, const JointDesc* __restrict__ jd,
                          const int* __restrict__ idx_q, int nb, int B, int ldm, T* __restrict__ cs)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int i = 1; i <= nb; ++i) {
    const double qi = q[(size_t)b * nq + idx_q[i]];
    T c, s;
    if (jd[i].flags & JF_REVOLUTE) {
      double sd, cd;
      sincos(qi, &sd, &cd);
      c = (T)cd; s = (T)sd;
    } else {
      c = (T)qi; s = T(0);
    }
    cs[(size_t)(2 * (i - 1)) * ldm + b] = c;
    cs[(size_t)(2 * (i - 1) + 1) * ldm + b] = s;
  }
}

// instance-major [B][n] (double, caller layout) -> SoA [n][ld] (T)
template <typename T>
__global__ void k_aos_to_soa(const double* __restrict__ src, int n, int B, int ldm, T* __restrict__ dst)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int k = 0; k < n; ++k) dst[(size_t)k * ldm + b] = (T)src[(size_t)b * n + k];
}

// SoA [n][ld] (T) -> instance-major [B][n] (double)
template <typename T>
__global__ void k_soa_to_aos(const T* __restrict__ src, int n, int B, int ldm, double* __restrict__ dst)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int k = 0; k < n; ++k) dst[(size_t)b * n + k] = (double)src[(size_t)k * ldm + b];
}

// per-instance constraint products: AtA (packed), Atb, bis_inf_norm (ik-id-description-optimized.hpp:160-170,
// :210-215).  A is [nc][36] shared or [nc][36][ld]; b is [nc][6][ld].  grow_only: single-constraint update
// only ever grows bis_inf_norm_ (hpp:213-215).
template <typename T>
__global__ void k_constraint_products(const T* __restrict__ A, const T* __restrict__ bvec, int nc, int c_lo,
                                      int c_hi, int a_shared, int B, int ldm, T* __restrict__ AtA,
                                      T* __restrict__ Atb, T* __restrict__ bnorm, int grow_only)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const size_t a_es = a_shared ? 1 : ldm, a_bs = a_shared ? 0 : 1;
  T bn = grow_only ? bnorm[b] : T(0);
  for (int c = c_lo; c < c_hi; ++c) {
    T Al[36], bl[6];
    for (int k = 0; k < 36; ++k) Al[k] = A[(size_t)(c * 36 + k) * a_es + (size_t)b * a_bs];
    for (int k = 0; k < 6; ++k) {
      bl[k] = bvec[(size_t)(6 * c + k) * ldm + b];
      bn = tmax(bn, tabs(bl[k]));
    }
    if (!a_shared) {
      for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) {
          T a = T(0);
          for (int k = 0; k < 6; ++k) a += Al[6 * k + i] * Al[6 * k + j];
          AtA[(size_t)(c * 21 + sym(i, j)) * ldm + b] = a;
        }
    }
    for (int i = 0; i < 6; ++i) {
      T a = T(0);
      for (int k = 0; k < 6; ++k) a += Al[6 * k + i] * bl[k];
      Atb[(size_t)(6 * c + i) * ldm + b] = a;
    }
  }
  bnorm[b] = bn;
}

template <typename T>
__global__ void k_fill(T* __restrict__ p, size_t n, T val)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = val;
}

__global__ void k_fill_int(int* __restrict__ p, size_t n, int val)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = val;
}

// ------------------------------------------------------------------------------------------------
// lane compaction: physical repack of the live instances of one buffer set into the first slots of another
// (dense wavefronts again), and return of the finished ones to their home slot.  One source wavefront per
// workgroup; `wave_off[w]` = exclusive prefix sum of the live-lane counts (host-side scan of `wave_live`).
// A field is a run of `rows` SoA rows of `esz`-byte elements.
// ------------------------------------------------------------------------------------------------
struct MoveField {
  const void* src;
  void* dst_live;   // destination set (compacted slots)
  void* dst_home;   // home set (slot = instance id), or nullptr when the source IS the home set
  int rows;
  int esz;          // 4 or 8
};
constexpr int MAX_MOVE_FIELDS = 32;
struct MovePlan {
  MoveField f[MAX_MOVE_FIELDS];
  int nfields;
  int n_src;        // slots in use in the source set
  int ld_src, ld_dst, ld_home;
  const int* status;     // source status
  const int* map_src;    // slot -> instance id in the source set (nullptr: identity, source is home)
  int* map_dst;          // slot -> instance id in the destination set
  const int* wave_off;   // [n_src/64]
  int force_home;        // 1: every slot of the source set goes home (end of the solve)
};

__global__ void __launch_bounds__(WAVE) k_move(const MovePlan P)
{
  const int lane = threadIdx.x;
  const int b = blockIdx.x * WAVE + lane;
  const bool inb = b < P.n_src;
  const bool live = inb && !P.force_home && !(P.status[inb ? b : 0] & ST_DONE);
  const unsigned long long mask = __ballot(live);
  const int rank = __popcll(mask & ((1ull << lane) - 1ull));
  const int inst = inb ? (P.map_src ? P.map_src[b] : b) : 0;
  const int dst = P.wave_off[blockIdx.x] + rank;
  const bool to_home = inb && !live && P.map_src != nullptr;
  if (live) P.map_dst[dst] = inst;
  if (!live && !to_home) return;
  for (int k = 0; k < P.nfields; ++k) {
    const MoveField F = P.f[k];
    if (F.esz == 8) {
      const unsigned long long* s = (const unsigned long long*)F.src;
      unsigned long long* d = live ? (unsigned long long*)F.dst_live : (unsigned long long*)F.dst_home;
      const size_t ldd = live ? P.ld_dst : P.ld_home;
      const size_t slot = live ? dst : inst;
      for (int r = 0; r < F.rows; ++r) d[(size_t)r * ldd + slot] = s[(size_t)r * P.ld_src + b];
    } else {
      const unsigned int* s = (const unsigned int*)F.src;
      unsigned int* d = live ? (unsigned int*)F.dst_live : (unsigned int*)F.dst_home;
      const size_t ldd = live ? P.ld_dst : P.ld_home;
      const size_t slot = live ? dst : inst;
      for (int r = 0; r < F.rows; ++r) d[(size_t)r * ldd + slot] = s[(size_t)r * P.ld_src + b];
    }
  }
}

}  // namespace loikb
