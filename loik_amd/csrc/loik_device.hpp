// loik_device.hpp -- CDNA4 (gfx950) device code of the batched LoIK ADMM solve.
//
// Execution model: one problem instance per lane, 64-lane wavefronts, one wavefront per workgroup, and the
// wavefront iterates ITS 64 instances to completion (persistent loop, no inter-wavefront communication --
// instances are independent).
//
// Memory layout: wavefront-tiled AoSoA.  All state of the 64 instances a wavefront owns is ONE contiguous
// "tile" in HBM.  A tile is an array of "pairs"; a pair is 64 lanes x 2 scalars, lane-interleaved
// ([lane][2]), so one 16-byte load per lane (dwordx4 in fp64) moves a contiguous 1-KiB run per wavefront
// and every access of a sweep is `tile base + compile-time pair offset` (no per-access address arithmetic).
// Inside a tile the data is grouped per joint ("joint record", JREC pairs), then per constraint, then the
// per-instance solver scalars.  The kinematic tree (parents[], joint type/axis, jointPlacements) is baked on
// the host into a per-joint `JointDesc` schedule, uniform across lanes (scalar loads, SGPR operands).
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

// ---- tile layout (units: pairs) -------------------------------------------------------------------
enum : int {
  // joint record
  JP_CS = 0,     // (cos q, sin q) | (q, 0) for prismatic joints           -- constant per configuration
  JP_V = 1,      // vis[i]                      3 pairs                     -- persistent ADMM state
  JP_F = 4,      // fis[i]                      3 pairs
  JP_G = 7,      // fis_diff_plus_Aty[i]        3 pairs
  JP_WZ = 10,    // (w_i, z_i)
  JP_NUS = 11,   // (nu_i, Stf_plus_w_i)
  JP_P = 12,     // p_i^base = -rho v_i^prev - Hv (+ A^T y - mu_eq A^T b): pis[i] before the children's contributions
                 //                             3 pairs                     -- inter-sweep temporaries
  JP_R = 15,     // (r_i after += S^T p, Dinv_i)   -- always written as a full 16-byte pair
  JP_LBUB = 16,  // (lb_i, ub_i) when the box is per instance
  JP_UD = 17,    // UDinv_i                     3 pairs  -- with Dinv_i the H cache: valid while mu is unchanged
  // H_i itself is NOT stored.  Upstream keeps His[i] for the forward pass' f_i = H_i v_i + p_i (hxx:139-140); here
  // f_i comes from the force-balance recursion f_i = H_i^base v_i + p_i^base + sum_children X*_c f_c (identical in
  // exact arithmetic, see sweep_bwd2), which needs neither H_i nor p_i -- H_i lives only in registers / LDS edge
  // slots during the H recursion.  The getters for His / pis rebuild them on demand (k_rebuild_his / k_rebuild_pis).
  // (Per-lane multi-slot caches keyed by the decade of mu were measured and rejected: lanes of a tile then sit in
  // different slots, every access touches partially used 1-KiB rows: 61 -> 71 ms/step.)
  JREC = 20,
  JP_NPERSIST = 12,  // pairs [0, JP_NPERSIST) (+ JP_LBUB) travel with an instance on compaction
  // constraint record
  CP_Y = 0,      // yis[c]     3 pairs
  CP_ATY = 3,    // Aty[c]     3 pairs
  CP_B = 6,      // bis_[c]    3 pairs
  CP_ATB = 9,    // Atb[c]     3 pairs
  CREC_SHARED_A = 12,
  CP_A = 12,     // Ais_[c] row-major 36 -> 18 pairs      (only when A is per instance)
  CP_ATA = 30,   // AtA[c] packed 21 + pad -> 11 pairs
  CREC_FULL = 42,
  // per-instance solver scalars
  SP_MU = 0,     // (mu_, k) with mu_ = mu0 * 10^k
  SP_BI = 1,     // (bis_inf_norm_, iter_)
  SP_ST = 2,     // (status bits, mu the LAST executed iteration used = the mu of the stored UDinv/Dinv/r/p^base)
  SP_TAG = 3,    // (mu the cached UDinv/Dinv were computed with; -1 = empty, unused)
  SP_FLIP = 4,   // (number of mu updates of this solve so far, unused): the stragglers are the instances that keep
                 // flipping mu -- the tail kernel's queue serves them first
  SP_SCAL = 5,   // scal[NSCAL] -> 15 pairs
  SREC = 20,
};

struct Layout {
  int nb, nc;
  int crec;        // CREC_SHARED_A or CREC_FULL
  int off_c;       // first constraint record  = nb * JREC
  int off_s;       // scalar record            = off_c + nc * crec
  int tile_pairs;  // off_s + SREC
};

// joint flags (uniform per joint)
enum : int {
  JF_PARENT_ROOT = 2,     // parent is the universe: contribution discarded (update_I = parent > 0, hxx:63)
  JF_MASSLESS = 4,        // intermediate link of the chain that stands for a multi-DoF joint: no rho I + H_ref, no
                          // reference term, not counted in the norms over the links (see build_schedule)
  JF_NOQ = 8,             // chain joint after the first: its transform is the identity whatever its JP_CS pair holds
  JF_REVOLUTE = 16,       // S = [0; axis], else prismatic S = [axis; 0]
  JF_CS_DIRECT = 32,      // unbounded revolute joint (JointModelRUBX/Y/Z): its configuration IS (cos, sin)
  JF_HELICAL = 64,        // JointModelHelicalX/Y/Z/Unaligned (with JF_REVOLUTE): M(q) = (Rot(axis, q), pitch q axis), S = [pitch axis;
                          // axis].  Its JP_CS pair holds (q, 0) -- the translation needs the angle itself -- and every formula
                          // with S gets the linear term next to the revolute joint's angular one (scalar branches in k_solve,
                          // the 6-vector S of the lane-per-joint engines)
};

// rotation generator selector for M(q).  ROT_FREE / ROT_SPH / ROT_TRANS: first joint of the chain of a free-flyer /
// spherical / translation joint -- M(q) = (R(quat), t) comes from the JP_CS pairs of this and the next chain records:
//   free-flyer (tx,ty) (tz,qx) (qy,qz) (qw,-) ; spherical (qx,qy) (qz,qw) ; translation (tx,ty) (tz,-)
//   planar (x, y) (cos, sin)
enum : int { ROT_X = 0, ROT_Y = 1, ROT_Z = 2, ROT_U = 3, ROT_NONE = 4, ROT_FREE = 5, ROT_SPH = 6, ROT_TRANS = 7, ROT_PLANAR = 8 };

struct JointDesc {
  double Rp[9];   // jointPlacements[i].rotation(), row-major
  double tp[3];   // jointPlacements[i].translation()
  double axis[3]; // joint axis in the joint frame
  int parent;
  int flags;
  int cslot;      // index into the active constraint list, or -1
  int rot;        // ROT_*
  double pitch;   // JF_HELICAL: translation along the axis per radian (JointModelHelical*: S = [pitch a; a]); 0 otherwise
};

// status bits per instance
enum : int { ST_CONVERGED = 1, ST_PRIMAL_INF = 2, ST_TAIL = 4, ST_DONE = 8,
             ST_PFULL = 16 };  // the p slot holds the accumulated p_i (left by k_tail), not p_i^base

// solver mode flags (uniform)
enum : int {
  MODE_FIXED_ITERS = 1,  // no convergence / feasibility / mu logic: exactly max_iter-1 iterations
  MODE_CACHE_H = 2,      // skip the H-recursion of the leaf->root sweep while no live lane changed mu
  MODE_A_SHARED = 4,     // one A per constraint for the whole batch
  MODE_BND_SHARED = 8,   // one lb/ub for the whole batch
  MODE_MU_OSQP = 16,     // UpdateMu: OSQP's rule instead of the DEFAULT decade steps (update_mu below)
  MODE_ZERO_STATE = 32,  // the launch takes every instance straight from a cold reset (vis = fis = g = w = z = 0 in every record:
                         // IkIdData::Reset(false) / ResetRecursion): the flat engine does not fetch those pairs
};


// per-iteration scalars dumped for the getters / parity tests (scalar record, SP_SCAL)
enum : int {
  SC_PRIMAL_RES = 0, SC_DUAL_RES, SC_PRIMAL_RES_TASK, SC_PRIMAL_RES_SLACK, SC_DUAL_RES_V, SC_DUAL_RES_NU,
  SC_TOL_PRIMAL, SC_TOL_DUAL, SC_MU, SC_MU_EQ, SC_MU_INEQ,
  SC_DELTA_X_QP, SC_DELTA_Z_QP, SC_DELTA_Y_QP, SC_AT_DELTA_Y_QP, SC_UB_DY_PLUS, SC_LB_DY_MINUS,
  SC_DELTA_FIS, SC_DELTA_YIS, SC_DELTA_W, SC_DELTA_VIS, SC_DELTA_NU,
  SC_AV_INF, SC_NU_INF, SC_HREF_V_INF, SC_G_INF, SC_STF_PLUS_W_INF,
  SC_COND1, SC_COND2, SC_TAIL_ITER,
  NSCAL
};
static_assert(SP_SCAL + (NSCAL + 1) / 2 <= SREC, "scalar record too small");

template <typename T>
struct Params {
  // uniform problem data (UpdateReference broadcasts ONE H_ref,v_ref: ik-id-description-optimized.hpp:78-97)
  T Href[36];  // full matrix (used as-is for Href*v, hxx:149)
  T Hv[6];     // H_ref * v_ref
  T Hv_inf_norm;
  // UpdateReferences (ik-id-description-optimized.hpp:103-121): one weight and one target per link.  nullptr = the
  // broadcast pair above; else [nj][HREF_ROW] rows (H_ref_i row-major 36, then H_ref_i v_ref_i), row 0 and the rows of
  // massless chain links zero.  The lean kernel does not take it (the engine plan sends such solves to k_tail / k_solve).
  const T* href_tab;
  T rho, mu0, mu_scale;
  T tol_abs, tol_rel, tol_primal_inf, tol_tail_solve;
  int max_iter;
  int mode;
  int B;    // instances (slots) of this launch
  int max_launch_iters;
  Layout L;
};

template <typename T>
struct Bufs {
  char* tiles;             // tile t at tiles + t * L.tile_pairs * PAIR_BYTES
  const T* uni;            // uniform inputs: A[nc][36], AtA[nc][21], lb[nb], ub[nb]
  unsigned int* counters;  // [0] live instances at exit, [1] instance-iterations executed,
                           // [2] tile-iterations, [3] of them with an H rebuild, [4] of them with the fused sweep
  int* wave_live;          // [tiles] live lanes of each wavefront at exit (feeds the host-side compaction scan)
  // SolverInfo lists of a handle created with logging = 1 (k_flat<.., LOG = true>; layout of k_pass_solve, loik_passes.hpp):
  // log[(list * log_B + b) * log_cap + k], k = iteration - 1;  log_rows[b] = rows written
  double* log;
  int* log_rows;
  int log_cap, log_B;
};

template <typename T> struct Vec2;
template <> struct Vec2<double> { using type = double2; };
template <> struct Vec2<float> { using type = float2; };

template <typename T>
__host__ __device__ constexpr size_t pair_bytes() { return (size_t)WAVE * 2 * sizeof(T); }

// pair `p` of the record `rec` (per-lane pointer: record base + lane * 2 * sizeof(T))
template <typename T>
__device__ __forceinline__ typename Vec2<T>::type ldp(const char* rec, int p)
{
  return *reinterpret_cast<const typename Vec2<T>::type*>(rec + (size_t)p * pair_bytes<T>());
}
template <typename T>
__device__ __forceinline__ void stp(char* rec, int p, T x, T y)
{
  typename Vec2<T>::type v;
  v.x = x; v.y = y;
  *reinterpret_cast<typename Vec2<T>::type*>(rec + (size_t)p * pair_bytes<T>()) = v;
}
template <typename T>
__device__ __forceinline__ void st_lo(char* rec, int p, T x) { *reinterpret_cast<T*>(rec + (size_t)p * pair_bytes<T>()) = x; }
template <typename T>
__device__ __forceinline__ void st_hi(char* rec, int p, T x) { *reinterpret_cast<T*>(rec + (size_t)p * pair_bytes<T>() + sizeof(T)) = x; }

template <typename T>
__device__ __forceinline__ void ld6(const char* rec, int p, T* x)
{
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const typename Vec2<T>::type v = ldp<T>(rec, p + k);
    x[2 * k] = v.x; x[2 * k + 1] = v.y;
  }
}
template <typename T>
__device__ __forceinline__ void st6(char* rec, int p, const T* x)
{
#pragma unroll
  for (int k = 0; k < 3; ++k) stp<T>(rec, p + k, x[2 * k], x[2 * k + 1]);
}

// ------------------------------------------------------------------------------------------------
// small fixed-size algebra, everything fully unrolled so all register arrays are statically indexed
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int sym(int i, int j)
{
  // upper-triangular row-major packing of a symmetric 6x6
  return (i <= j) ? (i * 6 - (i * (i - 1)) / 2 + (j - i)) : (j * 6 - (j * (j - 1)) / 2 + (i - j));
}

// |x|, max, min map to v_max_f64 / v_min_f64 with |.| source modifiers (no compare + select chains).
// NaN handling equals the reference's `if (x > norm) norm = x` idiom: a NaN candidate never wins.
__host__ __device__ __forceinline__ double tabs(double x) { return __builtin_fabs(x); }
__host__ __device__ __forceinline__ float tabs(float x) { return __builtin_fabsf(x); }
__host__ __device__ __forceinline__ double tmax(double a, double b) { return __builtin_fmax(a, b); }
__host__ __device__ __forceinline__ float tmax(float a, float b) { return __builtin_fmaxf(a, b); }
__host__ __device__ __forceinline__ double tmin(double a, double b) { return __builtin_fmin(a, b); }
__host__ __device__ __forceinline__ float tmin(float a, float b) { return __builtin_fminf(a, b); }

// UpdateMu (loik-loid-optimized.hxx:613-641).  DEFAULT: the reference's decade steps.  OSQP: declared upstream
// (ADMMPenaltyUpdateStrat::OSQP, task-solver-base.hpp:13-18) but never implemented there (it throws, hxx:632-637); this is
// OSQP's published rule on LoIK's quantities (the CPU checker used by the tests evaluates the same expression):
//   mu <- mu * sqrt( (r_p / max(|Av|, |nu|, |b|)) / (r_d / max(|H_ref v|, |g|, |S^T f + w|, |H_ref v_ref|)) ), clipped to
//   [1e-6, 1e6], applied only when it moves mu by more than a factor of 5.  Returns true when mu changed.
template <typename T>
__device__ __forceinline__ bool update_mu(int mode, T primal, T dual, T norm_p, T norm_d, T& mu, int& kexp)
{
  if (mode & MODE_MU_OSQP) {
    const T rp = primal / (norm_p + T(1e-10)), rd = dual / (norm_d + T(1e-10));
    T mu_new = mu * (T)__builtin_sqrt((double)(rp / (rd + T(1e-10))));
    mu_new = tmin(tmax(mu_new, T(1e-6)), T(1e6));
    if (mu_new > T(5.0) * mu || mu_new < T(0.2) * mu) { mu = mu_new; return true; }
    return false;
  }
  if (primal > T(10) * dual) { mu *= T(10); ++kexp; return true; }
  if (dual > T(10) * primal) { mu *= T(0.1); --kexp; return true; }
  return false;
}

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
// joints `c` carries q.  liMi is never stored: every sweep rebuilds it from (c,s) + uniform constants.
template <typename T>
__device__ __forceinline__ void make_liMi(const JointDesc& d, T c, T s, T* R, T* t)
{
  T Rp[9], tp[3], ax[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rp[k] = (T)d.Rp[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { tp[k] = (T)d.tp[k]; ax[k] = (T)d.axis[k]; }
  if (d.flags & JF_REVOLUTE) {
    // R = Rp * Rot(axis, q).  For the axis-aligned joints only two columns of Rp are mixed (the dense product's
    // other terms are exact zeros / ones).
    if (d.rot == ROT_X) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        R[3 * i] = Rp[3 * i];
        R[3 * i + 1] = Rp[3 * i + 1] * c + Rp[3 * i + 2] * s;
        R[3 * i + 2] = Rp[3 * i + 2] * c - Rp[3 * i + 1] * s;
      }
    } else if (d.rot == ROT_Y) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        R[3 * i] = Rp[3 * i] * c - Rp[3 * i + 2] * s;
        R[3 * i + 1] = Rp[3 * i + 1];
        R[3 * i + 2] = Rp[3 * i] * s + Rp[3 * i + 2] * c;
      }
    } else if (d.rot == ROT_Z) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        R[3 * i] = Rp[3 * i] * c + Rp[3 * i + 1] * s;
        R[3 * i + 1] = Rp[3 * i + 1] * c - Rp[3 * i] * s;
        R[3 * i + 2] = Rp[3 * i + 2];
      }
    } else {  // Rodrigues: c I + (1-c) a a^T + s [a]x
      T M[9];
      const T c1 = T(1) - c;
      T tmp;
      tmp = c1 * ax[0] * ax[1]; M[1] = tmp - s * ax[2]; M[3] = tmp + s * ax[2];
      tmp = c1 * ax[0] * ax[2]; M[2] = tmp + s * ax[1]; M[6] = tmp - s * ax[1];
      tmp = c1 * ax[1] * ax[2]; M[5] = tmp - s * ax[0]; M[7] = tmp + s * ax[0];
      M[0] = c1 * ax[0] * ax[0] + c; M[4] = c1 * ax[1] * ax[1] + c; M[8] = c1 * ax[2] * ax[2] + c;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          R[3 * i + j] = Rp[3 * i] * M[j] + Rp[3 * i + 1] * M[3 + j] + Rp[3 * i + 2] * M[6 + j];
    }
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

// Eigen::Quaternion::toRotationMatrix, coefficients (x, y, z, w) as Pinocchio stores them in q
template <typename T>
__device__ __forceinline__ void quat_to_rot(T x, T y, T z, T w, T* R)
{
  const T tx = T(2) * x, ty = T(2) * y, tz = T(2) * z;
  const T twx = tx * w, twy = ty * w, twz = tz * w;
  const T txx = tx * x, txy = ty * x, txz = tz * x;
  const T tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = T(1) - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = T(1) - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = T(1) - (txx + tyy);
}

// liMi of any joint of the (all-1-DoF) device tree.  1-DoF joints: make_liMi from their own (c,s).  The first joint of
// the chain of a multi-DoF joint carries the whole M(q) = (R(quat), t) of that joint
// (JointModelFreeFlyer/Spherical/Translation::calc), read from the JP_CS pairs of the chain's records; the other
// chain joints are the identity.
template <typename T>
__device__ __forceinline__ void joint_xform(const JointDesc& d, const char* rec, T c, T s, T* R, T* t)
{
  if (__builtin_expect(d.rot >= ROT_FREE, 0)) {
    constexpr size_t RB = (size_t)JREC * pair_bytes<T>();
    T Rq[9], tq[3] = {T(0), T(0), T(0)}, Rp[9], rt[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) { Rp[k] = (T)d.Rp[k]; Rq[k] = (k % 4 == 0) ? T(1) : T(0); }
    const typename Vec2<T>::type c1 = ldp<T>(rec + RB, JP_CS);
    if (d.rot == ROT_FREE) {
      const typename Vec2<T>::type c2 = ldp<T>(rec + 2 * RB, JP_CS), c3 = ldp<T>(rec + 3 * RB, JP_CS);
      tq[0] = c; tq[1] = s; tq[2] = c1.x;
      quat_to_rot(c1.y, c2.x, c2.y, c3.x, Rq);
    } else if (d.rot == ROT_SPH) {
      quat_to_rot(c, s, c1.x, c1.y, Rq);
    } else if (d.rot == ROT_PLANAR) {  // JointModelPlanar::calc: M = (Rz(theta), (x, y, 0)), theta given as (cos, sin)
      tq[0] = c; tq[1] = s;
      Rq[0] = c1.x; Rq[1] = -c1.y; Rq[3] = c1.y; Rq[4] = c1.x;
    } else {
      tq[0] = c; tq[1] = s; tq[2] = c1.x;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[3 * i + j] = Rp[3 * i] * Rq[j] + Rp[3 * i + 1] * Rq[3 + j] + Rp[3 * i + 2] * Rq[6 + j];
    mat3_vec(Rp, tq, rt);
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = (T)d.tp[k] + rt[k];
    return;
  }
  if (__builtin_expect(d.flags & JF_NOQ, 0)) { c = (d.flags & JF_REVOLUTE) ? T(1) : T(0); s = T(0); }
  if (__builtin_expect(d.flags & JF_HELICAL, 0)) {  // (q, 0) in the pair: the rotation by q about the axis, the translation pitch q axis
    const T q = c;
    T ra[3], ax[3], Rp[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rp[k] = (T)d.Rp[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) ax[k] = (T)d.axis[k];
    if constexpr (sizeof(T) == 8) { double sd, cd; sincos((double)q, &sd, &cd); c = (T)cd; s = (T)sd; }
    else { float sf, cf; sincosf((float)q, &sf, &cf); c = (T)cf; s = (T)sf; }
    make_liMi(d, c, s, R, t);
    mat3_vec(Rp, ax, ra);
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] += ((T)d.pitch * q) * ra[k];
    return;
  }
  make_liMi(d, c, s, R, t);
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

constexpr int HREF_ROW = 42;

// Href * v (hxx:149, :228).  HDIAG: H_ref is diagonal (e.g. the identity of every reference test), 6 uniform
// scalars instead of 36 -- keeps the kernel's SGPR budget for the joint descriptor.
template <typename T, bool HDIAG>
__device__ __forceinline__ void href_mul(const T* Href, const T* v, T* o)
{
  if (HDIAG) {
#pragma unroll
    for (int r = 0; r < 6; ++r) o[r] = Href[7 * r] * v[r];
  } else {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      T a = Href[6 * r] * v[0];
#pragma unroll
      for (int k = 1; k < 6; ++k) a += Href[6 * r + k] * v[k];
      o[r] = a;
    }
  }
}

// The cost a link carries by itself: rho I + H_ref and the reference term H_ref v_ref -- all zero for a massless chain
// link (JF_MASSLESS).  The flag is uniform per joint and the parameters are kernel arguments, so these are scalar
// selects on SGPRs: the vector work per joint is the same as without the flag.
template <typename T, bool HDIAG>
struct LinkCost {
  T rho, Hv[6], Href[36];
  __device__ __forceinline__ LinkCost(const Params<T>& P, int jflags, int joint)
  {
    const bool ml = jflags & JF_MASSLESS;
    rho = ml ? T(0) : P.rho;
    if (P.href_tab) {  // per-link references: the joint's row of the table (uniform address; zero for a massless link)
      const T* row = P.href_tab + (size_t)joint * HREF_ROW;
#pragma unroll
      for (int k = 0; k < 6; ++k) Hv[k] = row[36 + k];
#pragma unroll
      for (int k = 0; k < 36; ++k) Href[k] = (HDIAG && (k % 7 != 0)) ? T(0) : row[k];
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) Hv[k] = ml ? T(0) : P.Hv[k];
#pragma unroll
      for (int k = 0; k < 36; ++k) Href[k] = (HDIAG && (k % 7 != 0)) ? T(0) : (ml ? T(0) : P.Href[k]);
    }
  }
};

// the same for the engines with one joint per lane (k_tail): `row` = the lane's row of the table or nullptr
template <typename T, bool HDIAG>
__device__ __forceinline__ void href_mul_link(const Params<T>& P, const T* row, const T* v, T* o)
{
  if (row) {
    if (HDIAG) {
#pragma unroll
      for (int r = 0; r < 6; ++r) o[r] = row[7 * r] * v[r];
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        T a = row[6 * r] * v[0];
#pragma unroll
        for (int k = 1; k < 6; ++k) a += row[6 * r + k] * v[k];
        o[r] = a;
      }
    }
  } else {
    href_mul<T, HDIAG>(P.Href, v, o);
  }
}

// ---- team schedule ---------------------------------------------------------------------------------
// A tile (64 instances, one per lane) is advanced by a TEAM of `nw` wavefronts (one workgroup).  Every wavefront
// keeps the lane = instance mapping, so all tile accesses stay full 1-KiB rows; the joints of a sweep are dealt
// out to the wavefronts by a host-built step schedule (list scheduling over the kinematic tree, see
// build_team_schedule() in loik_host.hip): at step t wavefront w handles joint sched[w][t] (0 = idle), all
// wavefronts meet at a workgroup barrier after every step.  Independent chains of the tree (legs, arms, head of a
// humanoid) therefore run concurrently and a sweep costs the tree's critical path instead of nb joint visits.
// nw = 1 is the plain one-wavefront-per-tile case (no barriers).
// Leaf->root hand-over of a joint's contribution to its parent: in registers when the same wavefront handles the
// parent next (SF_OUT_REG / SF_IN_REG), otherwise through an LDS "edge slot" [slot][entry][lane] (SF_OUT_LDS; the
// parent adds the slots listed in rlist[rstart .. rstart+nread)).
struct StepDesc {
  JointDesc d; // copy of the joint's descriptor (one scalar-load burst per step)
  int joint;   // 1..nb, 0 = idle step
  int flags;   // SF_*
  int wslot;   // edge slot written (SF_OUT_LDS)
  int rstart;  // leaf->root: first entry of this joint's edge-slot list in rlist; root->leaf: slot of the parent's v
  int nread;   // number of edge slots the joint adds
  int pad_;
};
enum : int {
  SF_IN_REG = 1,    // leaf->root: add the register accumulator (the chain child was the wavefront's previous joint)
  SF_OUT_REG = 2,   // leaf->root: keep the contribution in registers (the parent is the wavefront's next joint)
  SF_OUT_LDS = 4,   // leaf->root: write the contribution to edge slot `wslot`; root->leaf: write v to v-slot `wslot`
  SF_VPAR_REG = 8,  // root->leaf: the parent's velocity is still in registers
};
constexpr int EDGE_ENT = 27;  // entries of an edge slot: 21 (H, symmetric) + 6 (p)
constexpr int MAX_TEAM = 4;   // wavefronts per tile (workgroup of up to 256 lanes: one wavefront per SIMD of a CU)

// The three tables are separate `const __restrict__` kernel arguments on purpose: only then the compiler proves them
// unclobbered by the tile stores and reads them with scalar loads (SMEM) instead of vector loads + readfirstlane.
struct Team {
  const StepDesc* __restrict__ up;    // [nw][T_up]   leaf -> root
  const StepDesc* __restrict__ down;  // [nw][T_down] root -> leaf
  const int* __restrict__ rlist;
  int T_up, T_down;
  int edge_ent;          // entries (x 64 lanes) of the edge-slot area; the v-slots of the root->leaf sweep follow it
};

template <typename T, int N0, int N>
__device__ __forceinline__ void edge_store(T* edge, int slot, const T* x, int lane)
{
#pragma unroll
  for (int k = 0; k < N; ++k) edge[(slot * EDGE_ENT + N0 + k) * WAVE + lane] = x[k];
}
template <typename T, int N0, int N>
__device__ __forceinline__ void edge_add(const T* edge, int slot, T* x, int lane)
{
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] += edge[(slot * EDGE_ENT + N0 + k) * WAVE + lane];
}
// x += contributions of the children of the step's joint (register accumulator first, then the edge slots)
template <typename T, int N0, int N>
__device__ __forceinline__ void edge_gather(const StepDesc& sd, const int* __restrict__ rlist, const T* edge,
                                            const T* acc, T* x, int lane)
{
  if (sd.flags & SF_IN_REG) {
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] += acc[k];
  }
  for (int r = 0; r < sd.nread; ++r) edge_add<T, N0, N>(edge, rlist[sd.rstart + r], x, lane);
}
template <typename T, int N0, int N>
__device__ __forceinline__ void edge_emit(const StepDesc& sd, T* edge, T* acc, const T* part, int lane)
{
  if (sd.flags & SF_OUT_REG) {
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = part[k];
  } else if (sd.flags & SF_OUT_LDS) {
    edge_store<T, N0, N>(edge, sd.wslot, part, lane);
  }
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
  static constexpr int NMAX = 16, NSUM = 4, NALL = NMAX + NSUM;
  __device__ __forceinline__ void pack(T* a) const
  {
    a[0] = nu_inf; a[1] = dfis; a[2] = href_v; a[3] = dvis; a[4] = dnu; a[5] = dz; a[6] = dw; a[7] = dyis;
    a[8] = av_inf; a[9] = pr_task; a[10] = pr_slack; a[11] = g_inf; a[12] = dg; a[13] = stf_w_inf; a[14] = dstf_w;
    a[15] = dual_v;
    a[16] = bTdy_plus; a[17] = bTdy_minus; a[18] = ub_dw_plus; a[19] = lb_dw_minus;
  }
  __device__ __forceinline__ void unpack(const T* a)
  {
    nu_inf = a[0]; dfis = a[1]; href_v = a[2]; dvis = a[3]; dnu = a[4]; dz = a[5]; dw = a[6]; dyis = a[7];
    av_inf = a[8]; pr_task = a[9]; pr_slack = a[10]; g_inf = a[11]; dg = a[12]; stf_w_inf = a[13]; dstf_w = a[14];
    dual_v = a[15];
    bTdy_plus = a[16]; bTdy_minus = a[17]; ub_dw_plus = a[18]; lb_dw_minus = a[19];
  }
  // combine the partial scalars of the `nw` wavefronts of a team (each covers its own joints): every wavefront
  // ends up with bit-identical totals (same order), so all of them take the same control-flow decisions
  __device__ __forceinline__ void team_combine(T* xch, int w, int nw, int lane)
  {
    T a[NALL];
    pack(a);
#pragma unroll
    for (int k = 0; k < NALL; ++k) xch[(w * NALL + k) * WAVE + lane] = a[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NALL; ++k) a[k] = xch[k * WAVE + lane];
    for (int ww = 1; ww < nw; ++ww) {
#pragma unroll
      for (int k = 0; k < NMAX; ++k) a[k] = tmax(a[k], xch[(ww * NALL + k) * WAVE + lane]);
#pragma unroll
      for (int k = NMAX; k < NALL; ++k) a[k] += xch[(ww * NALL + k) * WAVE + lane];
    }
    unpack(a);
    __syncthreads();  // the exchange area aliases the edge slots of the next sweep
  }
};

// barrier between two steps of a sweep: only the LDS hand-overs have to be visible to the other wavefronts of the
// team (every joint record is touched by one wavefront per sweep), so outstanding global loads/stores are NOT
// drained here -- the prefetch of the next step stays in flight across it.
__device__ __forceinline__ void step_barrier(int nw)
{
  if (nw > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// end of a sweep: the next sweep reads records written by other wavefronts of the team -> full workgroup fence
__device__ __forceinline__ void sweep_barrier(int nw)
{
  if (nw > 1) __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// leaf -> root sweep: FwdPass1 + BwdPass.  WITH_H=false re-uses the cached H/UDinv/Dinv (valid while
// mu is unchanged: they depend only on rho, mu, liMi, H_ref, AtA -- never on the iterates).
// `lp` = this lane's pointer into its team's tile.
// ------------------------------------------------------------------------------------------------
template <typename T, bool WITH_H, bool HDIAG>
__device__ __forceinline__ void sweep_bwd(const Params<T>& P, const Bufs<T>& Bf, const Team& tm, int w, int nw,
                                          T* edge, char* lp, int lane, bool live, T mu_eq, T mu_in)
{
  const Layout& L = P.L;
  T accH[21], accp[6];
#pragma unroll
  for (int k = 0; k < 21; ++k) accH[k] = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) accp[k] = T(0);
  const StepDesc* __restrict__ steps = tm.up + (size_t)w * tm.T_up;

  int jn = steps[0].joint;
  for (int t_ = 0; t_ < tm.T_up; ++t_) {
    const StepDesc sd = steps[t_];  // by value: one scalar-load burst, never re-read behind a store
    // the joint index is fetched one step ahead: the record loads below must not wait for a descriptor load
    const int i = jn;
    jn = steps[t_ + 1 < tm.T_up ? t_ + 1 : t_].joint;
    if (i > 0 && live) {
      const JointDesc& d = sd.d;
      char* rec = lp + (size_t)(i - 1) * JREC * pair_bytes<T>();
      T vprev[6], hh[21], pp[6], U[6], UD[6];
      const typename Vec2<T>::type cs = ldp<T>(rec, JP_CS), wz = ldp<T>(rec, JP_WZ);
      ld6<T>(rec, JP_V, vprev);
      T dd_cached = T(0);
      if (!WITH_H) {
        ld6<T>(rec, JP_UD, UD);
        dd_cached = ldp<T>(rec, JP_R).y;  // Dinv shares the pair of r: read it to rewrite the full pair below
      }
      // FwdPass1 (hxx:304-315): H_i = rho I + H_ref ; p_i = -rho v_prev - Hv   (a massless chain link: both zero)
      const LinkCost<T, HDIAG> lc(P, d.flags, i);
      if (WITH_H) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int cc = r; cc < 6; ++cc)
            hh[sym(r, cc)] = (r == cc ? lc.rho : T(0)) + ((HDIAG && r != cc) ? T(0) : lc.Href[6 * r + cc]);
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) pp[k] = -lc.rho * vprev[k] - lc.Hv[k];
      // constraint terms (hxx:321-334)
      if (d.cslot >= 0) {
        const char* crec = lp + (size_t)(L.off_c + d.cslot * L.crec) * pair_bytes<T>();
        if (WITH_H) {
          if (P.mode & MODE_A_SHARED) {
            const T* AtA = Bf.uni + L.nc * 36 + d.cslot * 21;
#pragma unroll
            for (int k = 0; k < 21; ++k) hh[k] += mu_eq * AtA[k];
          } else {
#pragma unroll
            for (int k = 0; k < 11; ++k) {
              const typename Vec2<T>::type a = ldp<T>(crec, CP_ATA + k);
              hh[2 * k] += mu_eq * a.x;
              if (2 * k + 1 < 21) hh[2 * k + 1] += mu_eq * a.y;
            }
          }
        }
        T aty[6], atb[6];
        ld6<T>(crec, CP_ATY, aty);
        ld6<T>(crec, CP_ATB, atb);
#pragma unroll
        for (int k = 0; k < 6; ++k) pp[k] += aty[k] - mu_eq * atb[k];
      }
      // p_i^base is what the residual sweep needs (force-balance recursion for f_i), not the accumulated p_i
      st6<T>(rec, JP_P, pp);
      // children contributions (hxx:66-67, :74-75)
      if (WITH_H) edge_gather<T, 0, 21>(sd, tm.rlist, edge, accH, hh, lane);
      edge_gather<T, 21, 6>(sd, tm.rlist, edge, accp, pp, lane);

      // calc_aba (hxx:60-63): U = H S ; Dinv = 1/(S^T U + R) ; UDinv = U Dinv
      const T ax0 = (T)d.axis[0], ax1 = (T)d.axis[1], ax2 = (T)d.axis[2];
      T Stp, dd = T(0);
      if (d.flags & JF_HELICAL) {  // S = [pitch a; a]
        const T ph = (T)d.pitch;
        if (WITH_H) {
#pragma unroll
          for (int k = 0; k < 6; ++k)
            U[k] = (hh[sym(k, 3)] * ax0 + hh[sym(k, 4)] * ax1 + hh[sym(k, 5)] * ax2) + ph * (hh[sym(k, 0)] * ax0 + hh[sym(k, 1)] * ax1 + hh[sym(k, 2)] * ax2);
          dd = T(1) / (((ax0 * U[3] + ax1 * U[4] + ax2 * U[5]) + ph * (ax0 * U[0] + ax1 * U[1] + ax2 * U[2])) + mu_in);
        }
        Stp = (ax0 * pp[3] + ax1 * pp[4] + ax2 * pp[5]) + ph * (ax0 * pp[0] + ax1 * pp[1] + ax2 * pp[2]);
      } else if (d.flags & JF_REVOLUTE) {
        if (WITH_H) {
#pragma unroll
          for (int k = 0; k < 6; ++k) U[k] = hh[sym(k, 3)] * ax0 + hh[sym(k, 4)] * ax1 + hh[sym(k, 5)] * ax2;
          dd = T(1) / ((ax0 * U[3] + ax1 * U[4] + ax2 * U[5]) + mu_in);
        }
        Stp = ax0 * pp[3] + ax1 * pp[4] + ax2 * pp[5];
      } else {
        if (WITH_H) {
#pragma unroll
          for (int k = 0; k < 6; ++k) U[k] = hh[sym(k, 0)] * ax0 + hh[sym(k, 1)] * ax1 + hh[sym(k, 2)] * ax2;
          dd = T(1) / ((ax0 * U[0] + ax1 * U[1] + ax2 * U[2]) + mu_in);
        }
        Stp = ax0 * pp[0] + ax1 * pp[1] + ax2 * pp[2];
      }
      // r_i = (w_i - mu_in z_i) + S^T p_i   (hxx:296, :70)
      const T ri = (wz.x - mu_in * wz.y) + Stp;
      if (WITH_H) {
#pragma unroll
        for (int k = 0; k < 6; ++k) UD[k] = U[k] * dd;
        st6<T>(rec, JP_UD, UD);  // UDinv and Dinv are all the forward sweep needs of the H recursion
      }
      stp<T>(rec, JP_R, ri, WITH_H ? dd : dd_cached);

      if (!(d.flags & JF_PARENT_ROOT)) {
        T R[9], t[3], part[27], pa[6];
        joint_xform<T>(d, rec, cs.x, cs.y, R, t);
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
        if (WITH_H) edge_emit<T, 0, 21>(sd, edge, accH, part, lane);
        edge_emit<T, 21, 6>(sd, edge, accp, part + 21, lane);
      }
    }
    step_barrier(nw);
  }
  sweep_barrier(nw);
}

// ------------------------------------------------------------------------------------------------
// root -> leaf sweep: FwdPass2 + BoxProj + DualUpdate.  The record of the NEXT step's joint is fetched
// while the current joint is computed (one step of software prefetch: a wavefront runs alone on its SIMD,
// nothing else hides the HBM latency of a step).
// ------------------------------------------------------------------------------------------------
template <typename T>
struct FwdIn {
  typename Vec2<T>::type cs, wz, nus, rd, lu;
  T UD[6], vprev[6];
  __device__ __forceinline__ void load(const char* rec, bool bnd_shared)
  {
    cs = ldp<T>(rec, JP_CS); wz = ldp<T>(rec, JP_WZ); nus = ldp<T>(rec, JP_NUS); rd = ldp<T>(rec, JP_R);
    ld6<T>(rec, JP_UD, UD);
    ld6<T>(rec, JP_V, vprev);
    if (!bnd_shared) lu = ldp<T>(rec, JP_LBUB);
  }
};

template <typename T, bool HDIAG, bool PF>
__device__ __forceinline__ void sweep_fwd(const Params<T>& P, const Bufs<T>& Bf, const Team& tm, int w, int nw,
                                          T* vedge, char* lp, int lane, bool live, T mu_eq, T mu_in, Norms<T>& N)
{
  const Layout& L = P.L;
  const bool bnd_shared = P.mode & MODE_BND_SHARED;
  T vcur[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) vcur[k] = T(0);
  const StepDesc* __restrict__ steps = tm.down + (size_t)w * tm.T_down;
  FwdIn<T> in;
  if (PF) {
    const int i0 = steps[0].joint;
    if (i0 > 0 && live) in.load(lp + (size_t)(i0 - 1) * JREC * pair_bytes<T>(), bnd_shared);
  }

  int jn = steps[0].joint;
  for (int t_ = 0; t_ < tm.T_down; ++t_) {
    const StepDesc sd = steps[t_];  // by value: one scalar-load burst, never re-read behind a store
    // the joint index is fetched one step ahead: the record loads below must not wait for a descriptor load
    const int i = jn;
    jn = steps[t_ + 1 < tm.T_down ? t_ + 1 : t_].joint;
    if (i > 0 && live) {
      const JointDesc& d = sd.d;
      char* rec = lp + (size_t)(i - 1) * JREC * pair_bytes<T>();
      if (!PF) in.load(rec, bnd_shared);
      T vpar[6], vp[6], vi[6], R[9], t[3];
      T lbi, ubi;
      if (bnd_shared) {
        lbi = Bf.uni[L.nc * 57 + (i - 1)];
        ubi = Bf.uni[L.nc * 57 + L.nb + (i - 1)];
      } else {
        lbi = in.lu.x; ubi = in.lu.y;
      }
      const T ri = in.rd.x, dd = in.rd.y, wi = in.wz.x, zprev = in.wz.y, nuprev = in.nus.x;
      // parent velocity: universe = 0, chain = registers, otherwise the parent's LDS hand-over slot
      if (d.parent == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) vpar[k] = T(0);
      } else if (sd.flags & SF_VPAR_REG) {
#pragma unroll
        for (int k = 0; k < 6; ++k) vpar[k] = vcur[k];
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) vpar[k] = vedge[(sd.rstart * 6 + k) * WAVE + lane];
      }
      joint_xform<T>(d, rec, in.cs.x, in.cs.y, R, t);
      actinv_motion(R, t, vpar, vp);  // hxx:125
      // nu_i = -UDinv^T v' - Dinv r_i  (hxx:127)
      T udv = in.UD[0] * vp[0];
#pragma unroll
      for (int k = 1; k < 6; ++k) udv += in.UD[k] * vp[k];
      const T nui = -udv - dd * ri;
      N.nu_inf = tmax(N.nu_inf, tabs(nui));
      // v_i = v' + S nu_i (hxx:133-134)
      const T ax0 = (T)d.axis[0], ax1 = (T)d.axis[1], ax2 = (T)d.axis[2];
#pragma unroll
      for (int k = 0; k < 6; ++k) vi[k] = vp[k];
      if (d.flags & JF_REVOLUTE) {
        vi[3] += ax0 * nui; vi[4] += ax1 * nui; vi[5] += ax2 * nui;
        if (d.flags & JF_HELICAL) { const T pn = (T)d.pitch * nui; vi[0] += ax0 * pn; vi[1] += ax1 * pn; vi[2] += ax2 * pn; }
      } else {
        vi[0] += ax0 * nui; vi[1] += ax1 * nui; vi[2] += ax2 * nui;
      }
      // f_i (hxx:139-140) and delta_fis (hxx:137-146) are produced by the residual sweep: see sweep_bwd2
      T dv6[6], hrv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) dv6[k] = vi[k] - in.vprev[k];
      // Href_v (hxx:149-153)
      // a massless chain link is not a body of the model: H_ref = 0 for it, and its velocity stays out of the norm
      const LinkCost<T, HDIAG> lc(P, d.flags, i);
      href_mul<T, HDIAG>(lc.Href, vi, hrv);
      N.href_v = tmax(N.href_v, inf6(hrv));
      if (!(d.flags & JF_MASSLESS)) N.dvis = tmax(N.dvis, inf6(dv6));  // hxx:156-158
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
      stp<T>(rec, JP_WZ, wi + dwi, zi);
      stp<T>(rec, JP_NUS, nui, in.nus.y);  // full 16-byte store: Stf_plus_w rewritten unchanged
      st6<T>(rec, JP_V, vi);
#pragma unroll
      for (int k = 0; k < 6; ++k) vcur[k] = vi[k];
      if (sd.flags & SF_OUT_LDS) {
#pragma unroll
        for (int k = 0; k < 6; ++k) vedge[(sd.wslot * 6 + k) * WAVE + lane] = vi[k];
      }
      // DualUpdate, task part (hxx:410-451)
      if (d.cslot >= 0) {
        char* crec = lp + (size_t)(L.off_c + d.cslot * L.crec) * pair_bytes<T>();
        T A[36], Av[6], e[6], yy[6], aty[6], bb[6];
        if (P.mode & MODE_A_SHARED) {
          const T* Au = Bf.uni + d.cslot * 36;
#pragma unroll
          for (int k = 0; k < 36; ++k) A[k] = Au[k];
        } else {
#pragma unroll
          for (int k = 0; k < 18; ++k) {
            const typename Vec2<T>::type a = ldp<T>(crec, CP_A + k);
            A[2 * k] = a.x; A[2 * k + 1] = a.y;
          }
        }
        ld6<T>(crec, CP_B, bb);
        ld6<T>(crec, CP_Y, yy);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          T a = T(0);
#pragma unroll
          for (int k = 0; k < 6; ++k) a += A[6 * r + k] * vi[k];
          Av[r] = a;
        }
        T plus = T(0), minus = T(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          e[k] = Av[k] - bb[k];
          const T dy = mu_eq * e[k];
          yy[k] += dy;
          N.dyis = tmax(N.dyis, tabs(dy));
          plus += bb[k] * tmax(dy, T(0));
          minus += bb[k] * tmin(dy, T(0));
        }
        N.bTdy_plus += plus;
        N.bTdy_minus += minus;
        N.pr_task = tmax(N.pr_task, inf6(e));
        N.av_inf = tmax(N.av_inf, inf6(Av));
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          T a = T(0);
#pragma unroll
          for (int k = 0; k < 6; ++k) a += A[6 * k + r] * yy[k];
          aty[r] = a;
        }
        st6<T>(crec, CP_Y, yy);
        st6<T>(crec, CP_ATY, aty);
      }
    }
    // request the next step's record behind this step's stores: it is in flight across the barrier.  (Requesting
    // it before the arithmetic measured slower: the compiler's vmcnt(0) waits inside the step then stall on it.)
    if (PF && t_ + 1 < tm.T_down && jn > 0 && live)
      in.load(lp + (size_t)(jn - 1) * JREC * pair_bytes<T>(), bnd_shared);
    step_barrier(nw);
  }
  sweep_barrier(nw);
}

// ------------------------------------------------------------------------------------------------
// f_i by force balance.  Upstream's forward pass sets f_i = H_i v_i + p_i with the accumulated H_i, p_i of the
// backward pass (hxx:139-140).  With H_i = H_i^base + sum_c X*_c H_c^aba X_c^-1, p_i = p_i^base + sum_c X*_c p_c^aba,
// v_c = X_c^-1 v_i + S_c nu_c and nu_c = -Dinv_c (U_c^T X_c^-1 v_i + r_c) one gets H_c^aba X_c^-1 v_i + p_c^aba = f_c,
// hence
//      f_i = H_i^base v_i + p_i^base + sum_children X*_c f_c ,      H_i^base = rho I + H_ref (+ mu_eq A^T A),
// the same number in exact arithmetic, computed leaf -> root from the children's f -- which the residual sweep
// accumulates anyway for g_i (hxx:210-212).  Neither H_i nor p_i is read: the backward sweep does not store H_i
// at all (11 of the 27 pairs the forward sweep used to load per joint, and 11 of the 18 the H sweep used to store).
// hv = H_ref v_i (also needed by the dual residual), sumf = sum_children act(liMi_c, f_c).
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void force_balance(const Params<T>& P, const Bufs<T>& Bf, const JointDesc& d, const char* lp,
                                              T mu_eq, T rho_i, const T* vi, const T* hv, const T* pb, const T* sumf, T* fi)
{
  // rho_i, hv: the link's own rho and H_ref v_i (LinkCost: zero for a massless chain link)
  const Layout& L = P.L;
#pragma unroll
  for (int k = 0; k < 6; ++k) fi[k] = (hv[k] + rho_i * vi[k]) + pb[k];
  if (d.cslot >= 0) {
    T ata[22], av[6];
    if (P.mode & MODE_A_SHARED) {
      const T* AtA = Bf.uni + L.nc * 36 + d.cslot * 21;
#pragma unroll
      for (int k = 0; k < 21; ++k) ata[k] = AtA[k];
    } else {
      const char* crec = lp + (size_t)(L.off_c + d.cslot * L.crec) * pair_bytes<T>();
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const typename Vec2<T>::type a = ldp<T>(crec, CP_ATA + k);
        ata[2 * k] = a.x; ata[2 * k + 1] = a.y;
      }
    }
    symv(ata, vi, av);
#pragma unroll
    for (int k = 0; k < 6; ++k) fi[k] += mu_eq * av[k];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) fi[k] += sumf[k];
}

// ------------------------------------------------------------------------------------------------
// leaf -> root residual sweep: f_i (force balance, above), BwdPass2 + dual residual
// ------------------------------------------------------------------------------------------------
template <typename T, bool HDIAG>
__device__ __forceinline__ void sweep_bwd2(const Params<T>& P, const Bufs<T>& Bf, const Team& tm, int w, int nw,
                                           T* edge, char* lp, int lane, bool live, T mu_eq, Norms<T>& N)
{
  const Layout& L = P.L;
  T acc[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[k] = T(0);
  const StepDesc* __restrict__ steps = tm.up + (size_t)w * tm.T_up;
  int jn = steps[0].joint;
  for (int t_ = 0; t_ < tm.T_up; ++t_) {
    const StepDesc sd = steps[t_];  // by value: one scalar-load burst, never re-read behind a store
    // the joint index is fetched one step ahead: the record loads below must not wait for a descriptor load
    const int i = jn;
    jn = steps[t_ + 1 < tm.T_up ? t_ + 1 : t_].joint;
    if (i > 0 && live) {
      const JointDesc& d = sd.d;
      char* rec = lp + (size_t)(i - 1) * JREC * pair_bytes<T>();
      T fold[6], fi[6], vi[6], gold[6], gi[6], pb[6], sf[6], hv[6];
      const typename Vec2<T>::type cs = ldp<T>(rec, JP_CS), wz = ldp<T>(rec, JP_WZ), nus = ldp<T>(rec, JP_NUS);
      ld6<T>(rec, JP_F, fold);
      ld6<T>(rec, JP_V, vi);
      ld6<T>(rec, JP_G, gold);
      ld6<T>(rec, JP_P, pb);
      const T wi = wz.x, sold = nus.y;
      // sum_children act(f_j), then f_i (hxx:139-140 via force balance) and delta_fis (hxx:137-146)
#pragma unroll
      for (int k = 0; k < 6; ++k) sf[k] = T(0);
      edge_gather<T, 0, 6>(sd, tm.rlist, edge, acc, sf, lane);
      const LinkCost<T, HDIAG> lc(P, d.flags, i);
      href_mul<T, HDIAG>(lc.Href, vi, hv);
      force_balance<T>(P, Bf, d, lp, mu_eq, lc.rho, vi, hv, pb, sf, fi);
      st6<T>(rec, JP_F, fi);
      // g_i = (Aty_c | 0) + sum_children act(f_j) - f_i   (hxx:438-439, :210-212)
      if (d.cslot >= 0) {
        ld6<T>(lp + (size_t)(L.off_c + d.cslot * L.crec) * pair_bytes<T>(), CP_ATY, gi);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) gi[k] = T(0);
      }
      T dg[6], dvr[6], df[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        df[k] = fi[k] - fold[k];
        gi[k] = (gi[k] + sf[k]) - fi[k];
        dg[k] = gi[k] - gold[k];
      }
      st6<T>(rec, JP_G, gi);
      // (the f of a massless chain link is no member of upstream's fis: between the revolute joints of a SphericalZYX chain it
      //  is a ROTATED copy of the body's f, with another inf-norm)
      N.dfis = tmax(N.dfis, (d.flags & JF_MASSLESS) ? T(0) : inf6(df));
      N.dg = tmax(N.dg, inf6(dg));      // hxx:215-220
      N.g_inf = tmax(N.g_inf, inf6(gi));  // hxx:223-225
      // dual residual, v block (hxx:228): Href v_i - Hv + g_i
#pragma unroll
      for (int r = 0; r < 6; ++r) dvr[r] = hv[r] - lc.Hv[r] + gi[r];
      N.dual_v = tmax(N.dual_v, inf6(dvr));
      // Stf_plus_w (hxx:231-236, :482-484)
      const T ax0 = (T)d.axis[0], ax1 = (T)d.axis[1], ax2 = (T)d.axis[2];
      T stf;
      if (d.flags & JF_REVOLUTE) stf = ax0 * fi[3] + ax1 * fi[4] + ax2 * fi[5];
      else stf = ax0 * fi[0] + ax1 * fi[1] + ax2 * fi[2];
      if (d.flags & JF_HELICAL) stf += (T)d.pitch * (ax0 * fi[0] + ax1 * fi[1] + ax2 * fi[2]);
      const T si = stf + wi;
      stp<T>(rec, JP_NUS, nus.x, si);  // full 16-byte store: nu rewritten unchanged
      N.stf_w_inf = tmax(N.stf_w_inf, tabs(si));
      N.dstf_w = tmax(N.dstf_w, tabs(si - sold));
      if (!(d.flags & JF_PARENT_ROOT)) {
        T R[9], t[3], part[6];
        joint_xform<T>(d, rec, cs.x, cs.y, R, t);
        act_force(R, t, fi, part);  // hxx:212
        edge_emit<T, 0, 6>(sd, edge, acc, part, lane);
      }
    }
    step_barrier(nw);
  }
  sweep_barrier(nw);
}

// ------------------------------------------------------------------------------------------------
// fused leaf -> root sweep: the residual sweep of iteration k (BwdPass2, as sweep_bwd2) AND, in the same
// pass over the tree, the p-recursion of iteration k+1 (FwdPass1 + BwdPass with the cached H/UDinv/Dinv, as
// sweep_bwd<.., false, ..>).  Both walk the joints leaf -> root and read the same v_i, w_i, z_i, liMi, so fusing
// them removes one of the three tree walks of an iteration.  The p-recursion is speculative in mu: it uses the
// mu of iteration k; if the epilogue of iteration k changes mu for any lane of the team, the next iteration
// starts with a full sweep_bwd<.., true, ..> that rebuilds H, UDinv, Dinv, p and r with the new mu.
// The arithmetic of each half is unchanged, so results are bit-identical to the unfused sweeps.
// Same one-step prefetch as sweep_fwd.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct FusedIn {
  typename Vec2<T>::type cs, wz, nus, rd;
  T fold[6], vi[6], gold[6], UD[6], pb[6];
  __device__ __forceinline__ void load(const char* rec)
  {
    cs = ldp<T>(rec, JP_CS); wz = ldp<T>(rec, JP_WZ); nus = ldp<T>(rec, JP_NUS); rd = ldp<T>(rec, JP_R);
    ld6<T>(rec, JP_F, fold);
    ld6<T>(rec, JP_V, vi);
    ld6<T>(rec, JP_G, gold);
    ld6<T>(rec, JP_UD, UD);
    ld6<T>(rec, JP_P, pb);
  }
};

template <typename T, bool HDIAG, bool PF>
__device__ __forceinline__ void sweep_fused(const Params<T>& P, const Bufs<T>& Bf, const Team& tm, int w, int nw,
                                            T* edge, char* lp, int lane, bool live, T mu_eq, T mu_in, Norms<T>& N)
{
  const Layout& L = P.L;
  T acc[12];  // [0,6): sum of act(f_child), [6,12): sum of act(p_aba child)
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = T(0);
  const StepDesc* __restrict__ steps = tm.up + (size_t)w * tm.T_up;
  FusedIn<T> in;
  if (PF) {
    const int i0 = steps[0].joint;
    if (i0 > 0 && live) in.load(lp + (size_t)(i0 - 1) * JREC * pair_bytes<T>());
  }
  int jn = steps[0].joint;
  for (int t_ = 0; t_ < tm.T_up; ++t_) {
    const StepDesc sd = steps[t_];  // by value: one scalar-load burst, never re-read behind a store
    // the joint index is fetched one step ahead: the record loads below must not wait for a descriptor load
    const int i = jn;
    jn = steps[t_ + 1 < tm.T_up ? t_ + 1 : t_].joint;
    if (i > 0 && live) {
      const JointDesc& d = sd.d;
      char* rec = lp + (size_t)(i - 1) * JREC * pair_bytes<T>();
      if (!PF) in.load(rec);
      T gi[6], pp[6], sf[6], fi[6], hv[6];
      const T wi = in.wz.x, sold = in.nus.y;
      // ---- iteration k: f_i by force balance, g_i = (Aty_c | 0) + sum_children act(f_j) - f_i   (hxx:438-439, :210-212)
      //      iteration k+1: p_i = -rho v_i - Hv (+ Aty_c - mu_eq Atb_c)       (hxx:304-315, :321-334)
      const LinkCost<T, HDIAG> lc(P, d.flags, i);
#pragma unroll
      for (int k = 0; k < 6; ++k) { pp[k] = -lc.rho * in.vi[k] - lc.Hv[k]; sf[k] = T(0); }
      if (d.cslot >= 0) {
        const char* crec = lp + (size_t)(L.off_c + d.cslot * L.crec) * pair_bytes<T>();
        T atb[6];
        ld6<T>(crec, CP_ATY, gi);
        ld6<T>(crec, CP_ATB, atb);
#pragma unroll
        for (int k = 0; k < 6; ++k) pp[k] += gi[k] - mu_eq * atb[k];
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) gi[k] = T(0);
      }
      st6<T>(rec, JP_P, pp);  // p^base of iteration k+1 (the slot's p^base of iteration k is already in `in.pb`)
      edge_gather<T, 0, 6>(sd, tm.rlist, edge, acc, sf, lane);
      edge_gather<T, 6, 6>(sd, tm.rlist, edge, acc + 6, pp, lane);
      href_mul<T, HDIAG>(lc.Href, in.vi, hv);
      force_balance<T>(P, Bf, d, lp, mu_eq, lc.rho, in.vi, hv, in.pb, sf, fi);
      st6<T>(rec, JP_F, fi);
      T dg[6], dvr[6], df[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        df[k] = fi[k] - in.fold[k];
        gi[k] = (gi[k] + sf[k]) - fi[k];
        dg[k] = gi[k] - in.gold[k];
      }
      st6<T>(rec, JP_G, gi);
      N.dfis = tmax(N.dfis, (d.flags & JF_MASSLESS) ? T(0) : inf6(df));
      N.dg = tmax(N.dg, inf6(dg));        // hxx:215-220
      N.g_inf = tmax(N.g_inf, inf6(gi));  // hxx:223-225
      // dual residual, v block (hxx:228): Href v_i - Hv + g_i
#pragma unroll
      for (int r = 0; r < 6; ++r) dvr[r] = hv[r] - lc.Hv[r] + gi[r];
      N.dual_v = tmax(N.dual_v, inf6(dvr));
      // Stf_plus_w (hxx:231-236, :482-484) and r_i = (w_i - mu_in z_i) + S^T p_i (hxx:296, :70)
      const T ax0 = (T)d.axis[0], ax1 = (T)d.axis[1], ax2 = (T)d.axis[2];
      T stf, Stp;
      if (d.flags & JF_REVOLUTE) {
        stf = ax0 * fi[3] + ax1 * fi[4] + ax2 * fi[5];
        Stp = ax0 * pp[3] + ax1 * pp[4] + ax2 * pp[5];
        if (d.flags & JF_HELICAL) {
          stf += (T)d.pitch * (ax0 * fi[0] + ax1 * fi[1] + ax2 * fi[2]);
          Stp += (T)d.pitch * (ax0 * pp[0] + ax1 * pp[1] + ax2 * pp[2]);
        }
      } else {
        stf = ax0 * fi[0] + ax1 * fi[1] + ax2 * fi[2];
        Stp = ax0 * pp[0] + ax1 * pp[1] + ax2 * pp[2];
      }
      const T si = stf + wi;
      const T ri = (in.wz.x - mu_in * in.wz.y) + Stp;
      stp<T>(rec, JP_NUS, in.nus.x, si);
      stp<T>(rec, JP_R, ri, in.rd.y);  // Dinv of the cache rewritten unchanged
      N.stf_w_inf = tmax(N.stf_w_inf, tabs(si));
      N.dstf_w = tmax(N.dstf_w, tabs(si - sold));
      if (!(d.flags & JF_PARENT_ROOT)) {
        T R[9], t[3], part[12], pa[6];
        joint_xform<T>(d, rec, in.cs.x, in.cs.y, R, t);
        act_force(R, t, fi, part);  // hxx:212
#pragma unroll
        for (int k = 0; k < 6; ++k) pa[k] = pp[k] - in.UD[k] * ri;  // hxx:71-73
        act_force(R, t, pa, part + 6);                               // hxx:74
        edge_emit<T, 0, 12>(sd, edge, acc, part, lane);
      }
    }
    if (PF && t_ + 1 < tm.T_up && jn > 0 && live) in.load(lp + (size_t)(jn - 1) * JREC * pair_bytes<T>());
    step_barrier(nw);
  }
  sweep_barrier(nw);
}

// scalar record accessors
template <typename T>
__device__ __forceinline__ T ld_scal(const char* srec, int idx)
{
  const typename Vec2<T>::type v = ldp<T>(srec, SP_SCAL + idx / 2);
  return (idx & 1) ? v.y : v.x;
}
template <typename T>
__device__ __forceinline__ void st_scal(char* srec, int idx, T x)
{
  if (idx & 1) st_hi<T>(srec, SP_SCAL + idx / 2, x);
  else st_lo<T>(srec, SP_SCAL + idx / 2, x);
}

// ------------------------------------------------------------------------------------------------
// persistent solve kernel: each wavefront iterates its 64 instances until all are done (or the launch
// iteration budget is spent).  No inter-wavefront communication: instances are independent.
// ------------------------------------------------------------------------------------------------
// (A build capped at 256 registers -- amdgpu_waves_per_eu(2,2), two workgroups per CU, the H sweep spilling ~330 B/lane
//  to scratch -- was measured: 289 vs 186 M instance-iterations/s in a 100-iteration steady-state run at B = 65536, but
//  on the real workload (8-iteration launches, H rebuilt at every iteration) every launch is slower, 2.54 vs 2.08 ms
//  for the first one, 47.1 vs 45.4 ms/step.  One wavefront per SIMD with the whole register file it is.)
template <typename T, bool HDIAG, bool TEAM>
__global__ void __launch_bounds__(TEAM ? WAVE * MAX_TEAM : WAVE)
k_solve(const Params<T> P, const Bufs<T> Bf, const StepDesc* __restrict__ sched_up,
        const StepDesc* __restrict__ sched_down, const int* __restrict__ sched_rlist, int T_up, int T_down, int edge_ent)
{
  const Team tm{sched_up, sched_down, sched_rlist, T_up, T_down, edge_ent};
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* edge = reinterpret_cast<T*>(smem_raw);  // edge slots of the leaf->root sweeps / scalar exchange of the team
  const Layout& L = P.L;
  const int lane = threadIdx.x & (WAVE - 1);
  const int w = TEAM ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;  // wavefront of the team
  const int nw = TEAM ? (int)(blockDim.x >> 6) : 1;
  const int b = blockIdx.x * WAVE + lane;
  const bool inb = b < P.B;
  char* lp = Bf.tiles + (size_t)blockIdx.x * L.tile_pairs * pair_bytes<T>() + (size_t)lane * 2 * sizeof(T);
  char* srec = lp + (size_t)L.off_s * pair_bytes<T>();

  const typename Vec2<T>::type mu2 = ldp<T>(srec, SP_MU), bi2 = ldp<T>(srec, SP_BI), st2 = ldp<T>(srec, SP_ST),
                               tg01 = ldp<T>(srec, SP_TAG);
  int status = inb ? (int)st2.x : ST_DONE;
  int iter = (int)bi2.y;
  T mu = mu2.x;
  int kexp = (int)mu2.y;                      // mu = mu0 * 10^kexp
  T mu_last = st2.y;                          // mu of the last executed iteration (for the His / pis / UDinv getters)
  T tag = tg01.x;                             // mu the cached UDinv / Dinv were computed with (-1: none)
  int nflip = (int)ldp<T>(srec, SP_FLIP).x;   // mu updates so far
  const T bnorm = bi2.x;
  bool live = inb && !(status & ST_DONE);
  // main-loop bound `for (i = 1; i < max_iter; ++i)` (hpp:377): nothing to do when max_iter <= 1
  if (live && !(status & ST_TAIL) && iter + 1 >= P.max_iter) { live = false; status |= ST_DONE; }
  unsigned int my_iters = 0;
  unsigned int n_tile_iters = 0, n_h_iters = 0, n_fused_iters = 0;  // sweep mix of this tile (diagnostics)
  // scalars of the per-iteration dump that keep their last value when an iteration does not re-evaluate them
  // (CheckFeasibility is skipped at iteration 1 and in the tail solve): carried in registers, no reload per iteration
  T p_tolp, p_told, p_dyqp, p_atdy, p_ubp, p_lbm;
  int p_c1, p_c2, p_tail_iter;
  {
    const typename Vec2<T>::type o3 = ldp<T>(srec, SP_SCAL + 3), o6 = ldp<T>(srec, SP_SCAL + 6),
                                 o7 = ldp<T>(srec, SP_SCAL + 7), o8 = ldp<T>(srec, SP_SCAL + 8),
                                 o13 = ldp<T>(srec, SP_SCAL + 13), o14 = ldp<T>(srec, SP_SCAL + 14);
    p_tolp = o3.x; p_told = o3.y; p_dyqp = o6.y; p_atdy = o7.x; p_ubp = o7.y; p_lbm = o8.x;
    p_c1 = (int)o13.y; p_c2 = (int)o14.x; p_tail_iter = (int)o14.y;
  }
  bool have_p = false;  // p_i, r_i of the coming iteration already built by the fused sweep (with the current mu)
  bool spec = true;     // fuse the next p-recursion into the residual sweep (pays only if no lane then changes mu)

  for (int k = 0; k < P.max_launch_iters; ++k) {
    if (!__any(live)) break;
    const T mu_eq = P.mu_scale * mu;  // hpp:183-184, hxx:620-621
    const T mu_in = mu;
    Norms<T> N;
    N.reset();
    if (live) { ++iter; ++my_iters; mu_last = mu; status &= ~ST_PFULL; }

    // leaf -> root sweep of this iteration, unless the previous iteration's fused sweep already did it
    if (!have_p) {
      const bool need_h = !(P.mode & MODE_CACHE_H) || __any(live && (tag != mu));
      n_h_iters += need_h;
      if (need_h) {
        sweep_bwd<T, true, HDIAG>(P, Bf, tm, w, nw, edge, lp, lane, live, mu_eq, mu_in);
        if (live) tag = mu;
      } else {
        sweep_bwd<T, false, HDIAG>(P, Bf, tm, w, nw, edge, lp, lane, live, mu_eq, mu_in);
      }
    }
    sweep_fwd<T, HDIAG, TEAM>(P, Bf, tm, w, nw, edge + (size_t)tm.edge_ent * WAVE, lp, lane, live, mu_eq, mu_in, N);
    ++n_tile_iters;
    n_fused_iters += ((P.mode & MODE_CACHE_H) && spec);
    if ((P.mode & MODE_CACHE_H) && spec) {
      sweep_fused<T, HDIAG, TEAM>(P, Bf, tm, w, nw, edge, lp, lane, live, mu_eq, mu_in, N);
      have_p = true;
    } else {
      sweep_bwd2<T, HDIAG>(P, Bf, tm, w, nw, edge, lp, lane, live, mu_eq, N);
      have_p = false;
    }
    if (nw > 1) N.team_combine(edge, w, nw, lane);
    const T mu_before = mu;

    if (live) {
      // ComputePrimalResiduals / ComputeDualResiduals (hxx:494-522)
      const T primal = tmax(N.pr_task, N.pr_slack);
      const T dual = tmax(N.dual_v, N.stf_w_inf);
      T tol_p = p_tolp, tol_d = p_told;
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
          if (update_mu<T>(P.mode, primal, dual, tmax(tmax(N.av_inf, N.nu_inf), bnorm),
                           tmax(tmax(N.href_v, tmax(N.g_inf, N.stf_w_inf)), P.Hv_inf_norm), mu, kexp))
            ++nflip;
          if (iter + 1 >= P.max_iter) { status |= ST_DONE; live = false; }
        }
      } else {
        // tail-solve iteration (hpp:286-308)
        tail_iter = p_tail_iter + 1;
        if (!(dx >= P.tol_tail_solve || N.dz >= P.tol_tail_solve) || iter >= P.max_iter) {
          status |= ST_DONE;
          live = false;
        }
      }
      // per-iteration scalar dump: full 16-byte pairs of the scalar record.  The feasibility scalars keep their
      // last evaluated value when CheckFeasibility did not run this iteration (iter 1, tail solve), as upstream.
      if (!ran_feas) { dyqp = p_dyqp; atdy = p_atdy; ubp = p_ubp; lbm = p_lbm; c1 = p_c1; c2 = p_c2; }
      if (!(status & ST_TAIL)) tail_iter = p_tail_iter;
      p_tolp = tol_p; p_told = tol_d; p_dyqp = dyqp; p_atdy = atdy; p_ubp = ubp; p_lbm = lbm;
      p_c1 = c1; p_c2 = c2; p_tail_iter = tail_iter;
      if (w == 0) {
      stp<T>(srec, SP_SCAL + 0, primal, dual);                     // PRIMAL_RES, DUAL_RES
      stp<T>(srec, SP_SCAL + 1, N.pr_task, N.pr_slack);            // PRIMAL_RES_TASK, _SLACK
      stp<T>(srec, SP_SCAL + 2, N.dual_v, N.stf_w_inf);            // DUAL_RES_V, DUAL_RES_NU
      stp<T>(srec, SP_SCAL + 3, tol_p, tol_d);                     // TOL_PRIMAL, TOL_DUAL
      stp<T>(srec, SP_SCAL + 4, mu, P.mu_scale * mu);              // MU, MU_EQ
      stp<T>(srec, SP_SCAL + 5, mu, dx);                           // MU_INEQ, DELTA_X_QP
      stp<T>(srec, SP_SCAL + 6, N.dz, dyqp);                       // DELTA_Z_QP, DELTA_Y_QP
      stp<T>(srec, SP_SCAL + 7, atdy, ubp);                        // AT_DELTA_Y_QP, UB_DY_PLUS
      stp<T>(srec, SP_SCAL + 8, lbm, N.dfis);                      // LB_DY_MINUS, DELTA_FIS
      stp<T>(srec, SP_SCAL + 9, N.dyis, N.dw);                     // DELTA_YIS, DELTA_W
      stp<T>(srec, SP_SCAL + 10, N.dvis, N.dnu);                   // DELTA_VIS, DELTA_NU
      stp<T>(srec, SP_SCAL + 11, N.av_inf, N.nu_inf);              // AV_INF, NU_INF
      stp<T>(srec, SP_SCAL + 12, N.href_v, N.g_inf);               // HREF_V_INF, G_INF
      stp<T>(srec, SP_SCAL + 13, N.stf_w_inf, (T)c1);              // STF_PLUS_W_INF, COND1
      stp<T>(srec, SP_SCAL + 14, (T)c2, (T)tail_iter);             // COND2, TAIL_ITER
      }
    }
    // the speculative p-recursion used this iteration's mu: redo the leaf -> root sweep if any lane moved on; and
    // speculate again only after an iteration in which no lane of the wavefront changed mu (wavefronts full of
    // stragglers flip mu almost every iteration: there the plain three-sweep iteration is cheaper)
    const bool changed = __any(live && (mu != mu_before));
    if (changed) have_p = false;
    spec = !changed;
  }

  if (inb && w == 0) {
    stp<T>(srec, SP_MU, mu, (T)kexp);
    stp<T>(srec, SP_TAG, tag, T(0));
    stp<T>(srec, SP_FLIP, (T)nflip, T(0));
    stp<T>(srec, SP_BI, bnorm, (T)iter);
    stp<T>(srec, SP_ST, (T)status, mu_last);
  }
  const unsigned long long live_mask = __ballot(live);
  unsigned int it_sum = my_iters;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) it_sum += __shfl_down(it_sum, off);
  if (lane == 0 && w == 0) {
    const unsigned int nlive = __popcll(live_mask);
    Bf.wave_live[blockIdx.x] = (int)nlive;
    if (nlive) atomicAdd(&Bf.counters[0], nlive);
    if (it_sum) atomicAdd(&Bf.counters[1], it_sum);
    atomicAdd(&Bf.counters[2], n_tile_iters);
    atomicAdd(&Bf.counters[3], n_h_iters);
    atomicAdd(&Bf.counters[4], n_fused_iters);
  }
}
static_assert(SC_PRIMAL_RES == 0 && SC_DUAL_RES == 1 && SC_TOL_PRIMAL == 6 && SC_MU == 8 && SC_MU_INEQ == 10 &&
              SC_DELTA_X_QP == 11 && SC_DELTA_Z_QP == 12 && SC_DELTA_Y_QP == 13 && SC_AT_DELTA_Y_QP == 14 &&
              SC_UB_DY_PLUS == 15 && SC_LB_DY_MINUS == 16 && SC_DELTA_FIS == 17 && SC_DELTA_YIS == 18 &&
              SC_DELTA_VIS == 20 && SC_AV_INF == 22 && SC_HREF_V_INF == 24 && SC_STF_PLUS_W_INF == 26 &&
              SC_COND1 == 27 && SC_COND2 == 28 && SC_TAIL_ITER == 29, "scalar dump pairs out of sync with the SC_* enum");

// ------------------------------------------------------------------------------------------------
// host-side plumbing kernels (one lane per instance, tile addressing)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ char* lane_ptr(char* tiles, const Layout& L, int b)
{
  return tiles + (size_t)(b / WAVE) * L.tile_pairs * pair_bytes<T>() + (size_t)(b % WAVE) * 2 * sizeof(T);
}
template <typename T>
__device__ __forceinline__ T* elem_ptr(char* lp, int pair, int half)
{
  return reinterpret_cast<T*>(lp + (size_t)pair * pair_bytes<T>() + (size_t)half * sizeof(T));
}

// FwdPassInit (hxx:253-283): joint configuration -> per-joint (cos q, sin q) | (q, 0); the first joint of the chain of
// a multi-DoF joint spreads that joint's (t, quat) over the JP_CS pairs of the chain (layout: see ROT_FREE).
// q is instance-major [B][nq] (the caller's layout) or one shared [nq]; idx_q[i] = where joint i's coordinates start.
template <typename T>
__device__ __forceinline__ void fk_init_joint(const double* __restrict__ qrow, const JointDesc* __restrict__ jd, const int* __restrict__ idx_q,
                                              char* lp, int i)
{
  constexpr size_t RB = (size_t)JREC * pair_bytes<T>();
  {
    if (jd[i].flags & JF_NOQ) return;  // its pair belongs to the first joint of the chain
    const double* qs = qrow + idx_q[i];
    char* rec = lp + (size_t)(i - 1) * RB;
    const int rot = jd[i].rot;
    if (rot == ROT_FREE) {
      stp<T>(rec, JP_CS, (T)qs[0], (T)qs[1]);
      stp<T>(rec + RB, JP_CS, (T)qs[2], (T)qs[3]);
      stp<T>(rec + 2 * RB, JP_CS, (T)qs[4], (T)qs[5]);
      stp<T>(rec + 3 * RB, JP_CS, (T)qs[6], T(0));
      return;
    }
    if (rot == ROT_SPH) {
      stp<T>(rec, JP_CS, (T)qs[0], (T)qs[1]);
      stp<T>(rec + RB, JP_CS, (T)qs[2], (T)qs[3]);
      return;
    }
    if (rot == ROT_TRANS) {
      stp<T>(rec, JP_CS, (T)qs[0], (T)qs[1]);
      stp<T>(rec + RB, JP_CS, (T)qs[2], T(0));
      return;
    }
    if (rot == ROT_PLANAR) {
      stp<T>(rec, JP_CS, (T)qs[0], (T)qs[1]);
      stp<T>(rec + RB, JP_CS, (T)qs[2], (T)qs[3]);
      return;
    }
    if (jd[i].flags & JF_CS_DIRECT) {  // JointModelRevoluteUnbounded: q = (cos, sin)
      stp<T>(rec, JP_CS, (T)qs[0], (T)qs[1]);
      return;
    }
    const double qi = qs[0];
    T c, s;
    if ((jd[i].flags & JF_REVOLUTE) && !(jd[i].flags & JF_HELICAL)) {
      double sd, cd;
      sincos(qi, &sd, &cd);
      c = (T)cd; s = (T)sd;
    } else {
      c = (T)qi; s = T(0);
    }
    stp<T>(rec, JP_CS, c, s);
  }
}
template <typename T>
__global__ void k_fk_init(const double* __restrict__ q, int nq, int q_shared, const JointDesc* __restrict__ jd,
                          const int* __restrict__ idx_q, Layout L, int B, char* tiles)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  char* lp = lane_ptr<T>(tiles, L, b);
  const double* qrow = q + (q_shared ? 0 : (size_t)b * nq);
  for (int i = 1; i <= L.nb; ++i) fk_init_joint<T>(qrow, jd, idx_q, lp, i);
}
// a small batch's SolveInit (one problem per call: the reference's own use): the caller's q into the resident copy AND FwdPassInit's pairs in one
// launch, one thread per coordinate / joint -- a thread per instance walks 32 joints one after the other, 19 + 7 us for one problem
template <typename T>
__global__ void k_set_q_fk_small(double* __restrict__ q_res, const double* __restrict__ src, int src_shared, int nq,
                                 const JointDesc* __restrict__ jd, const int* __restrict__ idx_q, Layout L, int B, char* tiles)
{
  const int per = nq > L.nb ? nq : L.nb;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = (int)(idx / per), t = (int)(idx - (long long)b * per);
  if (b >= B) return;
  const double* qrow = src + (src_shared ? 0 : (size_t)b * nq);
  if (t < nq) q_res[(size_t)b * nq + t] = qrow[t];
  if (t < L.nb) fk_init_joint<T>(qrow, jd, idx_q, lane_ptr<T>(tiles, L, b), t + 1);
}

// ---- configuration-space integration of the quaternion joints (what pinocchio::integrate does for
// JointModelFreeFlyer / JointModelSpherical: SpecialEuclideanOperation<3> / SpecialOrthogonalOperation<3>), fp64 --------
__device__ __forceinline__ void quat_mul_xyzw(const double* a, const double* b, double* o)
{
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
// quaternion of exp([w]x)
__device__ __forceinline__ void so3_exp_quat(const double* w, double* qe)
{
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double k, cw;
  if (t2 > 1.220703125e-4) {  // sqrt(sqrt(eps))
    const double t = sqrt(t2);
    double sh, ch;
    sincos(0.5 * t, &sh, &ch);
    k = sh / t; cw = ch;
  } else {
    k = 0.5 - t2 / 48.0; cw = 1.0 - t2 / 8.0;
  }
  qe[0] = k * w[0]; qe[1] = k * w[1]; qe[2] = k * w[2]; qe[3] = cw;
}
// q.coeffs() *= (3 - |q|^2) / 2   (quaternion::firstOrderNormalize)
__device__ __forceinline__ void quat_first_order_normalize(double* qt)
{
  const double n2 = qt[0] * qt[0] + qt[1] * qt[1] + qt[2] * qt[2] + qt[3] * qt[3];
  const double a = (3.0 - n2) / 2.0;
  for (int k = 0; k < 4; ++k) qt[k] *= a;
}
// (t, quat) <- (t, quat) * exp6(v), v = [linear; angular] in the body frame
__device__ __forceinline__ void se3_integrate(double* q7, const double* v)
{
  const double* w = v + 3;
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double a, b2;  // p = v_l + a (w x v_l) + b (w x (w x v_l))
  if (t2 > 1.220703125e-4) {
    const double t = sqrt(t2);
    double st, ct;
    sincos(t, &st, &ct);
    a = (1.0 - ct) / t2;
    b2 = (t - st) / (t2 * t);
  } else {
    a = 0.5 - t2 / 24.0;
    b2 = 1.0 / 6.0 - t2 / 120.0;
  }
  const double wxv[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
  const double wwv[3] = {w[1] * wxv[2] - w[2] * wxv[1], w[2] * wxv[0] - w[0] * wxv[2], w[0] * wxv[1] - w[1] * wxv[0]};
  double p[3], R[9], qe[4], qo[4];
  for (int k = 0; k < 3; ++k) p[k] = v[k] + a * wxv[k] + b2 * wwv[k];
  quat_to_rot<double>(q7[3], q7[4], q7[5], q7[6], R);
  for (int k = 0; k < 3; ++k) q7[k] += R[3 * k] * p[0] + R[3 * k + 1] * p[1] + R[3 * k + 2] * p[2];
  so3_exp_quat(w, qe);
  quat_mul_xyzw(q7 + 3, qe, qo);
  const double dot = qo[0] * q7[3] + qo[1] * q7[4] + qo[2] * q7[5] + qo[3] * q7[6];
  if (dot < 0.0)
    for (int k = 0; k < 4; ++k) qo[k] = -qo[k];
  quat_first_order_normalize(qo);
  for (int k = 0; k < 4; ++k) q7[3 + k] = qo[k];
}

// outer loop: q <- q (+) dt * z on the resident configurations (1-DoF and translation joints: a plain sum; free-flyer
// and spherical joints: the Lie-group update above), `src` != nullptr first (re)fills the resident copy from a
// caller's q (one shared row or one row per instance)
template <typename T>
__global__ void k_advance_q(double* __restrict__ q_res, const double* __restrict__ src, int src_shared, int nq,
                            const JointDesc* __restrict__ jd, const int* __restrict__ idx_q, Layout L, int B,
                            const char* tiles, double dt)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (src) {
    for (int k = 0; k < nq; ++k) q_res[(size_t)b * nq + k] = src[(src_shared ? 0 : (size_t)b * nq) + k];
    return;
  }
  const char* lp = lane_ptr<T>(const_cast<char*>(tiles), L, b);
  constexpr size_t RB = (size_t)JREC * pair_bytes<T>();
  for (int i = 1; i <= L.nb; ++i) {
    if (jd[i].flags & JF_NOQ) continue;
    const char* rec = lp + (size_t)(i - 1) * RB;
    double* qs = q_res + (size_t)b * nq + idx_q[i];
    const int rot = jd[i].rot;
    const int n = rot == ROT_FREE ? 6 : (rot == ROT_SPH || rot == ROT_TRANS || rot == ROT_PLANAR) ? 3 : 1;
    double v[6];
    for (int k = 0; k < n; ++k) v[k] = dt * (double)ldp<T>(rec + k * RB, JP_WZ).y;  // z of the chain's joints
    if (rot == ROT_FREE) {
      se3_integrate(qs, v);
    } else if (rot == ROT_SPH) {
      double qe[4], qo[4];
      so3_exp_quat(v, qe);
      quat_mul_xyzw(qs, qe, qo);
      quat_first_order_normalize(qo);
      for (int k = 0; k < 4; ++k) qs[k] = qo[k];
    } else if (rot == ROT_PLANAR) {
      // SpecialEuclideanOperationTpl<2>::integrate: (x, y, cos, sin) * exp(vx, vy, w), first-order re-normalised (cos, sin)
      const double c0 = qs[2], s0 = qs[3], w = v[2];
      double sw, cw, tx, ty;
      sincos(w, &sw, &cw);
      if (fabs(w) > 1e-14) { tx = (sw * v[0] - (1.0 - cw) * v[1]) / w; ty = ((1.0 - cw) * v[0] + sw * v[1]) / w; }
      else { tx = v[0]; ty = v[1]; }
      qs[0] += c0 * tx - s0 * ty;
      qs[1] += s0 * tx + c0 * ty;
      double c1 = c0 * cw - s0 * sw, s1 = s0 * cw + c0 * sw;
      const double nrm = 0.5 * (3.0 - (c1 * c1 + s1 * s1));
      qs[2] = c1 * nrm; qs[3] = s1 * nrm;
    } else if (jd[i].flags & JF_CS_DIRECT) {
      // SpecialOrthogonalOperationTpl<2>::integrate: (cos, sin) rotated by w, first-order re-normalised
      const double c0 = qs[0], s0 = qs[1];
      double sw, cw;
      sincos(v[0], &sw, &cw);
      double c1 = c0 * cw - s0 * sw, s1 = s0 * cw + c0 * sw;
      const double nrm = 0.5 * (3.0 - (c1 * c1 + s1 * s1));
      qs[0] = c1 * nrm; qs[1] = s1 * nrm;
    } else {
      for (int k = 0; k < n; ++k) qs[k] += v[k];
    }
  }
}

// instance-major [B][n] doubles (or one shared [n]) -> tile elements given by rowmap[r] = pair*2 + half
template <typename T>
__global__ void k_upload_rows(const double* __restrict__ src, int n, int shared, const int* __restrict__ rowmap,
                              Layout L, int B, char* tiles)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  char* lp = lane_ptr<T>(tiles, L, b);
  for (int r = 0; r < n; ++r) {
    const int m = rowmap[r];
    *elem_ptr<T>(lp, m >> 1, m & 1) = (T)src[(shared ? 0 : (size_t)b * n) + r];
  }
}

// tile elements -> instance-major [B][n] doubles; as_int: write int32 instead
template <typename T>
__global__ void k_download_rows(char* tiles, Layout L, const int* __restrict__ rowmap, int n, int B,
                                double* __restrict__ dst, int as_int, int mask)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  char* lp = lane_ptr<T>(tiles, L, b);
  for (int r = 0; r < n; ++r) {
    const int m = rowmap[r];
    const T x = *elem_ptr<T>(lp, m >> 1, m & 1);
    if (as_int) {
      const int v = (int)x;
      // mask > 0: flag test; mask < 0: keep the bits of -mask; 0: the value itself
      reinterpret_cast<int*>(dst)[(size_t)b * n + r] = mask > 0 ? ((v & mask) ? 1 : 0) : (mask < 0 ? (v & -mask) : v);
    } else {
      dst[(size_t)b * n + r] = (double)x;
    }
  }
}

// the same, one ELEMENT per thread: what a small batch wants (one problem per call -- the reference's own use -- is one instance
// and ~400 rows: a thread per instance walks them one dependent load after the other)
template <typename T>
__global__ void k_download_elems(char* tiles, Layout L, const int* __restrict__ rowmap, int n, int B, double* __restrict__ dst)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * n) return;
  const int b = (int)(i / n), r = (int)(i - (long long)b * n);
  const int m = rowmap[r];
  dst[i] = (double)*elem_ptr<T>(lane_ptr<T>(tiles, L, b), m >> 1, m & 1);
}

// ---- getters for quantities the hot path no longer materialises -----------------------------------------------
// ik_id_data.His[i] (accumulated, pre-projection: what upstream leaves in His after BwdPass, hxx:60-67) for the mu of
// the last executed iteration.  One instance per thread, out = [B][nb][21]; the output rows double as the running
// accumulators of the leaf -> root recursion.  Same arithmetic as sweep_bwd<.., true, ..>.
template <typename T>
__global__ void k_rebuild_his(const char* tiles, Layout L, const JointDesc* __restrict__ jd, const T* __restrict__ uni,
                              T rho, T mu_scale, const T* __restrict__ href_tab, int a_shared, int B, double* __restrict__ out)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const char* lp = lane_ptr<T>(const_cast<char*>(tiles), L, b);
  const T mu = *elem_ptr<T>(const_cast<char*>(lp) + (size_t)L.off_s * pair_bytes<T>(), SP_ST, 1);
  const T mu_eq = mu_scale * mu, mu_in = mu;
  double* o = out + (size_t)b * L.nb * 21;
  for (int i = 1; i <= L.nb; ++i) {
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c)
        o[(i - 1) * 21 + sym(r, c)] =
            (jd[i].flags & JF_MASSLESS) ? 0.0 : (double)((r == c ? rho : T(0)) + href_tab[(size_t)i * HREF_ROW + 6 * r + c]);
    if (jd[i].cslot >= 0) {
      const int cs = jd[i].cslot;
      for (int k = 0; k < 21; ++k) {
        const T a = a_shared ? uni[L.nc * 36 + cs * 21 + k]
                             : *elem_ptr<T>(const_cast<char*>(lp) + (size_t)(L.off_c + cs * L.crec) * pair_bytes<T>(), CP_ATA + k / 2, k & 1);
        o[(i - 1) * 21 + k] += (double)(mu_eq * a);
      }
    }
  }
  for (int i = L.nb; i >= 1; --i) {
    const JointDesc d = jd[i];
    if (d.parent == 0) continue;
    T hh[21], U[6], UD[6], R[9], t[3], part[21];
    for (int k = 0; k < 21; ++k) hh[k] = (T)o[(i - 1) * 21 + k];
    const int a0 = (d.flags & JF_REVOLUTE) ? 3 : 0;
    const T ax0 = (T)d.axis[0], ax1 = (T)d.axis[1], ax2 = (T)d.axis[2];
    const T ph = (d.flags & JF_HELICAL) ? (T)d.pitch : T(0);  // S = [pitch a; a]
    for (int k = 0; k < 6; ++k) {
      U[k] = hh[sym(k, a0)] * ax0 + hh[sym(k, a0 + 1)] * ax1 + hh[sym(k, a0 + 2)] * ax2;
      if (d.flags & JF_HELICAL) U[k] += ph * (hh[sym(k, 0)] * ax0 + hh[sym(k, 1)] * ax1 + hh[sym(k, 2)] * ax2);
    }
    T sus = ax0 * U[a0] + ax1 * U[a0 + 1] + ax2 * U[a0 + 2];
    if (d.flags & JF_HELICAL) sus = (ax0 * U[3] + ax1 * U[4] + ax2 * U[5]) + ph * (ax0 * U[0] + ax1 * U[1] + ax2 * U[2]);
    const T dd = T(1) / (sus + mu_in);
    for (int k = 0; k < 6; ++k) UD[k] = U[k] * dd;
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c) hh[sym(r, c)] -= UD[r] * U[c];
    const char* rec = lp + (size_t)(i - 1) * JREC * pair_bytes<T>();
    const typename Vec2<T>::type cs = ldp<T>(rec, JP_CS);
    joint_xform<T>(d, rec, cs.x, cs.y, R, t);
    congr_sym(R, t, hh, part);
    for (int k = 0; k < 21; ++k) o[(d.parent - 1) * 21 + k] += (double)part[k];
  }
}

// ik_id_data.pis[i] (accumulated p_i of the last backward pass, hxx:70-75) from the stored p_i^base, UDinv_i, r_i.
// k_tail leaves the accumulated p_i itself in the slot and flags the instance ST_PFULL.
template <typename T>
__global__ void k_rebuild_pis(const char* tiles, Layout L, const JointDesc* __restrict__ jd, int B, double* __restrict__ out)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const char* lp = lane_ptr<T>(const_cast<char*>(tiles), L, b);
  const int status = (int)*elem_ptr<T>(const_cast<char*>(lp) + (size_t)L.off_s * pair_bytes<T>(), SP_ST, 0);
  double* o = out + (size_t)b * L.nb * 6;
  for (int i = 1; i <= L.nb; ++i) {
    T p[6];
    ld6<T>(lp + (size_t)(i - 1) * JREC * pair_bytes<T>(), JP_P, p);
    for (int k = 0; k < 6; ++k) o[(i - 1) * 6 + k] = (double)p[k];
  }
  if (status & ST_PFULL) return;
  for (int i = L.nb; i >= 1; --i) {
    const JointDesc d = jd[i];
    if (d.parent == 0) continue;
    const char* rec = lp + (size_t)(i - 1) * JREC * pair_bytes<T>();
    T UD[6], pa[6], pc[6], R[9], t[3];
    ld6<T>(rec, JP_UD, UD);
    const T ri = ldp<T>(rec, JP_R).x;
    for (int k = 0; k < 6; ++k) pa[k] = (T)o[(i - 1) * 6 + k] - UD[k] * ri;
    const typename Vec2<T>::type cs = ldp<T>(rec, JP_CS);
    joint_xform<T>(d, rec, cs.x, cs.y, R, t);
    act_force(R, t, pa, pc);
    for (int k = 0; k < 6; ++k) o[(d.parent - 1) * 6 + k] += (double)pc[k];
  }
}

// per-instance constraint products: AtA (packed), Atb, bis_inf_norm (ik-id-description-optimized.hpp:160-170,
// :210-215).  grow_only: the single-constraint update only ever grows bis_inf_norm_ (hpp:213-215).
template <typename T>
__global__ void k_constraint_products(char* tiles, Layout L, const T* __restrict__ uni, int a_shared, int c_lo,
                                      int c_hi, int B, int grow_only)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  char* lp = lane_ptr<T>(tiles, L, b);
  char* srec = lp + (size_t)L.off_s * pair_bytes<T>();
  T bn = grow_only ? *elem_ptr<T>(srec, SP_BI, 0) : T(0);
  for (int c = c_lo; c < c_hi; ++c) {
    char* crec = lp + (size_t)(L.off_c + c * L.crec) * pair_bytes<T>();
    T Al[36], bl[6];
    for (int k = 0; k < 36; ++k) Al[k] = a_shared ? uni[c * 36 + k] : *elem_ptr<T>(crec, CP_A + k / 2, k & 1);
    for (int k = 0; k < 6; ++k) {
      bl[k] = *elem_ptr<T>(crec, CP_B + k / 2, k & 1);
      bn = tmax(bn, tabs(bl[k]));
    }
    if (!a_shared) {
      for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) {
          T a = T(0);
          for (int k = 0; k < 6; ++k) a += Al[6 * k + i] * Al[6 * k + j];
          const int e = sym(i, j);
          *elem_ptr<T>(crec, CP_ATA + e / 2, e & 1) = a;
        }
    }
    for (int i = 0; i < 6; ++i) {
      T a = T(0);
      for (int k = 0; k < 6; ++k) a += Al[6 * k + i] * bl[k];
      *elem_ptr<T>(crec, CP_ATB + i / 2, i & 1) = a;
    }
  }
  *elem_ptr<T>(srec, SP_BI, 0) = bn;
}

// Editing the constraint set between solves (AddEqConstraint / RemoveEqConstraint, ik-id-description-optimized.hpp:244-319).
// shift != 0: records [c_lo, c_hi) of every instance <- the records one slot up (an erased entry: the later ones move down,
// hpp:305-309, and their duals yis/Aty travel with them); the last record of the range -- with shift == 0 every record of it --
// becomes the NULL constraint: A = 0, b = 0, y = 0.  A null constraint adds mu_eq * 0 to H_i, 0 to p_i, 0 to every norm and
// keeps y = 0: the slot is there for the kernels' fixed layout and changes no number.
template <typename T>
__global__ void k_edit_constraints(char* tiles, Layout L, int c_lo, int c_hi, int shift, int B)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  char* lp = lane_ptr<T>(tiles, L, b);
  for (int c = c_lo; c < c_hi; ++c) {
    char* dst = lp + (size_t)(L.off_c + c * L.crec) * pair_bytes<T>();
    const char* src = dst + (size_t)L.crec * pair_bytes<T>();
    const bool copy = shift && c + 1 < c_hi;
    for (int k = 0; k < L.crec; ++k)
      for (int h = 0; h < 2; ++h) *elem_ptr<T>(dst, k, h) = copy ? *elem_ptr<T>(const_cast<char*>(src), k, h) : T(0);
  }
}

// resets (one wavefront per tile), bits of `what`:
enum : int {
  RS_DATA_COLD = 1,   // IkIdData::Reset(false): w, z, nu, vis, fis, fis_diff_plus_Aty = 0  (data-optimized.hxx:117-126)
  RS_RECURSION = 2,   // IkIdData::ResetRecursion: w, z, vis, fis, g, yis, Aty = 0; nu and Stf_plus_w kept (hxx:138-154)
  RS_SOLVER = 4,      // ResetSolver: iter = 0, flags = 0, mu = mu0, logged scalars = 0 (optimized.hpp:168-186)
  RS_Y = 8,           // FwdPassInit cold start: yis = 0, Aty = 0 (optimized.hxx:270-278)
  RS_HCACHE = 16,     // invalidate the H/UDinv/Dinv cache
  RS_MU = 32,         // mu = mu0 only (the MAXEIGENVALUE rule's starting value, known once the solve's references are)
};
template <typename T>
__device__ __forceinline__ void reset_tile(char* tiles, const Layout& L, int what, T mu0, int tile, int lane)
{
  char* lp = tiles + (size_t)tile * L.tile_pairs * pair_bytes<T>() + (size_t)lane * 2 * sizeof(T);
  if (what & (RS_DATA_COLD | RS_RECURSION)) {
    for (int j = 0; j < L.nb; ++j) {
      char* rec = lp + (size_t)j * JREC * pair_bytes<T>();
      for (int p = JP_V; p < JP_WZ; ++p) stp<T>(rec, p, T(0), T(0));
      stp<T>(rec, JP_WZ, T(0), T(0));
      if (what & RS_DATA_COLD) st_lo<T>(rec, JP_NUS, T(0));
    }
  }
  if (what & (RS_RECURSION | RS_Y)) {
    for (int c = 0; c < L.nc; ++c) {
      char* crec = lp + (size_t)(L.off_c + c * L.crec) * pair_bytes<T>();
      for (int p = CP_Y; p < CP_B; ++p) stp<T>(crec, p, T(0), T(0));
    }
  }
  char* srec = lp + (size_t)L.off_s * pair_bytes<T>();
  if (what & RS_SOLVER) {
    stp<T>(srec, SP_MU, mu0, T(0));
    st_hi<T>(srec, SP_BI, T(0));
    stp<T>(srec, SP_ST, T(0), T(0));
    for (int p = SP_SCAL; p < SREC; ++p) stp<T>(srec, p, T(0), T(0));
  }
  if (what & RS_SOLVER) stp<T>(srec, SP_FLIP, T(0), T(0));
  if (what & RS_MU) stp<T>(srec, SP_MU, mu0, T(0));
  if (what & RS_HCACHE) stp<T>(srec, SP_TAG, T(-1), T(0));
}
template <typename T>
__global__ void __launch_bounds__(WAVE) k_reset(char* tiles, Layout L, int what, T mu0)
{
  reset_tile<T>(tiles, L, what, mu0, (int)blockIdx.x, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// lane compaction: physical repack of the live instances of one buffer set into the first slots of another
// (dense wavefronts again), and return of the finished ones to their home slot.  One source wavefront (tile)
// per workgroup; `wave_off[w]` = exclusive prefix sum of the live-lane counts (host-side scan of `wave_live`).
// Only the persistent pairs travel; the inter-sweep temporaries are rebuilt (empty H-cache tags force the H sweep).
// ------------------------------------------------------------------------------------------------
struct MovePlan {
  char* src;
  char* dst_live;
  char* dst_home;        // nullptr when the source IS the home set
  Layout L;
  int n_src;             // slots in use in the source set
  int move_bounds;       // JP_LBUB travels too (per-instance box)
  const int* map_src;    // slot -> instance id in the source set (nullptr: identity, source is home)
  int* map_dst;          // slot -> instance id in the destination set
  const int* wave_off;   // [source tiles]
  int force_home;        // 1: every slot of the source set goes home (end of the solve)
};

template <typename T>
__global__ void __launch_bounds__(WAVE) k_move(const MovePlan M)
{
  const Layout& L = M.L;
  const int lane = threadIdx.x;
  const int b = blockIdx.x * WAVE + lane;
  const bool inb = b < M.n_src;
  char* sp = lane_ptr<T>(M.src, L, b);
  char* ssrec = sp + (size_t)L.off_s * pair_bytes<T>();
  const int status = inb ? (int)ldp<T>(ssrec, SP_ST).x : ST_DONE;
  const bool live = inb && !M.force_home && !(status & ST_DONE);
  const unsigned long long mask = __ballot(live);
  const int rank = __popcll(mask & ((1ull << lane) - 1ull));
  const int inst = inb ? (M.map_src ? M.map_src[b] : b) : 0;
  const int dst = M.wave_off[blockIdx.x] + rank;
  const bool to_home = inb && !live && M.dst_home != nullptr;
  if (live) M.map_dst[dst] = inst;
  if (!live && !to_home) return;
  char* dp = live ? lane_ptr<T>(M.dst_live, L, dst) : lane_ptr<T>(M.dst_home, L, inst);
  for (int j = 0; j < L.nb; ++j) {
    const char* sr = sp + (size_t)j * JREC * pair_bytes<T>();
    char* dr = dp + (size_t)j * JREC * pair_bytes<T>();
#pragma unroll
    for (int p = 0; p < JP_NPERSIST; ++p) {
      const typename Vec2<T>::type v = ldp<T>(sr, p);
      stp<T>(dr, p, v.x, v.y);
    }
    if (M.move_bounds) {
      const typename Vec2<T>::type v = ldp<T>(sr, JP_LBUB);
      stp<T>(dr, JP_LBUB, v.x, v.y);
    }
  }
  for (int p = L.off_c; p < L.tile_pairs; ++p) {
    const typename Vec2<T>::type v = ldp<T>(sp, p);
    stp<T>(dp, p, v.x, v.y);
  }
  // the UDinv/Dinv cache did not travel
  stp<T>(dp + (size_t)L.off_s * pair_bytes<T>(), SP_TAG, T(-1), T(0));
}

}  // namespace loikb
