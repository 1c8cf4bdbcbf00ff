/*
 * oracle/loik_ref.c -- CPU ORACLE (test infrastructure, NOT product code).  See loik_ref.h.
 *
 * Restates, function by function, the reference's optimized solver.  Every function cites the
 * reference file:line it follows (paths relative to /root/reference/).  Pinocchio 3.0.0
 * primitives (not in the reference tree; pinned in pixi.lock:167) are restated from their
 * published algorithms in the "Pinocchio primitives" section.
 *
 * PARITY UNPINNED (no golden vectors exist upstream and the reference cannot be built here);
 * pinned relationally against oracle/loik_dense.py exactly as the reference's own tests pin the
 * optimized solver against the plain one.
 */
#include "loik_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* small fixed-size helpers.  6-vectors are [linear(3); angular(3)] (Pinocchio order).       */
/* 6x6 matrices are row-major double[36].  SE3 = R row-major [9] followed by t [3].          */
/* ------------------------------------------------------------------------------------------ */

static double inf_norm(const double *x, int n)
{
  double m = 0.0;
  for (int i = 0; i < n; ++i) {
    double a = fabs(x[i]);
    if (a > m) m = a;
  }
  return m;
}

static void cross3(const double *a, const double *b, double *o)
{
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

static void mat3_mul(const double *A, const double *B, double *C) /* C = A B */
{
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
      C[3 * i + j] = s;
    }
}

static void mat3_mul_Bt(const double *A, const double *B, double *C) /* C = A B^T */
{
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * j + k];
      C[3 * i + j] = s;
    }
}

static void mat3_vec(const double *A, const double *x, double *y)
{
  for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}

static void mat3t_vec(const double *A, const double *x, double *y)
{
  for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}

static void mat6_vec(const double *A, const double *x, double *y)
{
  for (int i = 0; i < 6; ++i) {
    double s = 0.0;
    for (int k = 0; k < 6; ++k) s += A[6 * i + k] * x[k];
    y[i] = s;
  }
}

static void mat6t_vec(const double *A, const double *x, double *y)
{
  for (int i = 0; i < 6; ++i) {
    double s = 0.0;
    for (int k = 0; k < 6; ++k) s += A[6 * k + i] * x[k];
    y[i] = s;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Pinocchio primitives (restated)                                      */
/* ------------------------------------------------------------------------------------------ */

/* SE3 composition: (R1,t1)*(R2,t2) = (R1 R2, t1 + R1 t2)   [used at loik-loid-optimized.hxx:264-265] */
static void se3_mul(const double *a, const double *b, double *o)
{
  double R[9], t[3];
  mat3_mul(a, b, R);
  mat3_vec(a, b + 9, t);
  for (int i = 0; i < 9; ++i) o[i] = R[i];
  for (int i = 0; i < 3; ++i) o[9 + i] = a[9 + i] + t[i];
}

static void se3_identity(double *o)
{
  memset(o, 0, 12 * sizeof(double));
  o[0] = o[4] = o[8] = 1.0;
}

/* SE3::act(Force): (R f_l, R f_a + t x (R f_l))        [call sites hxx:74, :212] */
static void se3_act_force(const double *M, const double *f, double *o)
{
  double l[3], a[3], c[3];
  mat3_vec(M, f, l);
  mat3_vec(M, f + 3, a);
  cross3(M + 9, l, c);
  for (int i = 0; i < 3; ++i) {
    o[i] = l[i];
    o[3 + i] = a[i] + c[i];
  }
}

/* SE3::actInv(Motion): (R^T (v_l - t x v_a), R^T v_a)   [call site hxx:125] */
static void se3_actinv_motion(const double *M, const double *v, double *o)
{
  double c[3], d[3];
  cross3(M + 9, v + 3, c);
  for (int i = 0; i < 3; ++i) d[i] = v[i] - c[i];
  mat3t_vec(M, d, o);
  mat3t_vec(M, v + 3, o + 3);
}

/*
 * pinocchio::impl::internal::SE3actOn<Scalar>::run(M, I)   [call site hxx:66]
 * = X*(M) I X(M)^-1 evaluated from the A (lin,lin), B (lin,ang) and D (ang,ang) blocks only
 * (assumes I symmetric):
 *   Ao = R A R^T ; Bo = R B R^T ; Do = R D R^T
 *   Do.row(k) += t x Bo.col(k)
 *   Co.col(k)  = t x Ao.col(k) ; Co += Bo^T ; Bo = Co^T
 *   Do.col(k) += t x Bo.col(k)
 */
static void se3_act_on(const double *M, const double *I, double *res)
{
  const double *R = M, *t = M + 9;
  double Ai[9], Bi[9], Di[9], tmp[9], Ao[9], Bo[9], Co[9], Do[9], col[3], cr[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Ai[3 * i + j] = I[6 * i + j];
      Bi[3 * i + j] = I[6 * i + 3 + j];
      Di[3 * i + j] = I[6 * (3 + i) + 3 + j];
    }
  mat3_mul(R, Ai, tmp);
  mat3_mul_Bt(tmp, R, Ao);
  mat3_mul(R, Bi, tmp);
  mat3_mul_Bt(tmp, R, Bo);
  mat3_mul(R, Di, tmp);
  mat3_mul_Bt(tmp, R, Do);

  for (int k = 0; k < 3; ++k) { /* Do.row(k) += t x Bo.col(k) */
    col[0] = Bo[k]; col[1] = Bo[3 + k]; col[2] = Bo[6 + k];
    cross3(t, col, cr);
    for (int j = 0; j < 3; ++j) Do[3 * k + j] += cr[j];
  }
  for (int k = 0; k < 3; ++k) { /* Co.col(k) = t x Ao.col(k) */
    col[0] = Ao[k]; col[1] = Ao[3 + k]; col[2] = Ao[6 + k];
    cross3(t, col, cr);
    for (int j = 0; j < 3; ++j) Co[3 * j + k] = cr[j];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Co[3 * i + j] += Bo[3 * j + i];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Bo[3 * i + j] = Co[3 * j + i];
  for (int k = 0; k < 3; ++k) { /* Do.col(k) += t x Bo.col(k) */
    col[0] = Bo[k]; col[1] = Bo[3 + k]; col[2] = Bo[6 + k];
    cross3(t, col, cr);
    for (int j = 0; j < 3; ++j) Do[3 * j + k] += cr[j];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      res[6 * i + j] = Ao[3 * i + j];
      res[6 * i + 3 + j] = Bo[3 * i + j];
      res[6 * (3 + i) + j] = Co[3 * i + j];
      res[6 * (3 + i) + 3 + j] = Do[3 * i + j];
    }
}

/* number of configuration / velocity coordinates of a joint type (JointModel::nq(), ::nv()) */
static int joint_nq(int jtype)
{
  switch (jtype) {
  case REF_J_NONE: return 0;
  case REF_J_FREEFLYER: return 7;
  case REF_J_SPHERICAL: return 4;
  case REF_J_TRANSLATION: return 3;
  case REF_J_SPHERICAL_ZYX: return 3;
  case REF_J_PLANAR: return 4;
  case REF_J_RUBX: case REF_J_RUBY: case REF_J_RUBZ: case REF_J_RUBU: return 2;
  default: return 1;
  }
}
static int joint_nv(int jtype)
{
  switch (jtype) {
  case REF_J_NONE: return 0;
  case REF_J_FREEFLYER: return 6;
  case REF_J_SPHERICAL: return 3;
  case REF_J_TRANSLATION: return 3;
  case REF_J_SPHERICAL_ZYX: return 3;
  case REF_J_PLANAR: return 3;
  default: return 1;
  }
}

/* Eigen::Quaternion::toRotationMatrix for the coefficient order (x, y, z, w) Pinocchio stores in q */
static void quat_to_rot(const double *qt, double *R)
{
  const double x = qt[0], y = qt[1], z = qt[2], w = qt[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}

/* JointModel{RX,RY,RZ,PX,PY,PZ,RevoluteUnaligned,PrismaticUnaligned,FreeFlyer,Spherical,Translation}::calc(jdata, q)
 * -> joint transform M(q) (call site hxx:263); `q` points at the joint's own segment of the configuration */
static void joint_calc(int jtype, const double *axis, const double *qs, double *M)
{
  se3_identity(M);
  if (jtype == REF_J_FREEFLYER) {
    quat_to_rot(qs + 3, M);
    M[9] = qs[0]; M[10] = qs[1]; M[11] = qs[2];
    return;
  }
  if (jtype == REF_J_SPHERICAL) { quat_to_rot(qs, M); return; }
  if (jtype == REF_J_TRANSLATION) { M[9] = qs[0]; M[10] = qs[1]; M[11] = qs[2]; return; }
  if (jtype == REF_J_SPHERICAL_ZYX) { /* JointModelSphericalZYX::calc: R = Rz(q0) Ry(q1) Rx(q2) */
    const double c0 = cos(qs[0]), s0 = sin(qs[0]), c1 = cos(qs[1]), s1 = sin(qs[1]), c2 = cos(qs[2]), s2 = sin(qs[2]);
    M[0] = c0 * c1; M[1] = c0 * s1 * s2 - s0 * c2; M[2] = c0 * s1 * c2 + s0 * s2;
    M[3] = s0 * c1; M[4] = s0 * s1 * s2 + c0 * c2; M[5] = s0 * s1 * c2 - c0 * s2;
    M[6] = -s1;     M[7] = c1 * s2;                M[8] = c1 * c2;
    return;
  }
  if (jtype == REF_J_PLANAR) { /* JointModelPlanar::calc: q = (x, y, cos, sin) */
    const double c = qs[2], s = qs[3];
    M[0] = c; M[1] = -s; M[3] = s; M[4] = c;
    M[9] = qs[0]; M[10] = qs[1];
    return;
  }
  double q = qs[0];
  double c = cos(q), s = sin(q);
  /* JointModelHelical*: the rotation of the revolute joint about the same axis; the caller adds the translation pitch q axis
   * (helical_axis / ref_fwd_pass_init) */
  if (jtype >= REF_J_HX && jtype <= REF_J_HZ) jtype = REF_J_RX + (jtype - REF_J_HX);
  else if (jtype == REF_J_HU) jtype = REF_J_RU;
  if (jtype == REF_J_RUBX || jtype == REF_J_RUBY || jtype == REF_J_RUBZ) { /* JointModelRevoluteUnbounded: q = (cos, sin) */
    c = qs[0]; s = qs[1];
    jtype = REF_J_RX + (jtype - REF_J_RUBX);
  } else if (jtype == REF_J_RUBU) { /* JointModelRevoluteUnboundedUnaligned: the same about an arbitrary axis */
    c = qs[0]; s = qs[1];
    jtype = REF_J_RU;
  }
  switch (jtype) {
  case REF_J_RX:
    M[4] = c; M[5] = -s; M[7] = s; M[8] = c;
    break;
  case REF_J_RY:
    M[0] = c; M[2] = s; M[6] = -s; M[8] = c;
    break;
  case REF_J_RZ:
    M[0] = c; M[1] = -s; M[3] = s; M[4] = c;
    break;
  case REF_J_RU: { /* Rodrigues: c I + (1-c) a a^T + s [a]x */
    double ax = axis[0], ay = axis[1], az = axis[2], c1 = 1.0 - c, tmp;
    tmp = c1 * ax * ay; M[1] = tmp - s * az; M[3] = tmp + s * az;
    tmp = c1 * ax * az; M[2] = tmp + s * ay; M[6] = tmp - s * ay;
    tmp = c1 * ay * az; M[5] = tmp - s * ax; M[7] = tmp + s * ax;
    M[0] = c1 * ax * ax + c; M[4] = c1 * ay * ay + c; M[8] = c1 * az * az + c;
    break;
  }
  case REF_J_PX: M[9] = q; break;
  case REF_J_PY: M[10] = q; break;
  case REF_J_PZ: M[11] = q; break;
  case REF_J_PU:
    M[9] = axis[0] * q; M[10] = axis[1] * q; M[11] = axis[2] * q;
    break;
  default: break;
  }
}

/* joint motion subspace S: 6 x nv_i, column c stored at S[6c .. 6c+6).  Constant for every joint type but
 * JointModelSphericalZYX, whose calc() sets it from q (`qs`, may be NULL for the others) */
static void joint_S(int jtype, const double *axis, const double *qs, double *S)
{
  memset(S, 0, 36 * sizeof(double));
  switch (jtype) {
  case REF_J_RUBX: S[3] = 1.0; break;
  case REF_J_RUBY: S[4] = 1.0; break;
  case REF_J_RUBZ: S[5] = 1.0; break;
  case REF_J_RUBU: S[3] = axis[0]; S[4] = axis[1]; S[5] = axis[2]; break;
  case REF_J_PLANAR: S[0] = 1.0; S[6 + 1] = 1.0; S[12 + 5] = 1.0; break;               /* ConstraintPlanar: vx, vy, wz */
  case REF_J_SPHERICAL_ZYX: {  /* S.angularSubspace() << -s1, 0, 1,  c1 s2, c2, 0,  c1 c2, -s2, 0 (rows) */
    const double q1 = qs ? qs[1] : 0.0, q2 = qs ? qs[2] : 0.0;
    const double c1 = cos(q1), s1 = sin(q1), c2 = cos(q2), s2 = sin(q2);
    S[3] = -s1; S[4] = c1 * s2; S[5] = c1 * c2;
    S[6 + 3] = 0.0; S[6 + 4] = c2; S[6 + 5] = -s2;
    S[12 + 3] = 1.0;
    break;
  }
  case REF_J_PX: S[0] = 1.0; break;
  case REF_J_PY: S[1] = 1.0; break;
  case REF_J_PZ: S[2] = 1.0; break;
  case REF_J_RX: case REF_J_HX: S[3] = 1.0; break;   /* (helical: the linear part pitch * axis is added by the caller) */
  case REF_J_RY: case REF_J_HY: S[4] = 1.0; break;
  case REF_J_RZ: case REF_J_HZ: S[5] = 1.0; break;
  case REF_J_HU: S[3] = axis[0]; S[4] = axis[1]; S[5] = axis[2]; break;
  case REF_J_PU: S[0] = axis[0]; S[1] = axis[1]; S[2] = axis[2]; break;
  case REF_J_RU: S[3] = axis[0]; S[4] = axis[1]; S[5] = axis[2]; break;
  case REF_J_FREEFLYER: for (int c = 0; c < 6; ++c) S[6 * c + c] = 1.0; break;        /* ConstraintIdentity */
  case REF_J_SPHERICAL: for (int c = 0; c < 3; ++c) S[6 * c + 3 + c] = 1.0; break;    /* angular block      */
  case REF_J_TRANSLATION: for (int c = 0; c < 3; ++c) S[6 * c + c] = 1.0; break;      /* linear block       */
  default: break;
  }
}

/* inverse of a placement */
static void se3_inv(const double *a, double *o)
{
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * c + r];
  for (int r = 0; r < 3; ++r) o[9 + r] = -(a[r] * a[9] + a[3 + r] * a[10] + a[6 + r] * a[11]);
}

/* inverse of a symmetric positive definite n x n matrix (n <= 6) by Cholesky, the way Pinocchio's
 * internal::PerformStYSInversion does it (Dinv.setIdentity(); StYS.llt().solveInPlace(Dinv)) */
static void spd_inverse(const double *A, int n, double *Ainv)
{
  double L[36];
  memset(L, 0, sizeof(L));
  for (int j = 0; j < n; ++j) {
    double d = A[n * j + j];
    for (int k = 0; k < j; ++k) d -= L[n * j + k] * L[n * j + k];
    d = sqrt(d);
    L[n * j + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double x = A[n * i + j];
      for (int k = 0; k < j; ++k) x -= L[n * i + k] * L[n * j + k];
      L[n * i + j] = x / d;
    }
  }
  for (int c = 0; c < n; ++c) {
    double y[6];
    for (int i = 0; i < n; ++i) { /* L y = e_c */
      double x = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) x -= L[n * i + k] * y[k];
      y[i] = x / L[n * i + i];
    }
    for (int i = n - 1; i >= 0; --i) { /* L^T x = y */
      double x = y[i];
      for (int k = i + 1; k < n; ++k) x -= L[n * k + i] * Ainv[n * k + c];
      Ainv[n * i + c] = x / L[n * i + i];
    }
  }
}

/* JointModel::calc_aba(jdata, armature, I, update_I)    [call site hxx:60-63]
 *   U = I S ; Dinv = (S^T U + diag(armature))^-1 ; UDinv = U Dinv ; if (update_I) I -= UDinv U^T
 * U, UDinv: 6 x n (column c at [6c, 6c+6)); Dinv: n x n row-major.  n = 1 is the scalar formula. */
static void joint_calc_aba(const double *S, int n, const double *armature, double *I, int update_I, double *U,
                           double *Dinv, double *UDinv)
{
  double StU[36];
  for (int c = 0; c < n; ++c) mat6_vec(I, S + 6 * c, U + 6 * c);
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      double d = 0.0;
      for (int k = 0; k < 6; ++k) d += S[6 * a + k] * U[6 * b + k];
      StU[n * a + b] = d;
    }
  for (int a = 0; a < n; ++a) StU[n * a + a] += armature[a];
  if (n == 1) Dinv[0] = 1.0 / StU[0];
  else spd_inverse(StU, n, Dinv);
  for (int c = 0; c < n; ++c)
    for (int k = 0; k < 6; ++k) {
      double x = 0.0;
      for (int j = 0; j < n; ++j) x += U[6 * j + k] * Dinv[n * j + c];
      UDinv[6 * c + k] = x;
    }
  if (update_I)
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        double x = 0.0;
        for (int c = 0; c < n; ++c) x += UDinv[6 * c + i] * U[6 * c + j];
        I[6 * i + j] -= x;
      }
}

/* ------------------------------------------------------------------------------------------ */
/* solver object = IkIdDataTypeOptimizedTpl + IkProblemFormulationOptimized + solver scalars   */
/* ------------------------------------------------------------------------------------------ */
struct ref_solver {
  /* model (copied, reference keeps `Model model_` by value, loik-loid-optimized.hpp:762) */
  int nj, nb, nq, nv, nc;
  int *parents, *jtype, *idx_q, *idx_v, *jnv, *massless;
  double *axis, *placement, *pitch;   /* pitch: [nj], zeros unless the model has helical joints */

  /* --- IkIdDataTypeOptimizedTpl members (loik-loid-data-optimized.hxx:40-86) --- */
  double *oMi, *liMi;                /* [nj][12] */
  double *jS, *jU, *jUDinvM, *jDinvM; /* JointData: S,U,UDinv [nj][36] (6 x nv_i, column-wise), Dinv [nj][36] */
  double *jUDinv, *jDinv;            /* first column / [0][0] entry of the above: the 1-DoF view ([nj][6], [nj]) */
  double *nu, *nu_prev;              /* [nv] */
  double *vis, *vis_prev;            /* [nj][6] */
  double *His, *His_aba;             /* [nj][36] */
  double *pis, *pis_aba;             /* [nj][6] */
  double *R, *r;                     /* [nv] */
  double *fis, *delta_fis;           /* [nj][6] */
  double *yis, *delta_yis;           /* [nc][6] */
  double *w, *delta_w, *z, *z_prev;  /* [nv] */
  double *Aty;                       /* [nc][6] */
  double *g, *delta_g;               /* fis_diff_plus_Aty, delta_fis_diff_plus_Aty [nj][6] */
  double *Href_v;                    /* [nj][6] */
  double *Av_minus_b;                /* [nc][6] */
  double *Stf_plus_w, *delta_Stf_plus_w; /* [nv] */
  double bT_delta_y_plus, bT_delta_y_minus;
  double Av_inf_norm, nu_inf_norm, Href_v_inf_norm, g_inf_norm, Stf_plus_w_inf_norm;
  double delta_g_inf_norm, delta_Stf_plus_w_inf_norm, delta_vis_inf_norm, delta_nu_inf_norm;
  double delta_z_inf_norm, delta_fis_inf_norm, delta_yis_inf_norm, delta_w_inf_norm;

  /* --- IkProblemFormulationOptimized members (ik-id-description-optimized.hpp:342-362) --- */
  int eq_c_dim;
  double *H_refs, *v_refs, *Hv;      /* [nj][36], [nj][6], [nj][6] */
  int *comp_first, *comp_count, *comp_jtype; double *comp_axis, *comp_placement, *comp_pitch; int n_sub; /* JointModelComposite */
  int *active_ids;                   /* [nc] */
  int nc_cap;                        /* allocated constraint slots (>= nc = nc_eq_) */
  double *Ais, *bis, *AtA, *Atb;     /* [nc][36], [nc][6], [nc][36], [nc][6] */
  double *lb, *ub;                   /* [nv] */
  double bis_inf_norm, Hv_inf_norm;
  int per_link_refs;                 /* UpdateReferences() in force (not a reference member: test bookkeeping) */

  /* --- IkIdSolverBaseTpl members (task-solver-base.hpp:145-170) --- */
  double rho, mu0, mu, mu_equality_scale_factor;
  int mu_update_strat, max_iter, iter, converged;
  double tol_abs, tol_rel, tol_primal, tol_dual, tol_primal_inf, tol_dual_inf;
  int primal_infeasible, dual_infeasible;
  double primal_residual, dual_residual;

  /* --- FirstOrderLoikOptimizedTpl members (loik-loid-optimized.hpp:762-806) --- */
  int tail_solve_iter;
  double primal_residual_task, primal_residual_slack;
  double *primal_residual_vec, *dual_residual_vec; /* [6nb+nv] */
  double dual_residual_v, dual_residual_nu;
  double delta_x_qp_inf_norm, delta_y_qp_inf_norm, A_qp_T_delta_y_qp_inf_norm;
  double ub_qp_T_delta_y_qp_plus, lb_qp_T_delta_y_qp_minus;
  int primal_infeasibility_cond_1, primal_infeasibility_cond_2;
  double mu_eq, mu_ineq;
  int warm_start;
  double tol_tail_solve;

  /* LoikSolverInfo (loik-loid-optimized.hpp:47-127): what `logging_` pushes after ComputeResiduals (hpp:406-420), kept for
     the LAST solve: log[list][k], k = iteration - 1 < n_log */
  double *log[9];
  int n_log, log_cap;
};

static double *dalloc(size_t n)
{
  return (double *)calloc(n ? n : 1, sizeof(double));
}

/* eigenvalues of a symmetric 6x6 matrix by cyclic Jacobi rotations (the matrix is overwritten; ev[0..5] unsorted) */
static void sym6_eigenvalues(double *a, double *ev)
{
  for (int sweep = 0; sweep < 50; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 6; ++q) off += a[6 * p + q] * a[6 * p + q];
    if (off < 1e-300) break;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 6; ++q) {
        const double apq = a[6 * p + q];
        if (apq == 0.0) continue;
        const double theta = (a[6 * q + q] - a[6 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 6; ++k) { /* columns p, q */
          const double akp = a[6 * k + p], akq = a[6 * k + q];
          a[6 * k + p] = c * akp - sn * akq;
          a[6 * k + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 6; ++k) { /* rows p, q */
          const double apk = a[6 * p + k], aqk = a[6 * q + k];
          a[6 * p + k] = c * apk - sn * aqk;
          a[6 * q + k] = sn * apk + c * aqk;
        }
      }
  }
  for (int k = 0; k < 6; ++k) ev[k] = a[7 * k];
}

/* ORACLE EXTENSION -- ADMMPenaltyUpdateStrat::MAXEIGENVALUE is declared upstream (task-solver-base.hpp:13-18) and throws "not
   yet implemented" (loik-loid-optimized.hxx:635-637).  Defined here (and in the device library, loik_host.hip::spectral_mu0)
   as a SPECTRAL INITIALISATION of the penalty followed by DEFAULT's decade steps: mu starts at the geometric mean of the
   extreme eigenvalues of the links' cost blocks rho I + H_ref,i (all links that carry a cost; the symmetric part of H_ref,i),
   snapped to a quarter decade (10^(k/4): the two implementations then agree whatever the last bits of their eigenvalues),
   clipped to [1e-6, 1e6]; the constructor's mu is not used (set at the start of the main loop of every Solve overload).  For the reference's fixture (H_ref = I, rho = 1e-5) that is
   mu = 1 instead of 1e-2. */
static double spectral_mu0(const ref_solver *s)
{
  double lo = 0.0, hi = 0.0;
  int any = 0;
  for (int i = 1; i < s->nj; ++i) {
    if (s->massless[i]) continue;
    double m[36], ev[6];
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c)
        m[6 * r + c] = 0.5 * (s->H_refs[36 * i + 6 * r + c] + s->H_refs[36 * i + 6 * c + r]) + (r == c ? s->rho : 0.0);
    sym6_eigenvalues(m, ev);
    for (int k = 0; k < 6; ++k) {
      if (!any || ev[k] < lo) lo = ev[k];
      if (!any || ev[k] > hi) hi = ev[k];
      any = 1;
    }
  }
  if (!any) return s->mu0;
  if (lo < s->rho) lo = s->rho; /* (an indefinite reference weight: the proximal term is what is left) */
  if (hi < lo) hi = lo;
  if (!(lo > 0.0)) return s->mu0;
  const long q = lround(4.0 * log10(sqrt(lo * hi)));
  double mu = pow(10.0, (double)q / 4.0);
  if (mu < 1e-6) mu = 1e-6;
  if (mu > 1e6) mu = 1e6;
  return mu;
}

/* IkIdSolverBaseTpl::Reset, task-solver-base.hpp:73-84 */
static void base_reset(ref_solver *s)
{
  s->iter = 0;
  s->converged = 0;
  s->primal_infeasible = 0;
  s->dual_infeasible = 0;
  s->mu = s->mu0;
}

/* FirstOrderLoikOptimizedTpl::ResetSolver, loik-loid-optimized.hpp:168-186 */
static void reset_solver(ref_solver *s)
{
  base_reset(s);
  s->tail_solve_iter = 0;
  s->delta_x_qp_inf_norm = 0.0;
  s->delta_y_qp_inf_norm = 0.0;
  s->A_qp_T_delta_y_qp_inf_norm = 0.0;
  s->ub_qp_T_delta_y_qp_plus = 0.0;
  s->lb_qp_T_delta_y_qp_minus = 0.0;
  s->primal_infeasibility_cond_1 = 0;
  s->primal_infeasibility_cond_2 = 0;
  s->mu_eq = s->mu_equality_scale_factor * s->mu;
  s->mu_ineq = s->mu;
}

/* IkProblemFormulationOptimized::Reset, ik-id-description-optimized.hpp:61-72, :369-420 */
static void problem_reset(ref_solver *s)
{
  memset(s->H_refs, 0, sizeof(double) * 36 * (size_t)s->nj);
  memset(s->v_refs, 0, sizeof(double) * 6 * (size_t)s->nj);
  memset(s->Hv, 0, sizeof(double) * 6 * (size_t)s->nj);
  s->Hv_inf_norm = 0.0;
  for (int c = 0; c < s->nc_cap; ++c) s->active_ids[c] = 0;
  memset(s->Ais, 0, sizeof(double) * 36 * (size_t)s->nc_cap);
  memset(s->bis, 0, sizeof(double) * 6 * (size_t)s->nc_cap);
  memset(s->AtA, 0, sizeof(double) * 36 * (size_t)s->nc_cap);
  memset(s->Atb, 0, sizeof(double) * 6 * (size_t)s->nc_cap);
  s->bis_inf_norm = 0.0;
  memset(s->lb, 0, sizeof(double) * (size_t)s->nv);
  memset(s->ub, 0, sizeof(double) * (size_t)s->nv);
}

int ref_create(const ref_model *m, const ref_params *p, ref_solver **out)
{
  if (!m || !p || !out) return REF_ERR_ARG;
  /* ik-id-description-optimized.hpp:41-44 */
  if (p->eq_c_dim != 6) return REF_ERR_EQ_DIM;
  ref_solver *s = (ref_solver *)calloc(1, sizeof(ref_solver));
  const int nj = m->njoints, nv = m->nv;
  /* the per-constraint arrays are sized for eq_c_capacity (>= num_eq_c) so that AddEqConstraint has room: upstream sizes
     them for num_eq_c only and AddEqConstraint ("deactivated for now", ik-id-description-optimized.hpp:242) would
     overrun yis/Aty -- the capacity is how the intended behaviour becomes well defined */
  const int nc = p->eq_c_capacity > p->num_eq_c ? p->eq_c_capacity : p->num_eq_c;
  s->nj = nj; s->nb = nj - 1; s->nq = m->nq; s->nv = nv; s->nc = p->num_eq_c; s->nc_cap = nc;
  s->parents = (int *)malloc(sizeof(int) * nj);
  s->jtype = (int *)malloc(sizeof(int) * nj);
  s->idx_q = (int *)malloc(sizeof(int) * nj);
  s->idx_v = (int *)malloc(sizeof(int) * nj);
  s->axis = dalloc(3 * nj); s->pitch = dalloc(nj);
  if (m->pitch) memcpy(s->pitch, m->pitch, sizeof(double) * nj);
  s->placement = dalloc(12 * nj);
  memcpy(s->parents, m->parents, sizeof(int) * nj);
  memcpy(s->jtype, m->jtype, sizeof(int) * nj);
  memcpy(s->idx_q, m->idx_q, sizeof(int) * nj);
  memcpy(s->idx_v, m->idx_v, sizeof(int) * nj);
  memcpy(s->axis, m->axis, sizeof(double) * 3 * nj);
  memcpy(s->placement, m->placement, sizeof(double) * 12 * nj);
  s->jnv = (int *)calloc(nj, sizeof(int));
  s->massless = (int *)calloc(nj, sizeof(int));
  for (int i = 1; i < nj; ++i) s->jnv[i] = joint_nv(s->jtype[i]);
  s->comp_first = (int *)calloc(nj, sizeof(int)); s->comp_count = (int *)calloc(nj, sizeof(int));
  s->n_sub = 0;
  if (m->comp_first && m->comp_count)
    for (int i = 1; i < nj; ++i)
      if (s->jtype[i] == REF_J_COMPOSITE) {
        s->comp_first[i] = m->comp_first[i]; s->comp_count[i] = m->comp_count[i];
        s->jnv[i] = 0;   /* sum over the sub-joints (1-DoF or multi-DoF: translation + spherical, planar ...), <= 6 */
        for (int k = 0; k < m->comp_count[i]; ++k) s->jnv[i] += joint_nv(m->comp_jtype[m->comp_first[i] + k]);
        if (m->comp_first[i] + m->comp_count[i] > s->n_sub) s->n_sub = m->comp_first[i] + m->comp_count[i];
      }
  s->comp_jtype = (int *)calloc(s->n_sub ? s->n_sub : 1, sizeof(int));
  s->comp_axis = dalloc(3 * (size_t)s->n_sub); s->comp_placement = dalloc(12 * (size_t)s->n_sub); s->comp_pitch = dalloc((size_t)s->n_sub);
  if (s->n_sub) {
    memcpy(s->comp_jtype, m->comp_jtype, sizeof(int) * s->n_sub);
    memcpy(s->comp_axis, m->comp_axis, sizeof(double) * 3 * s->n_sub);
    if (m->comp_pitch) memcpy(s->comp_pitch, m->comp_pitch, sizeof(double) * s->n_sub);
    memcpy(s->comp_placement, m->comp_placement, sizeof(double) * 12 * s->n_sub);
  }
  if (m->massless) memcpy(s->massless, m->massless, sizeof(int) * nj);
  (void)joint_nq;

  /* data ctor, loik-loid-data-optimized.hxx:40-86 */
  s->oMi = dalloc(12 * nj); s->liMi = dalloc(12 * nj);
  for (int i = 0; i < nj; ++i) { se3_identity(s->oMi + 12 * i); se3_identity(s->liMi + 12 * i); }
  s->jS = dalloc(36 * nj); s->jU = dalloc(36 * nj); s->jUDinvM = dalloc(36 * nj); s->jDinvM = dalloc(36 * nj);
  s->jUDinv = dalloc(6 * nj); s->jDinv = dalloc(nj);
  for (int i = 0; i < nj; ++i) {
    joint_S(s->jtype[i], s->axis + 3 * i, NULL, s->jS + 36 * i);
    if (s->jtype[i] >= REF_J_HX && s->jtype[i] <= REF_J_HU)   /* JointMotionSubspaceHelical: S = [pitch a; a] */
      for (int k = 0; k < 3; ++k) s->jS[36 * i + k] = s->pitch[i] * s->jS[36 * i + 3 + k];
  }
  s->nu = dalloc(nv); s->nu_prev = dalloc(nv);
  s->vis = dalloc(6 * nj); s->vis_prev = dalloc(6 * nj);
  s->His = dalloc(36 * nj); s->His_aba = dalloc(36 * nj);
  for (int i = 0; i < nj; ++i)
    for (int k = 0; k < 6; ++k) { s->His[36 * i + 7 * k] = 1.0; s->His_aba[36 * i + 7 * k] = 1.0; }
  s->pis = dalloc(6 * nj); s->pis_aba = dalloc(6 * nj);
  s->R = dalloc(nv); s->r = dalloc(nv);
  s->fis = dalloc(6 * nj); s->delta_fis = dalloc(6 * nj);
  s->yis = dalloc(6 * nc); s->delta_yis = dalloc(6 * nc);
  s->w = dalloc(nv); s->delta_w = dalloc(nv); s->z = dalloc(nv); s->z_prev = dalloc(nv);
  s->Aty = dalloc(6 * nc);
  s->g = dalloc(6 * nj); s->delta_g = dalloc(6 * nj);
  s->Href_v = dalloc(6 * nj);
  s->Av_minus_b = dalloc(6 * nc);
  s->Stf_plus_w = dalloc(nv); s->delta_Stf_plus_w = dalloc(nv);

  /* problem ctor, ik-id-description-optimized.hpp:30-59 */
  s->eq_c_dim = p->eq_c_dim;
  s->H_refs = dalloc(36 * nj); s->v_refs = dalloc(6 * nj); s->Hv = dalloc(6 * nj);
  s->active_ids = (int *)calloc(nc ? nc : 1, sizeof(int));
  s->Ais = dalloc(36 * nc); s->bis = dalloc(6 * nc); s->AtA = dalloc(36 * nc); s->Atb = dalloc(6 * nc);
  s->lb = dalloc(nv); s->ub = dalloc(nv);
  problem_reset(s);

  /* base ctor, task-solver-base.hpp:54-70 */
  s->rho = p->rho; s->mu0 = p->mu; s->mu = p->mu;
  s->mu_equality_scale_factor = p->mu_equality_scale_factor;
  s->mu_update_strat = p->mu_update_strat;
  s->max_iter = p->max_iter;
  s->tol_abs = p->tol_abs; s->tol_rel = p->tol_rel;
  s->tol_primal_inf = p->tol_primal_inf; s->tol_dual_inf = p->tol_dual_inf;
  s->tol_primal = 0.0; s->tol_dual = 0.0;
  s->primal_residual = 0.0; s->dual_residual = 0.0;

  /* solver ctor, loik-loid-optimized.hpp:144-161 */
  s->warm_start = p->warm_start;
  s->tol_tail_solve = p->tol_tail_solve;
  s->primal_residual_vec = dalloc(6 * (nj - 1) + nv);
  s->dual_residual_vec = dalloc(6 * (nj - 1) + nv);
  reset_solver(s);
  *out = s;
  return REF_OK;
}

void ref_destroy(ref_solver *s)
{
  if (!s) return;
  free(s->parents); free(s->jtype); free(s->idx_q); free(s->idx_v); free(s->axis); free(s->pitch); free(s->placement);
  free(s->jnv); free(s->massless);
  free(s->comp_first); free(s->comp_count); free(s->comp_jtype); free(s->comp_axis); free(s->comp_placement); free(s->comp_pitch);
  free(s->oMi); free(s->liMi); free(s->jS); free(s->jU); free(s->jUDinvM); free(s->jDinvM);
  free(s->jUDinv); free(s->jDinv);
  free(s->nu); free(s->nu_prev); free(s->vis); free(s->vis_prev); free(s->His); free(s->His_aba);
  free(s->pis); free(s->pis_aba); free(s->R); free(s->r); free(s->fis); free(s->delta_fis);
  free(s->yis); free(s->delta_yis); free(s->w); free(s->delta_w); free(s->z); free(s->z_prev);
  free(s->Aty); free(s->g); free(s->delta_g); free(s->Href_v); free(s->Av_minus_b);
  free(s->Stf_plus_w); free(s->delta_Stf_plus_w);
  free(s->H_refs); free(s->v_refs); free(s->Hv); free(s->active_ids);
  free(s->Ais); free(s->bis); free(s->AtA); free(s->Atb); free(s->lb); free(s->ub);
  free(s->primal_residual_vec); free(s->dual_residual_vec);
  for (int l = 0; l < 9; ++l) free(s->log[l]);
  free(s);
}

/* ------------------------------------------------------------------------------------------ */
/* IkIdDataTypeOptimizedTpl methods                                                            */
/* ------------------------------------------------------------------------------------------ */

/* Reset(warm_start), loik-loid-data-optimized.hxx:114-127 */
static void data_reset(ref_solver *s, int warm_start)
{
  if (!warm_start) {
    memset(s->w, 0, sizeof(double) * s->nv);
    memset(s->z, 0, sizeof(double) * s->nv);
    memset(s->nu, 0, sizeof(double) * s->nv);
    memset(s->vis, 0, sizeof(double) * 6 * s->nj);
    memset(s->fis, 0, sizeof(double) * 6 * s->nj);
    memset(s->g, 0, sizeof(double) * 6 * s->nj);
  }
}

/* ResetRecursion(), loik-loid-data-optimized.hxx:138-154 (nu, nu_prev, Stf_plus_w NOT reset) */
static void data_reset_recursion(ref_solver *s)
{
  memset(s->w, 0, sizeof(double) * s->nv);
  memset(s->z, 0, sizeof(double) * s->nv);
  memset(s->vis, 0, sizeof(double) * 6 * s->nj);
  memset(s->fis, 0, sizeof(double) * 6 * s->nj);
  memset(s->g, 0, sizeof(double) * 6 * s->nj);
  memset(s->yis, 0, sizeof(double) * 6 * s->nc);
  memset(s->Aty, 0, sizeof(double) * 6 * s->nc);
}

/* ResetInfNorms(), loik-loid-data-optimized.hxx:165-182 */
void ref_reset_inf_norms(ref_solver *s)
{
  s->bT_delta_y_plus = 0.0; s->bT_delta_y_minus = 0.0;
  s->Av_inf_norm = 0.0; s->nu_inf_norm = 0.0; s->Href_v_inf_norm = 0.0;
  s->g_inf_norm = 0.0; s->Stf_plus_w_inf_norm = 0.0;
  s->delta_g_inf_norm = 0.0; s->delta_Stf_plus_w_inf_norm = 0.0;
  s->delta_vis_inf_norm = 0.0; s->delta_nu_inf_norm = 0.0; s->delta_z_inf_norm = 0.0;
  s->delta_fis_inf_norm = 0.0; s->delta_yis_inf_norm = 0.0; s->delta_w_inf_norm = 0.0;
}

/* UpdatePrev(), loik-loid-data-optimized.hxx:192-197 */
void ref_update_prev(ref_solver *s)
{
  memcpy(s->vis_prev, s->vis, sizeof(double) * 6 * s->nj);
  memcpy(s->nu_prev, s->nu, sizeof(double) * s->nv);
  memcpy(s->z_prev, s->z, sizeof(double) * s->nv);
}

/* ------------------------------------------------------------------------------------------ */
/* IkProblemFormulationOptimized methods                                                       */
/* ------------------------------------------------------------------------------------------ */

/* UpdateReference, ik-id-description-optimized.hpp:78-97 (one H_ref,v_ref broadcast to all nj
 * links; Hv_inf_norm_ taken from link 0 only) */
static void problem_update_reference(ref_solver *s, const double *H_ref, const double *v_ref)
{
  s->per_link_refs = 0;
  for (int i = 0; i < s->nj; ++i) {
    memcpy(s->H_refs + 36 * i, H_ref, 36 * sizeof(double));
    memcpy(s->v_refs + 6 * i, v_ref, 6 * sizeof(double));
    mat6_vec(s->H_refs + 36 * i, s->v_refs + 6 * i, s->Hv + 6 * i);
    if (s->massless[i]) { /* test-only concept, see loik_ref.h: the link has no reference cost */
      memset(s->H_refs + 36 * i, 0, 36 * sizeof(double));
      memset(s->Hv + 6 * i, 0, 6 * sizeof(double));
    }
  }
  s->Hv_inf_norm = inf_norm(s->Hv, 6);
}

/* UpdateIneqConstraints, ik-id-description-optimized.hpp:325-339 */
static int problem_update_ineq(ref_solver *s, const double *lb, const double *ub, int n)
{
  if (n != s->nv) return REF_ERR_INEQ_DIM;
  memcpy(s->lb, lb, sizeof(double) * n);
  memcpy(s->ub, ub, sizeof(double) * n);
  return REF_OK;
}

/* AtA = A^T A, Atb = A^T b  (ik-id-description-optimized.hpp:162-163, :210-211) */
static void compute_AtA_Atb(const double *A, const double *b, double *AtA, double *Atb)
{
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) {
      double sum = 0.0;
      for (int k = 0; k < 6; ++k) sum += A[6 * k + i] * A[6 * k + j];
      AtA[6 * i + j] = sum;
    }
  }
  mat6t_vec(A, b, Atb);
}

/* UpdateEqConstraints, ik-id-description-optimized.hpp:127-171 */
static int problem_update_eq(ref_solver *s, const int *ids, int nc, const double *Ais, const double *bis)
{
  if (nc != s->nc) return REF_ERR_EQ_SIZE;
  for (int c = 0; c < nc; ++c) s->active_ids[c] = ids[c];
  memcpy(s->Ais, Ais, sizeof(double) * 36 * nc);
  memcpy(s->bis, bis, sizeof(double) * 6 * nc);
  s->bis_inf_norm = 0.0;
  for (int c = 0; c < nc; ++c) {
    compute_AtA_Atb(s->Ais + 36 * c, s->bis + 6 * c, s->AtA + 36 * c, s->Atb + 6 * c);
    double n = inf_norm(s->bis + 6 * c, 6);
    if (n > s->bis_inf_norm) s->bis_inf_norm = n;
  }
  return REF_OK;
}

/* UpdateEqConstraint(c_id, Ai, bi), ik-id-description-optimized.hpp:178-218
 * (bis_inf_norm_ only ever grows here) */
static int problem_update_eq_single(ref_solver *s, int c_id, const double *Ai, const double *bi)
{
  int found = -1, count = 0;
  for (int c = 0; c < s->nc; ++c)
    if (s->active_ids[c] == c_id) {
      if (found < 0) found = c;
      ++count;
    }
  if (found < 0) return REF_ERR_NO_SUCH_CONSTRAINT;
  if (count > 1) return REF_ERR_DUP_CONSTRAINT;
  memcpy(s->Ais + 36 * found, Ai, 36 * sizeof(double));
  memcpy(s->bis + 6 * found, bi, 6 * sizeof(double));
  compute_AtA_Atb(Ai, bi, s->AtA + 36 * found, s->Atb + 6 * found);
  double n = inf_norm(bi, 6);
  if (n > s->bis_inf_norm) s->bis_inf_norm = n;
  return REF_OK;
}


/* UpdateReferences(H_refs, v_refs), ik-id-description-optimized.hpp:103-121: one weight and one target per link
 * (index 0 = the universe, carried but never read by the passes).  Hv_inf_norm_ is NOT reset there: it keeps growing from
 * whatever UpdateReference / an earlier UpdateReferences left (quirk kept). */
int ref_update_references(ref_solver *s, const double *H_refs, const double *v_refs, int n)
{
  if (n != s->nj) return REF_ERR_REFS_SIZE;
  memcpy(s->H_refs, H_refs, sizeof(double) * 36 * (size_t)s->nj);
  memcpy(s->v_refs, v_refs, sizeof(double) * 6 * (size_t)s->nj);
  for (int i = 0; i < s->nj; ++i) {
    mat6_vec(s->H_refs + 36 * i, s->v_refs + 6 * i, s->Hv + 6 * i);
    if (s->massless[i]) { /* test-only concept, see loik_ref.h */
      memset(s->H_refs + 36 * i, 0, 36 * sizeof(double));
      memset(s->Hv + 6 * i, 0, 6 * sizeof(double));
    }
    const double n_i = inf_norm(s->Hv + 6 * i, 6);
    if (n_i > s->Hv_inf_norm) s->Hv_inf_norm = n_i;
  }
  s->per_link_refs = 1;
  return REF_OK;
}

/* UpdateEqConstraint(c_id, Ai, bi) / UpdateEqConstraint(c_id, bi) (Ai == NULL), :178-238 */
int ref_update_eq_constraint(ref_solver *s, int c_id, const double *Ai, const double *bi)
{
  if (!Ai) {
    for (int c = 0; c < s->nc; ++c)
      if (s->active_ids[c] == c_id) {
        double keep[36];
        memcpy(keep, s->Ais + 36 * c, sizeof(keep));
        return problem_update_eq_single(s, c_id, keep, bi);
      }
    return REF_ERR_NO_SUCH_CONSTRAINT;
  }
  return problem_update_eq_single(s, c_id, Ai, bi);
}

/* AddEqConstraint(c_id, Ai, bi), :244-286: present -> UpdateEqConstraint; else appended, nc_eq_++, bis_inf_norm_ grows.
 * The dual of the new constraint starts at zero (upstream would read past the end of yis: see ref_create). */
int ref_add_eq_constraint(ref_solver *s, int c_id, const double *Ai, const double *bi)
{
  for (int c = 0; c < s->nc; ++c)
    if (s->active_ids[c] == c_id) return problem_update_eq_single(s, c_id, Ai, bi);
  if (s->nc >= s->nc_cap) return REF_ERR_EQ_SIZE;
  const int c = s->nc++;
  s->active_ids[c] = c_id;
  memcpy(s->Ais + 36 * c, Ai, 36 * sizeof(double));
  memcpy(s->bis + 6 * c, bi, 6 * sizeof(double));
  compute_AtA_Atb(Ai, bi, s->AtA + 36 * c, s->Atb + 6 * c);
  memset(s->yis + 6 * c, 0, 6 * sizeof(double));
  memset(s->Aty + 6 * c, 0, 6 * sizeof(double));
  memset(s->delta_yis + 6 * c, 0, 6 * sizeof(double));
  memset(s->Av_minus_b + 6 * c, 0, 6 * sizeof(double));
  const double n = inf_norm(bi, 6);
  if (n > s->bis_inf_norm) s->bis_inf_norm = n;
  return REF_OK;
}

/* RemoveEqConstraint(c_id), :292-319: absent -> nothing (upstream warns on stderr; returns 1 here); else the entry is
 * erased from every per-constraint vector (the later ones move down), bis_inf_norm_ is recomputed, nc_eq_--.  Upstream
 * leaves Aty alone ("// Aty.erase(found_it)"): here yis/Aty move with their constraints, which is what a warm start
 * after the edit needs. */
int ref_remove_eq_constraint(ref_solver *s, int c_id)
{
  int found = -1;
  for (int c = 0; c < s->nc; ++c)
    if (s->active_ids[c] == c_id) { found = c; break; }
  if (found < 0) return 1;
  /* the link's rows of primal_residual_vec_ are only ever written while it carries a constraint (hxx:433) and
     primal_residual_ is the norm of the whole vector (hxx:498): without this line the removed constraint's last residual
     would stay in every later primal residual.  Not upstream (whose RemoveEqConstraint is deactivated and has no access to
     the solver's vector): part of making the intended behaviour well defined, like eq_c_capacity. */
  memset(s->primal_residual_vec + 6 * (c_id - 1), 0, 6 * sizeof(double));
  for (int c = found; c + 1 < s->nc; ++c) {
    s->active_ids[c] = s->active_ids[c + 1];
    memcpy(s->Ais + 36 * c, s->Ais + 36 * (c + 1), 36 * sizeof(double));
    memcpy(s->AtA + 36 * c, s->AtA + 36 * (c + 1), 36 * sizeof(double));
    memcpy(s->bis + 6 * c, s->bis + 6 * (c + 1), 6 * sizeof(double));
    memcpy(s->Atb + 6 * c, s->Atb + 6 * (c + 1), 6 * sizeof(double));
    memcpy(s->yis + 6 * c, s->yis + 6 * (c + 1), 6 * sizeof(double));
    memcpy(s->Aty + 6 * c, s->Aty + 6 * (c + 1), 6 * sizeof(double));
  }
  --s->nc;
  s->bis_inf_norm = 0.0;
  for (int c = 0; c < s->nc; ++c) {
    const double n = inf_norm(s->bis + 6 * c, 6);
    if (n > s->bis_inf_norm) s->bis_inf_norm = n;
  }
  return REF_OK;
}

const double *ref_solver_info(const ref_solver *s, int list, int *n)
{
  if (list < 0 || list >= 9) return NULL;
  if (n) *n = s->n_log;
  return s->log[list];
}

int ref_num_eq_c(const ref_solver *s) { return s->nc; }
int ref_active_id(const ref_solver *s, int c) { return (c >= 0 && c < s->nc) ? s->active_ids[c] : -1; }

/* ------------------------------------------------------------------------------------------ */
/* solver passes                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* JointModelCompositeTpl::calc(jdata, q) (pinocchio/multibody/joint/joint-composite.hxx, JointCompositeCalcZeroOrderStep):
 * the sub-joints are visited last to first; with iMlast[k] = the placement of sub-joint k's frame (before its own motion:
 * jointPlacements[k] * M_k) seen ... from the LAST sub-joint's frame,
 *     M = prod_k jointPlacements[k] * M_k(q_k)            (the composite's transform)
 *     S.cols(k) = iMlast[k+1].actInv(S_k)                  (S_k seen from the last frame; the last columns are S_{n-1} itself;
 *                                                           a multi-DoF sub-joint contributes its nv_k columns)
 * `qs` = the composite's segment of q, S: 6 x nv columns */
static void composite_calc(const ref_solver *s, int idx, const double *qs, double *M, double *S)
{
  const int first = s->comp_first[idx], n = s->comp_count[idx];
  double T[7][12];   /* T[k] = prod_{j<k} P_j M_j : frame reached before sub-joint k's placement; T[n] = M */
  double after[6][12];
  se3_identity(T[0]);
  int oq = 0;
  for (int k = 0; k < n; ++k) {
    const int st = s->comp_jtype[first + k];
    double Mk[12], PM[12];
    joint_calc(st, s->comp_axis + 3 * (first + k), qs + oq, Mk);
    if (st >= REF_J_HX && st <= REF_J_HU) {   /* helical sub-joint: translation pitch q axis */
      double Sh[36];
      joint_S(st, s->comp_axis + 3 * (first + k), NULL, Sh);
      for (int c = 0; c < 3; ++c) Mk[9 + c] = s->comp_pitch[first + k] * qs[oq] * Sh[3 + c];
    }
    se3_mul(s->comp_placement + 12 * (first + k), Mk, PM);
    se3_mul(T[k], PM, T[k + 1]);
    memcpy(after[k], T[k + 1], sizeof(double) * 12);   /* the frame in which S_k is expressed (after sub-joint k moved) */
    oq += joint_nq(st);
  }
  memcpy(M, T[n], sizeof(double) * 12);
  memset(S, 0, 36 * sizeof(double));
  int col = 0;
  oq = 0;
  for (int k = 0; k < n; ++k) {
    const int st = s->comp_jtype[first + k];
    double Sk[36], inv[12], kMlast[12];
    joint_S(st, s->comp_axis + 3 * (first + k), qs + oq, Sk);   /* nv_k columns (q-dependent for a ZYX sub-joint) */
    if (st >= REF_J_HX && st <= REF_J_HU)
      for (int c = 0; c < 3; ++c) Sk[c] = s->comp_pitch[first + k] * Sk[3 + c];   /* S = [pitch a; a] */
    se3_inv(after[k], inv);
    se3_mul(inv, T[n], kMlast);                       /* placement of the last frame seen from sub-joint k's */
    for (int c = 0; c < joint_nv(st) && col < 6; ++c, ++col)
      se3_actinv_motion(kMlast, Sk + 6 * c, S + 6 * col);   /* S_k seen from the last frame */
    oq += joint_nq(st);
  }
}

/* FwdPassInit(q), loik-loid-optimized.hxx:253-283 */
void ref_fwd_pass_init(ref_solver *s, const double *q)
{
  double M[12];
  for (int idx = 1; idx < s->nj; ++idx) {
    int parent = s->parents[idx];
    if (s->jtype[idx] == REF_J_COMPOSITE) composite_calc(s, idx, q + s->idx_q[idx], M, s->jS + 36 * idx);
    else joint_calc(s->jtype[idx], s->axis + 3 * idx, q + s->idx_q[idx], M);
    if (s->jtype[idx] >= REF_J_HX && s->jtype[idx] <= REF_J_HU) {   /* JointModelHelical*::calc: translation pitch q axis */
      const double *S = s->jS + 36 * idx, hq = s->pitch[idx] * q[s->idx_q[idx]];
      for (int k = 0; k < 3; ++k) M[9 + k] = hq * S[3 + k];
    }
    if (s->jtype[idx] == REF_J_SPHERICAL_ZYX) joint_S(s->jtype[idx], s->axis + 3 * idx, q + s->idx_q[idx], s->jS + 36 * idx);
    se3_mul(s->placement + 12 * idx, M, s->liMi + 12 * idx);
    se3_mul(s->oMi + 12 * parent, s->liMi + 12 * idx, s->oMi + 12 * idx);
  }
  if (!s->warm_start) {
    memset(s->yis, 0, sizeof(double) * 6 * s->nc);
    memset(s->Aty, 0, sizeof(double) * 6 * s->nc);
  }
}

/* FwdPass1(), loik-loid-optimized.hxx:290-338 */
void ref_fwd_pass1(ref_solver *s)
{
  for (int k = 0; k < s->nv; ++k) {
    s->R[k] = 1.0;
    s->R[k] *= s->mu_ineq;
    s->r[k] = s->w[k] - s->mu_ineq * s->z[k];
  }
  for (int idx = 1; idx < s->nj; ++idx) {
    double *Hi = s->His + 36 * idx;
    const double *H_ref = s->H_refs + 36 * idx;
    const double *Hv_i = s->Hv + 6 * idx;
    /* a massless link (test-only concept, see loik_ref.h) has no rho I + H_ref and no reference term: its
       H_refs / Hv rows are zero (problem_update_reference) and rho is dropped here */
    const double rho_i = s->massless[idx] ? 0.0 : s->rho;
    memset(Hi, 0, 36 * sizeof(double));
    for (int k = 0; k < 6; ++k) Hi[7 * k] = 1.0;
    for (int k = 0; k < 36; ++k) Hi[k] *= rho_i;
    for (int k = 0; k < 36; ++k) Hi[k] += H_ref[k];
    memcpy(s->His_aba + 36 * idx, Hi, 36 * sizeof(double));
    for (int k = 0; k < 6; ++k) {
      s->pis[6 * idx + k] = -rho_i * s->vis_prev[6 * idx + k];
      s->pis[6 * idx + k] -= Hv_i[k];
    }
    memcpy(s->pis_aba + 6 * idx, s->pis + 6 * idx, 6 * sizeof(double));
  }
  for (int c = 0; c < s->nc; ++c) {
    int c_id = s->active_ids[c];
    const double *AtA_i = s->AtA + 36 * c, *Atb_i = s->Atb + 6 * c, *Aty_i = s->Aty + 6 * c;
    for (int k = 0; k < 36; ++k) {
      s->His[36 * c_id + k] += s->mu_eq * AtA_i[k];
      s->His_aba[36 * c_id + k] += s->mu_eq * AtA_i[k];
    }
    for (int k = 0; k < 6; ++k) s->pis[6 * c_id + k] += (Aty_i[k] - s->mu_eq * Atb_i[k]);
    memcpy(s->pis_aba + 6 * c_id, s->pis + 6 * c_id, 6 * sizeof(double));
  }
}

/* BwdPassOptimizedVisitor + LoikBackwardStepVisitor::algo, loik-loid-optimized.hxx:345-354, :31-81 */
void ref_bwd_pass(ref_solver *s)
{
  double acted[36], tmp[6], f[6];
  for (int idx = s->nj - 1; idx > 0; --idx) {
    int parent = s->parents[idx];
    const double *liMi = s->liMi + 12 * idx;
    double *Hi_aba = s->His_aba + 36 * idx;
    const double *pi = s->pis + 6 * idx;
    double *pi_aba = s->pis_aba + 6 * idx;
    const double *S = s->jS + 36 * idx;
    const int iv = s->idx_v[idx], n = s->jnv[idx];
    double *UDinv = s->jUDinvM + 36 * idx;

    joint_calc_aba(S, n, s->R + iv, Hi_aba, parent > 0, s->jU + 36 * idx, s->jDinvM + 36 * idx, UDinv);
    memcpy(s->jUDinv + 6 * idx, UDinv, 6 * sizeof(double));
    s->jDinv[idx] = s->jDinvM[36 * idx];

    se3_act_on(liMi, Hi_aba, acted);
    for (int k = 0; k < 36; ++k) s->His_aba[36 * parent + k] += acted[k];
    memcpy(s->His + 36 * parent, s->His_aba + 36 * parent, 36 * sizeof(double));

    for (int c = 0; c < n; ++c) { /* jointVelocitySelector(r) += S^T p (hxx:70) */
      double Stp = 0.0;
      for (int k = 0; k < 6; ++k) Stp += S[6 * c + k] * pi[k];
      s->r[iv + c] += Stp;
    }
    for (int k = 0; k < 6; ++k) {
      double x = 0.0;
      for (int c = 0; c < n; ++c) x += UDinv[6 * c + k] * s->r[iv + c];
      tmp[k] = x;
    }
    for (int k = 0; k < 6; ++k) pi_aba[k] -= tmp[k];
    se3_act_force(liMi, pi_aba, f);
    for (int k = 0; k < 6; ++k) s->pis[6 * parent + k] += f[k];
    memcpy(s->pis_aba + 6 * parent, s->pis + 6 * parent, 6 * sizeof(double));
  }
}

/* FwdPass2OptimizedVisitor + LoikForwardStep2Visitor::algo, loik-loid-optimized.hxx:361-377, :102-163 */
void ref_fwd_pass2(ref_solver *s)
{
  double vp[6], Hv6[6], d6[6];
  memcpy(s->delta_g, s->g, sizeof(double) * 6 * s->nj);
  for (int idx = 1; idx < s->nj; ++idx) {
    int parent = s->parents[idx];
    int iv = s->idx_v[idx];
    const double *Hi = s->His + 36 * idx, *pi = s->pis + 6 * idx, *liMi = s->liMi + 12 * idx;
    const double *S = s->jS + 36 * idx, *UDinv = s->jUDinvM + 36 * idx, *Dinv = s->jDinvM + 36 * idx;
    const int nvj = s->jnv[idx];

    se3_actinv_motion(liMi, s->vis + 6 * parent, vp);
    for (int c = 0; c < nvj; ++c) { /* nu_i = -UDinv^T v' - Dinv r_i (hxx:127) */
      double udv = 0.0, dr = 0.0;
      for (int k = 0; k < 6; ++k) udv += UDinv[6 * c + k] * vp[k];
      for (int j = 0; j < nvj; ++j) dr += Dinv[nvj * c + j] * s->r[iv + j];
      s->nu[iv + c] = -udv - dr;
      if (fabs(s->nu[iv + c]) > s->nu_inf_norm) s->nu_inf_norm = fabs(s->nu[iv + c]);
    }

    for (int k = 0; k < 6; ++k) s->vis[6 * idx + k] = vp[k];
    for (int c = 0; c < nvj; ++c)
      for (int k = 0; k < 6; ++k) s->vis[6 * idx + k] += S[6 * c + k] * s->nu[iv + c];

    memcpy(s->delta_fis + 6 * idx, s->fis + 6 * idx, 6 * sizeof(double));
    mat6_vec(Hi, s->vis + 6 * idx, Hv6);
    for (int k = 0; k < 6; ++k) s->fis[6 * idx + k] = Hv6[k] + pi[k];
    for (int k = 0; k < 6; ++k) s->delta_fis[6 * idx + k] = s->fis[6 * idx + k] - s->delta_fis[6 * idx + k];
    double n = inf_norm(s->delta_fis + 6 * idx, 6);
    if (n > s->delta_fis_inf_norm) s->delta_fis_inf_norm = n;

    mat6_vec(s->H_refs + 36 * idx, s->vis + 6 * idx, s->Href_v + 6 * idx);
    n = inf_norm(s->Href_v + 6 * idx, 6);
    if (n > s->Href_v_inf_norm) s->Href_v_inf_norm = n;

    for (int k = 0; k < 6; ++k) d6[k] = s->vis[6 * idx + k] - s->vis_prev[6 * idx + k];
    n = inf_norm(d6, 6);
    /* a massless chain link (test-only concept, loik_ref.h) is not a body of the reference's model: its partial
       velocity must not enter the norm over the links */
    if (n > s->delta_vis_inf_norm && !s->massless[idx]) s->delta_vis_inf_norm = n;

    memset(s->g + 6 * idx, 0, 6 * sizeof(double)); /* hxx:370 */
  }
  double m = 0.0;
  for (int k = 0; k < s->nv; ++k) {
    double a = fabs(s->nu[k] - s->nu_prev[k]);
    if (a > m) m = a;
  }
  s->delta_nu_inf_norm = m;
}

/* BoxProj(), loik-loid-optimized.hxx:384-397 */
void ref_box_proj(ref_solver *s)
{
  double m = 0.0;
  for (int k = 0; k < s->nv; ++k) {
    double x = s->nu[k] + (1.0 / s->mu_ineq) * s->w[k];
    double lo = s->lb[k] > x ? s->lb[k] : x;   /* lb.cwiseMax(x) */
    s->z[k] = s->ub[k] < lo ? s->ub[k] : lo;   /* ub.cwiseMin(.) */
    double a = fabs(s->z[k] - s->z_prev[k]);
    if (a > m) m = a;
    s->primal_residual_vec[6 * s->nb + k] = s->nu[k] - s->z[k];
  }
  s->delta_z_inf_norm = m;
}

/* DualUpdate(), loik-loid-optimized.hxx:404-461 */
void ref_dual_update(ref_solver *s)
{
  double Av[6];
  for (int c = 0; c < s->nc; ++c) {
    int c_id = s->active_ids[c];
    const double *Ai = s->Ais + 36 * c, *bi = s->bis + 6 * c, *vi = s->vis + 6 * c_id;
    mat6_vec(Ai, vi, Av);
    for (int k = 0; k < 6; ++k) s->Av_minus_b[6 * c + k] = Av[k] - bi[k];
    for (int k = 0; k < 6; ++k) s->delta_yis[6 * c + k] = s->mu_eq * s->Av_minus_b[6 * c + k];
    for (int k = 0; k < 6; ++k) s->yis[6 * c + k] += s->delta_yis[6 * c + k];
    mat6t_vec(Ai, s->yis + 6 * c, s->Aty + 6 * c);
    double n = inf_norm(s->delta_yis + 6 * c, 6);
    if (n > s->delta_yis_inf_norm) s->delta_yis_inf_norm = n;
    for (int k = 0; k < 6; ++k) s->primal_residual_vec[6 * (c_id - 1) + k] = s->Av_minus_b[6 * c + k];
    for (int k = 0; k < 6; ++k) s->g[6 * c_id + k] = s->Aty[6 * c + k];
    double plus = 0.0, minus = 0.0;
    for (int k = 0; k < 6; ++k) {
      double dy = s->delta_yis[6 * c + k];
      plus += bi[k] * (dy > 0.0 ? dy : 0.0);
      minus += bi[k] * (dy < 0.0 ? dy : 0.0);
    }
    s->bT_delta_y_plus += plus;
    s->bT_delta_y_minus += minus;
    n = inf_norm(Av, 6);
    if (n > s->Av_inf_norm) s->Av_inf_norm = n;
  }
  for (int k = 0; k < s->nv; ++k) {
    s->delta_w[k] = s->mu_ineq * (s->nu[k] - s->z[k]);
    s->w[k] += s->delta_w[k];
  }
  s->delta_w_inf_norm = inf_norm(s->delta_w, s->nv);
}

/* BwdPass2OptimizedVisitor + LoikBackwardStep2Visitor::algo, loik-loid-optimized.hxx:468-487, :185-241 */
static void bwd_pass2(ref_solver *s)
{
  double f[6];
  memcpy(s->delta_Stf_plus_w, s->Stf_plus_w, sizeof(double) * s->nv);
  for (int idx = s->nj - 1; idx > 0; --idx) {
    int parent = s->parents[idx];
    int iv = s->idx_v[idx];
    const double *liMi = s->liMi + 12 * idx, *fi = s->fis + 6 * idx, *S = s->jS + 36 * idx;
    for (int k = 0; k < 6; ++k) s->g[6 * idx + k] += -fi[k];
    se3_act_force(liMi, fi, f);
    for (int k = 0; k < 6; ++k) s->g[6 * parent + k] += f[k];
    for (int k = 0; k < 6; ++k) s->delta_g[6 * idx + k] = s->g[6 * idx + k] - s->delta_g[6 * idx + k];
    double n = inf_norm(s->delta_g + 6 * idx, 6);
    if (n > s->delta_g_inf_norm) s->delta_g_inf_norm = n;
    n = inf_norm(s->g + 6 * idx, 6);
    if (n > s->g_inf_norm) s->g_inf_norm = n;
    for (int k = 0; k < 6; ++k)
      s->dual_residual_vec[6 * (idx - 1) + k] = s->Href_v[6 * idx + k] - s->Hv[6 * idx + k] + s->g[6 * idx + k];
    for (int c = 0; c < s->jnv[idx]; ++c) {
      double Stf = 0.0;
      for (int k = 0; k < 6; ++k) Stf += S[6 * c + k] * fi[k];
      s->Stf_plus_w[iv + c] = Stf + s->w[iv + c];
      if (fabs(s->Stf_plus_w[iv + c]) > s->Stf_plus_w_inf_norm) s->Stf_plus_w_inf_norm = fabs(s->Stf_plus_w[iv + c]);
    }
  }
  for (int k = 0; k < s->nv; ++k) s->delta_Stf_plus_w[k] = s->Stf_plus_w[k] - s->delta_Stf_plus_w[k];
  s->delta_Stf_plus_w_inf_norm = inf_norm(s->delta_Stf_plus_w, s->nv);
  for (int k = 0; k < s->nv; ++k) s->dual_residual_vec[6 * s->nb + k] = s->Stf_plus_w[k];
}

/* ComputeResiduals = ComputePrimalResiduals (hxx:494-503) + ComputeDualResiduals (hxx:510-522) */
void ref_compute_residuals(ref_solver *s)
{
  s->primal_residual = inf_norm(s->primal_residual_vec, 6 * s->nb + s->nv);
  s->primal_residual_task = inf_norm(s->primal_residual_vec, 6 * s->nb);
  s->primal_residual_slack = inf_norm(s->primal_residual_vec + 6 * s->nb, s->nv);
  bwd_pass2(s);
  s->dual_residual = inf_norm(s->dual_residual_vec, 6 * s->nb + s->nv);
  s->dual_residual_v = inf_norm(s->dual_residual_vec, 6 * s->nb);
  s->dual_residual_nu = inf_norm(s->dual_residual_vec + 6 * s->nb, s->nv);
}

static double dmax(double a, double b) { return a > b ? a : b; }

/* CheckConvergence(), loik-loid-optimized.hxx:540-565 (nu_inf_norm appears twice, as upstream) */
void ref_check_convergence(ref_solver *s)
{
  s->tol_primal = s->tol_abs +
                  s->tol_rel * dmax(dmax(s->Av_inf_norm, s->nu_inf_norm), dmax(s->bis_inf_norm, s->nu_inf_norm));
  s->tol_dual = s->tol_abs + s->tol_rel * dmax(dmax(s->Href_v_inf_norm, dmax(s->g_inf_norm, s->Stf_plus_w_inf_norm)),
                                               s->Hv_inf_norm);
  if ((s->primal_residual < s->tol_primal) && (s->dual_residual < s->tol_dual)) s->converged = 1;
}

/* CheckFeasibility(), loik-loid-optimized.hxx:572-606 */
void ref_check_feasibility(ref_solver *s)
{
  s->delta_y_qp_inf_norm = dmax(s->delta_fis_inf_norm, dmax(s->delta_yis_inf_norm, s->delta_w_inf_norm));
  s->A_qp_T_delta_y_qp_inf_norm = dmax(s->delta_g_inf_norm, s->delta_Stf_plus_w_inf_norm);
  s->primal_infeasibility_cond_1 = s->A_qp_T_delta_y_qp_inf_norm <= s->tol_primal_inf * s->delta_y_qp_inf_norm;

  s->ub_qp_T_delta_y_qp_plus = s->bT_delta_y_plus;
  double acc = 0.0;
  for (int k = 0; k < s->nv; ++k) acc += s->ub[k] * (s->delta_w[k] > 0.0 ? s->delta_w[k] : 0.0);
  s->ub_qp_T_delta_y_qp_plus += acc;
  s->lb_qp_T_delta_y_qp_minus = s->bT_delta_y_minus;
  acc = 0.0;
  for (int k = 0; k < s->nv; ++k) acc += s->lb[k] * (s->delta_w[k] < 0.0 ? s->delta_w[k] : 0.0);
  s->lb_qp_T_delta_y_qp_minus += acc;

  s->primal_infeasibility_cond_2 =
      (s->ub_qp_T_delta_y_qp_plus + s->lb_qp_T_delta_y_qp_minus) <= s->tol_primal_inf * s->delta_y_qp_inf_norm;
  if (s->primal_infeasibility_cond_1 && s->primal_infeasibility_cond_2) s->primal_infeasible = 1;
  s->delta_x_qp_inf_norm = dmax(s->delta_vis_inf_norm, s->delta_nu_inf_norm);
}

/* UpdateMu(), loik-loid-optimized.hxx:613-641 */
int ref_update_mu(ref_solver *s)
{
  if (s->mu_update_strat == REF_MU_DEFAULT || s->mu_update_strat == REF_MU_MAXEIGENVALUE) { /* (the latter: see spectral_mu0) */
    if (s->primal_residual > 10 * s->dual_residual) {
      s->mu *= 10;
      s->mu_eq = s->mu_equality_scale_factor * s->mu;
      s->mu_ineq = s->mu;
    } else if (s->dual_residual > 10 * s->primal_residual) {
      s->mu *= 0.1;
      s->mu_eq = s->mu_equality_scale_factor * s->mu;
      s->mu_ineq = s->mu;
    }
    return REF_OK;
  }
  if (s->mu_update_strat == REF_MU_OSQP) {
    /* ORACLE EXTENSION -- upstream declares ADMMPenaltyUpdateStrat::OSQP (task-solver-base.hpp:13-18) and throws
       "not yet implemented" for it (hxx:632-637).  This is OSQP's published rule (Stellato et al. 2020, sec. 5.2) on LoIK's
       quantities: mu <- mu * sqrt( (r_p / max(|Av|, |nu|, |b|)) / (r_d / max(|H_ref v|, |g|, |S^T f + w|, |H_ref v_ref|)) ),
       clipped to [1e-6, 1e6], applied only when it changes mu by more than a factor of 5 (adaptive_rho_tolerance);
       the normalisers are those of CheckConvergence (hxx:544-552).  The device implements the same expression. */
    const double np = dmax(dmax(s->Av_inf_norm, s->nu_inf_norm), s->bis_inf_norm);
    const double nd = dmax(dmax(s->Href_v_inf_norm, dmax(s->g_inf_norm, s->Stf_plus_w_inf_norm)), s->Hv_inf_norm);
    const double rp = s->primal_residual / (np + 1e-10), rd = s->dual_residual / (nd + 1e-10);
    double mu_new = s->mu * sqrt(rp / (rd + 1e-10));
    if (mu_new < 1e-6) mu_new = 1e-6;
    if (mu_new > 1e6) mu_new = 1e6;
    if (mu_new > 5.0 * s->mu || mu_new < 0.2 * s->mu) {
      s->mu = mu_new;
      s->mu_eq = s->mu_equality_scale_factor * s->mu;
      s->mu_ineq = s->mu;
    }
    return REF_OK;
  }
  return REF_ERR_MU_STRAT;
}

/* body shared by the three Solve loops and the tail solve (hpp:384-404, :292-306) */
void ref_iteration_body(ref_solver *s)
{
  ref_update_prev(s);
  ref_reset_inf_norms(s);
  ref_fwd_pass1(s);
  ref_bwd_pass(s);
  ref_fwd_pass2(s);
  ref_box_proj(s);
  ref_dual_update(s);
  ref_compute_residuals(s);
}

/* InfeasibilityTailSolve(), loik-loid-optimized.hpp:271-319 */
static void infeasibility_tail_solve(ref_solver *s)
{
  s->tail_solve_iter = 0;
  while (s->delta_x_qp_inf_norm >= s->tol_tail_solve || s->delta_z_inf_norm >= s->tol_tail_solve) {
    if (s->iter >= s->max_iter) return;
    s->iter++;
    s->tail_solve_iter++;
    ref_iteration_body(s);
    s->delta_x_qp_inf_norm = dmax(s->delta_vis_inf_norm, s->delta_nu_inf_norm);
  }
}

/* main loop, identical in the three Solve overloads (hpp:377-454, :502-579, :616-693) */
static int main_loop(ref_solver *s)
{
  if (s->mu_update_strat == REF_MU_MAXEIGENVALUE) {
    /* (at the start of the main loop, where every Solve overload has the solve's references in place -- SolveInit resets the
        solver BEFORE it stores H_ref, hpp:343-360) */
    s->mu = spectral_mu0(s);
    s->mu_eq = s->mu_equality_scale_factor * s->mu;
    s->mu_ineq = s->mu;
  }
  s->n_log = 0;
  if (s->max_iter - 1 > s->log_cap) {
    s->log_cap = s->max_iter - 1;
    for (int l = 0; l < 9; ++l) { free(s->log[l]); s->log[l] = dalloc((size_t)s->log_cap); }
  }
  for (int i = 1; i < s->max_iter; i++) {
    s->iter = i;
    ref_iteration_body(s);
    { /* if (logging_) ... push_back (hpp:406-420): mu_list_ holds the mu this iteration ran with */
      const double v[9] = {s->primal_residual_task, s->primal_residual_slack, s->primal_residual, s->dual_residual_nu,
                           s->dual_residual_v, s->dual_residual, s->mu, s->mu_eq, s->mu_ineq};
      for (int l = 0; l < 9; ++l) s->log[l][s->n_log] = v[l];
      ++s->n_log;
    }
    ref_check_convergence(s);
    if (s->iter > 1) ref_check_feasibility(s);
    if (s->converged) {
      break;
    } else if (s->primal_infeasible) {
      infeasibility_tail_solve(s);
      break;
    } else if (s->dual_infeasible) { /* never set by the optimized solver */
      infeasibility_tail_solve(s);
      break;
    }
    int rc = ref_update_mu(s);
    if (rc != REF_OK) return rc;
  }
  return REF_OK;
}

int ref_solve_init(ref_solver *s, const double *q, const double *H_ref, const double *v_ref, const int *c_ids,
                   int nc, const double *Ais, const double *bis, const double *lb, const double *ub, int nbound)
{
  problem_reset(s);
  data_reset(s, s->warm_start);
  reset_solver(s);
  problem_update_reference(s, H_ref, v_ref);
  int rc = problem_update_ineq(s, lb, ub, nbound);
  if (rc != REF_OK) return rc;
  rc = problem_update_eq(s, c_ids, nc, Ais, bis);
  if (rc != REF_OK) return rc;
  ref_fwd_pass_init(s, q);
  return REF_OK;
}

int ref_solve(ref_solver *s)
{
  data_reset_recursion(s);
  reset_solver(s);
  return main_loop(s);
}

int ref_solve_full(ref_solver *s, const double *q, const double *H_ref, const double *v_ref, const int *c_ids,
                   int nc, const double *Ais, const double *bis, const double *lb, const double *ub, int nbound)
{
  int rc = ref_solve_init(s, q, H_ref, v_ref, c_ids, nc, Ais, bis, lb, ub, nbound);
  if (rc != REF_OK) return rc;
  return main_loop(s);
}

int ref_solve_tailored(ref_solver *s, const double *q, int c_id, const double *Ai, const double *bi)
{
  data_reset(s, s->warm_start);
  reset_solver(s);
  /* c_id < 0: no constraint update (not upstream: the way to solve once AddEq/RemoveEq changed the set, e.g. to none) */
  int rc = c_id < 0 ? REF_OK : problem_update_eq_single(s, c_id, Ai, bi);
  if (rc != REF_OK) return rc;
  ref_fwd_pass_init(s, q);
  return main_loop(s);
}

/* ------------------------------------------------------------------------------------------ */
/* accessors                                                                                   */
/* ------------------------------------------------------------------------------------------ */
const double *ref_field(const ref_solver *s, int field, int *len)
{
  const double *p = NULL;
  int n = 0;
  switch (field) {
  case REF_F_LIMI: p = s->liMi; n = 12 * s->nj; break;
  case REF_F_OMI: p = s->oMi; n = 12 * s->nj; break;
  case REF_F_VIS: p = s->vis; n = 6 * s->nj; break;
  case REF_F_VIS_PREV: p = s->vis_prev; n = 6 * s->nj; break;
  case REF_F_FIS: p = s->fis; n = 6 * s->nj; break;
  case REF_F_HIS: p = s->His; n = 36 * s->nj; break;
  case REF_F_PIS: p = s->pis; n = 6 * s->nj; break;
  case REF_F_NU: p = s->nu; n = s->nv; break;
  case REF_F_Z: p = s->z; n = s->nv; break;
  case REF_F_W: p = s->w; n = s->nv; break;
  case REF_F_YIS: p = s->yis; n = 6 * s->nc; break;
  case REF_F_ATY: p = s->Aty; n = 6 * s->nc; break;
  case REF_F_G: p = s->g; n = 6 * s->nj; break;
  case REF_F_STF_PLUS_W: p = s->Stf_plus_w; n = s->nv; break;
  case REF_F_R_VEC: p = s->r; n = s->nv; break;
  case REF_F_UDINV: p = s->jUDinv; n = 6 * s->nj; break;
  case REF_F_DINV: p = s->jDinv; n = s->nj; break;
  case REF_F_UDINV_FULL: p = s->jUDinvM; n = 36 * s->nj; break;
  case REF_F_DINV_FULL: p = s->jDinvM; n = 36 * s->nj; break;
  case REF_F_PRIMAL_RES_VEC: p = s->primal_residual_vec; n = 6 * s->nb + s->nv; break;
  case REF_F_DUAL_RES_VEC: p = s->dual_residual_vec; n = 6 * s->nb + s->nv; break;
  case REF_F_DELTA_W: p = s->delta_w; n = s->nv; break;
  case REF_F_HIS_ABA: p = s->His_aba; n = 36 * s->nj; break;
  case REF_F_PIS_ABA: p = s->pis_aba; n = 6 * s->nj; break;
  default: break;
  }
  if (len) *len = n;
  return p;
}

double ref_scalar(ref_solver *s, int which)
{
  switch (which) {
  case REF_S_ITER: return s->iter;
  case REF_S_CONVERGED: return s->converged;
  case REF_S_PRIMAL_INFEASIBLE: return s->primal_infeasible;
  case REF_S_DUAL_INFEASIBLE: return s->dual_infeasible;
  case REF_S_PRIMAL_RESIDUAL: return s->primal_residual;
  case REF_S_DUAL_RESIDUAL: return s->dual_residual;
  case REF_S_PRIMAL_RESIDUAL_TASK: return s->primal_residual_task;
  case REF_S_PRIMAL_RESIDUAL_SLACK: return s->primal_residual_slack;
  case REF_S_DUAL_RESIDUAL_V: return s->dual_residual_v;
  case REF_S_DUAL_RESIDUAL_NU: return s->dual_residual_nu;
  case REF_S_TOL_PRIMAL: return s->tol_primal;
  case REF_S_TOL_DUAL: return s->tol_dual;
  case REF_S_MU: return s->mu;
  case REF_S_MU_EQ: return s->mu_eq;
  case REF_S_MU_INEQ: return s->mu_ineq;
  /* debug getters recompute like upstream, loik-loid-optimized.hpp:706-755 */
  case REF_S_DELTA_X_QP_INF_NORM:
    s->delta_x_qp_inf_norm = dmax(s->delta_vis_inf_norm, s->delta_nu_inf_norm);
    return s->delta_x_qp_inf_norm;
  case REF_S_DELTA_Z_QP_INF_NORM: return s->delta_z_inf_norm;
  case REF_S_DELTA_Y_QP_INF_NORM:
    s->delta_y_qp_inf_norm = dmax(s->delta_fis_inf_norm, dmax(s->delta_yis_inf_norm, s->delta_w_inf_norm));
    return s->delta_y_qp_inf_norm;
  case REF_S_A_QP_T_DELTA_Y_QP_INF_NORM:
    s->A_qp_T_delta_y_qp_inf_norm = dmax(s->delta_g_inf_norm, s->delta_Stf_plus_w_inf_norm);
    return s->A_qp_T_delta_y_qp_inf_norm;
  case REF_S_UB_QP_T_DELTA_Y_QP_PLUS: {
    double acc = 0.0;
    for (int k = 0; k < s->nv; ++k) acc += s->ub[k] * (s->delta_w[k] > 0.0 ? s->delta_w[k] : 0.0);
    s->ub_qp_T_delta_y_qp_plus = s->bT_delta_y_plus + acc;
    return s->ub_qp_T_delta_y_qp_plus;
  }
  case REF_S_LB_QP_T_DELTA_Y_QP_MINUS: {
    double acc = 0.0;
    for (int k = 0; k < s->nv; ++k) acc += s->lb[k] * (s->delta_w[k] < 0.0 ? s->delta_w[k] : 0.0);
    s->lb_qp_T_delta_y_qp_minus = s->bT_delta_y_minus + acc;
    return s->lb_qp_T_delta_y_qp_minus;
  }
  case REF_S_PRIMAL_INFEASIBILITY_COND_1:
    return s->A_qp_T_delta_y_qp_inf_norm <= s->tol_primal_inf * s->delta_y_qp_inf_norm;
  case REF_S_PRIMAL_INFEASIBILITY_COND_2:
    return (s->ub_qp_T_delta_y_qp_plus + s->lb_qp_T_delta_y_qp_minus) <= s->tol_primal_inf * s->delta_y_qp_inf_norm;
  case REF_S_TAIL_SOLVE_ITER: return s->tail_solve_iter;
  case REF_S_DELTA_FIS_INF_NORM: return s->delta_fis_inf_norm;
  case REF_S_DELTA_YIS_INF_NORM: return s->delta_yis_inf_norm;
  case REF_S_DELTA_W_INF_NORM: return s->delta_w_inf_norm;
  case REF_S_DELTA_VIS_INF_NORM: return s->delta_vis_inf_norm;
  case REF_S_DELTA_NU_INF_NORM: return s->delta_nu_inf_norm;
  case REF_S_AV_INF_NORM: return s->Av_inf_norm;
  case REF_S_NU_INF_NORM: return s->nu_inf_norm;
  case REF_S_HREF_V_INF_NORM: return s->Href_v_inf_norm;
  case REF_S_G_INF_NORM: return s->g_inf_norm;
  case REF_S_STF_PLUS_W_INF_NORM: return s->Stf_plus_w_inf_norm;
  case REF_S_BIS_INF_NORM: return s->bis_inf_norm;
  case REF_S_HV_INF_NORM: return s->Hv_inf_norm;
  default: return NAN;
  }
}

void ref_set_max_iter(ref_solver *s, int max_iter) { s->max_iter = max_iter; }
void ref_set_tols(ref_solver *s, double tol_abs, double tol_rel) { s->tol_abs = tol_abs; s->tol_rel = tol_rel; }
void ref_set_warm_start(ref_solver *s, int warm) { s->warm_start = warm; }

/* ------------------------------------------------------------------------------------------ */
/* batch driver (cpu_baseline + batch parity tests)                                            */
/* ------------------------------------------------------------------------------------------ */
static int solve_batch_impl(const ref_model *model, const ref_params *prm, int B, const double *q, const double *H_ref,
                            const double *v_ref, const int *c_ids, int nc, const double *Ais, const double *bis,
                            const double *lb, const double *ub, int shared_mask, int nthreads, double *z_out,
                            double *nu_out, int *iters_out, int *flags_out, double *res_out, const double *H_refs,
                            const double *v_refs)
{
  const int nq = model->nq, nv = model->nv;
  int err = REF_OK;
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    ref_solver *s = NULL;
    int rc = ref_create(model, prm, &s);
    /* iteration counts are heavy-tailed (1 % of the Talos workload runs 40x the median): hand the instances out in small
       dynamic blocks, a static contiguous partition leaves most threads idle behind the unluckiest one */
    int failed = rc != REF_OK;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (long b = 0; b < B; ++b) {
      if (failed) continue;
      const double *Ab = (shared_mask & 1) ? Ais : Ais + (size_t)b * 36 * nc;
      const double *lbb = (shared_mask & 2) ? lb : lb + (size_t)b * nv;
      const double *ubb = (shared_mask & 2) ? ub : ub + (size_t)b * nv;
      if (H_refs) { /* SolveInit, UpdateReferences (one table for the batch), Solve() */
        rc = ref_solve_init(s, q + (size_t)b * nq, H_ref, v_ref, c_ids, nc, Ab, bis + (size_t)b * 6 * nc, lbb, ubb, nv);
        if (rc == REF_OK) rc = ref_update_references(s, H_refs, v_refs, model->njoints);
        if (rc == REF_OK) rc = ref_solve(s);
      } else {
        rc = ref_solve_full(s, q + (size_t)b * nq, H_ref, v_ref, c_ids, nc, Ab, bis + (size_t)b * 6 * nc, lbb, ubb, nv);
      }
      if (rc != REF_OK) { failed = 1; continue; }
      memcpy(z_out + (size_t)b * nv, s->z, sizeof(double) * nv);
      if (nu_out) memcpy(nu_out + (size_t)b * nv, s->nu, sizeof(double) * nv);
      if (iters_out) iters_out[b] = s->iter;
      if (flags_out) flags_out[b] = (s->converged ? 1 : 0) | (s->primal_infeasible ? 2 : 0);
      if (res_out) {
        res_out[2 * b] = s->primal_residual;
        res_out[2 * b + 1] = s->dual_residual;
      }
    }
    if (rc != REF_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
      err = rc;
    }
    ref_destroy(s);
  }
  return err;
}

int ref_solve_batch(const ref_model *model, const ref_params *prm, int B, const double *q, const double *H_ref,
                    const double *v_ref, const int *c_ids, int nc, const double *Ais, const double *bis,
                    const double *lb, const double *ub, int shared_mask, int nthreads, double *z_out,
                    double *nu_out, int *iters_out, int *flags_out, double *res_out)
{
  return solve_batch_impl(model, prm, B, q, H_ref, v_ref, c_ids, nc, Ais, bis, lb, ub, shared_mask, nthreads, z_out, nu_out,
                          iters_out, flags_out, res_out, NULL, NULL);
}

/* the same with per-link references: SolveInit(q, H_ref, v_ref, ...) ; UpdateReferences(H_refs, v_refs) ; Solve() */
int ref_solve_batch_refs(const ref_model *model, const ref_params *prm, int B, const double *q, const double *H_ref,
                         const double *v_ref, const int *c_ids, int nc, const double *Ais, const double *bis,
                         const double *lb, const double *ub, int shared_mask, int nthreads, double *z_out,
                         double *nu_out, int *iters_out, int *flags_out, double *res_out, const double *H_refs,
                         const double *v_refs)
{
  return solve_batch_impl(model, prm, B, q, H_ref, v_ref, c_ids, nc, Ais, bis, lb, ub, shared_mask, nthreads, z_out, nu_out,
                          iters_out, flags_out, res_out, H_refs, v_refs);
}
