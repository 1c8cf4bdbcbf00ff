/*
 * oracle/loik_ref.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Single-instance, array-of-structs, fp64 restatement of the reference's
 * `FirstOrderLoikOptimizedTpl<double>` ADMM differential-IK solver
 * (reference: include/loik/loik-loid-optimized.{hpp,hxx},
 *  include/loik/loik-loid-data-optimized.hxx,
 *  include/loik/ik-id-description-optimized.hpp,
 *  include/loik/task-solver-base.hpp).
 *
 * PARITY UNPINNED: the reference ships no golden vectors / known-answer tests and
 * cannot be built in this image (Pinocchio 3.0.0, Eigen 3.4.0, Boost 1.84 are absent,
 * pixi.lock:94,104,167).  The Pinocchio primitives used by the path (`calc`, `calc_aba`,
 * `SE3actOn`, `SE3::act`, `SE3::actInv`) are restated from their published algorithms.
 * What pins this file is (a) the reference's own *relational* test strategy -- the
 * recursive solver must agree with the dense-QP "plain" solver at 1e-10 abs-or-rel
 * (tests/loik-loid.cpp:39-83, :305-556) which oracle/loik_dense.py reproduces, and
 * (b) first-principles KKT checks (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef LOIK_REF_H
#define LOIK_REF_H

#ifdef __cplusplus
extern "C" {
#endif

/* joint types (Pinocchio names in comments).  The multi-DoF ones are the joints whose motion subspace is a constant
 * selection of the columns of I6 */
enum {
  REF_J_NONE = 0, /* universe */
  REF_J_RX = 1,   /* JointModelRX */
  REF_J_RY = 2,
  REF_J_RZ = 3,
  REF_J_PX = 4,   /* JointModelPX */
  REF_J_PY = 5,
  REF_J_PZ = 6,
  REF_J_RU = 7,   /* JointModelRevoluteUnaligned  */
  REF_J_PU = 8,   /* JointModelPrismaticUnaligned */
  REF_J_FREEFLYER = 9,   /* JointModelFreeFlyer:   nq 7 (t, quat xyzw), nv 6, S = I6               */
  REF_J_SPHERICAL = 10,  /* JointModelSpherical:   nq 4 (quat xyzw),    nv 3, S = [0; I3]          */
  REF_J_TRANSLATION = 11,/* JointModelTranslation: nq 3,                nv 3, S = [I3; 0]          */
  REF_J_SPHERICAL_ZYX = 12, /* JointModelSphericalZYX: nq 3 (z, y, x angles), nv 3, S(q) angular: R = Rz Ry Rx */
  REF_J_PLANAR = 13,     /* JointModelPlanar: nq 4 (x, y, cos, sin), nv 3 (vx, vy, wz in the joint frame)    */
  REF_J_RUBX = 14,       /* JointModelRUBX / RUBY / RUBZ: nq 2 (cos, sin), nv 1                               */
  REF_J_RUBY = 15,
  REF_J_RUBZ = 16,
  REF_J_COMPOSITE = 17,  /* JointModelComposite of any sub-joints but composites, ref_model.comp_* (nv <= 6)                */
  REF_J_RUBU = 18,       /* JointModelRevoluteUnboundedUnaligned: nq 2 (cos, sin), nv 1, about ref_model.axis          */
  REF_J_HX = 19,         /* JointModelHelicalX / Y / Z / Unaligned: nq 1, nv 1; M(q) = (Rot(axis, q), pitch q axis),      */
  REF_J_HY = 20,         /* S = [pitch axis; axis]; pitch = ref_model.pitch[i]                                             */
  REF_J_HZ = 21,
  REF_J_HU = 22
};

/* mirrors enum ADMMPenaltyUpdateStrat, task-solver-base.hpp:13-18 */
enum { REF_MU_DEFAULT = 0, REF_MU_OSQP = 1, REF_MU_MAXEIGENVALUE = 3 };

/* error codes = the reference's throw sites */
enum {
  REF_OK = 0,
  REF_ERR_EQ_DIM = -1,          /* ik-id-description-optimized.hpp:41-44   */
  REF_ERR_EQ_SIZE = -2,         /* :132-145                                */
  REF_ERR_INEQ_DIM = -3,        /* :328-335                                */
  REF_ERR_NO_SUCH_CONSTRAINT = -4, /* :184-186                             */
  REF_ERR_DUP_CONSTRAINT = -5,  /* :197-199                                */
  REF_ERR_MU_STRAT = -6,        /* loik-loid-optimized.hxx:632-640         */
  REF_ERR_ARG = -7,
  REF_ERR_REFS_SIZE = -8        /* ik-id-description-optimized.hpp:105-107 */
};

typedef struct ref_model {
  int njoints;             /* incl. universe joint 0 */
  int nq, nv;
  const int *parents;      /* [nj], parents[i] < i                                   */
  const int *jtype;        /* [nj]                                                   */
  const double *axis;      /* [nj][3] unit axis (used by RU/PU; e_k for aligned)     */
  const int *idx_q;        /* [nj]                                                   */
  const int *idx_v;        /* [nj]                                                   */
  const double *placement; /* [nj][12] jointPlacements: R row-major (9) then t (3)   */
  const int *massless;     /* [nj] or NULL.  1 = the link carries no cost of its own (no rho I + H_ref, no
                              H_ref v_ref term): the intermediate bodies of a multi-DoF joint written as a chain
                              of 1-DoF joints.  Not a reference concept -- it exists so that the tests can prove
                              that such a chain reproduces the multi-DoF joint (the device's representation).   */
  /* JointModelComposite (NULL when the model has none): joint i of type REF_J_COMPOSITE = the sub-joints
     comp_first[i] .. + comp_count[i] - 1: type (1-DoF), axis, placement relative to the previous sub-joint
     (JointModelComposite::addJoint(jmodel, placement)) */
  const int *comp_first, *comp_count;   /* [nj]                 */
  const int *comp_jtype;                /* [n_sub]              */
  const double *comp_axis;              /* [n_sub][3]           */
  const double *comp_placement;         /* [n_sub][12]          */
  const double *pitch;                  /* [nj] or NULL: JointModelHelical*::m_pitch */
  const double *comp_pitch;             /* [n_sub] or NULL: the same for helical sub-joints of composites */
} ref_model;

typedef struct ref_params {
  int max_iter;
  double tol_abs, tol_rel, tol_primal_inf, tol_dual_inf;
  double rho, mu, mu_equality_scale_factor;
  int mu_update_strat;
  int num_eq_c, eq_c_dim;
  int warm_start;
  double tol_tail_solve;
  int eq_c_capacity;       /* constraint slots to allocate (0 = num_eq_c): room for AddEqConstraint, see ref_create */
} ref_params;

typedef struct ref_solver ref_solver;

/* ctor = IkIdDataTypeOptimizedTpl(model,num_eq_c) + FirstOrderLoikOptimizedTpl(...)
 * (loik-loid-data-optimized.hxx:40-104, loik-loid-optimized.hpp:129-162) */
int ref_create(const ref_model *model, const ref_params *prm, ref_solver **out);
void ref_destroy(ref_solver *s);

/* loik-loid-optimized.hpp:335-361 */
int ref_solve_init(ref_solver *s, const double *q, const double *H_ref /*[36] row-major*/,
                   const double *v_ref /*[6]*/, const int *c_ids /*[nc]*/, int nc,
                   const double *Ais /*[nc][36]*/, const double *bis /*[nc][6]*/,
                   const double *lb, const double *ub, int nbound);
/* loik-loid-optimized.hpp:368-455 */
int ref_solve(ref_solver *s);
/* loik-loid-optimized.hpp:475-580 */
int ref_solve_full(ref_solver *s, const double *q, const double *H_ref, const double *v_ref,
                   const int *c_ids, int nc, const double *Ais, const double *bis,
                   const double *lb, const double *ub, int nbound);
/* loik-loid-optimized.hpp:596-695 */
int ref_solve_tailored(ref_solver *s, const double *q, int c_id, const double *Ai, const double *bi);

/* IkProblemFormulationOptimized's editing methods (ik-id-description-optimized.hpp).  The solver class keeps problem_
 * protected, so upstream they are reachable from a subclass only; they act between SolveInit / Solve calls:
 *   ref_update_references     UpdateReferences(H_refs, v_refs), :103-121  ([nj][36], [nj][6]; n must be nj)
 *   ref_update_eq_constraint  UpdateEqConstraint(c_id, Ai, bi) :178-218, (c_id, bi) :224-238 (Ai == NULL)
 *   ref_add_eq_constraint     AddEqConstraint, :244-286 ("deactivated for now" upstream; needs eq_c_capacity)
 *   ref_remove_eq_constraint  RemoveEqConstraint, :292-319 (returns 1 when there was nothing to remove)             */
int ref_update_references(ref_solver *s, const double *H_refs, const double *v_refs, int n);
int ref_update_eq_constraint(ref_solver *s, int c_id, const double *Ai, const double *bi);
int ref_add_eq_constraint(ref_solver *s, int c_id, const double *Ai, const double *bi);
int ref_remove_eq_constraint(ref_solver *s, int c_id);
int ref_num_eq_c(const ref_solver *s);
/* LoikSolverInfo lists of the last solve (hpp:406-420): 0 primal_residual_task, 1 _slack, 2 primal_residual, 3 dual_residual_nu,
 * 4 dual_residual_v, 5 dual_residual, 6 mu, 7 mu_eq, 8 mu_ineq; *n = entries (iterations of the main loop) */
const double *ref_solver_info(const ref_solver *s, int list, int *n);
int ref_active_id(const ref_solver *s, int c);

/* pass-level entry points (loik-loid-optimized.hpp:192-264) */
void ref_fwd_pass_init(ref_solver *s, const double *q);
void ref_update_prev(ref_solver *s);
void ref_reset_inf_norms(ref_solver *s);
void ref_fwd_pass1(ref_solver *s);
void ref_bwd_pass(ref_solver *s);
void ref_fwd_pass2(ref_solver *s);
void ref_box_proj(ref_solver *s);
void ref_dual_update(ref_solver *s);
void ref_compute_residuals(ref_solver *s);
void ref_check_convergence(ref_solver *s);
void ref_check_feasibility(ref_solver *s);
int ref_update_mu(ref_solver *s);
/* one full ADMM iteration body (UpdatePrev .. ComputeResiduals), no control logic */
void ref_iteration_body(ref_solver *s);

/* field access for tests: returns pointer + length (doubles) or NULL */
enum {
  REF_F_LIMI = 0,   /* [nj][12]  */
  REF_F_OMI,        /* [nj][12]  */
  REF_F_VIS,        /* [nj][6]   */
  REF_F_VIS_PREV,   /* [nj][6]   */
  REF_F_FIS,        /* [nj][6]   */
  REF_F_HIS,        /* [nj][36]  */
  REF_F_PIS,        /* [nj][6]   */
  REF_F_NU,         /* [nv]      */
  REF_F_Z,          /* [nv]      */
  REF_F_W,          /* [nv]      */
  REF_F_YIS,        /* [nc][6]   */
  REF_F_ATY,        /* [nc][6]   */
  REF_F_G,          /* fis_diff_plus_Aty [nj][6] */
  REF_F_STF_PLUS_W, /* [nv]      */
  REF_F_R_VEC,      /* r [nv]    */
  REF_F_UDINV,      /* [nj][6]: first column (all of it for a 1-DoF joint) */
  REF_F_DINV,       /* [nj]:    [0][0] entry                               */
  REF_F_PRIMAL_RES_VEC, /* [6nb+nv] */
  REF_F_DUAL_RES_VEC,   /* [6nb+nv] */
  REF_F_DELTA_W,    /* [nv] */
  REF_F_HIS_ABA,    /* [nj][36] */
  REF_F_PIS_ABA,    /* [nj][6]  */
  REF_F_UDINV_FULL, /* [nj][36]: 6 x nv_i, column c at [6c, 6c+6)                 */
  REF_F_DINV_FULL,  /* [nj][36]: nv_i x nv_i row-major in the leading entries      */
  REF_F_COUNT
};
const double *ref_field(const ref_solver *s, int field, int *len);

/* scalar getters (task-solver-base.hpp:87-141, loik-loid-optimized.hpp:698-755) */
enum {
  REF_S_ITER = 0,
  REF_S_CONVERGED,
  REF_S_PRIMAL_INFEASIBLE,
  REF_S_DUAL_INFEASIBLE,
  REF_S_PRIMAL_RESIDUAL,
  REF_S_DUAL_RESIDUAL,
  REF_S_PRIMAL_RESIDUAL_TASK,
  REF_S_PRIMAL_RESIDUAL_SLACK,
  REF_S_DUAL_RESIDUAL_V,
  REF_S_DUAL_RESIDUAL_NU,
  REF_S_TOL_PRIMAL,
  REF_S_TOL_DUAL,
  REF_S_MU,
  REF_S_MU_EQ,
  REF_S_MU_INEQ,
  REF_S_DELTA_X_QP_INF_NORM,
  REF_S_DELTA_Z_QP_INF_NORM,
  REF_S_DELTA_Y_QP_INF_NORM,
  REF_S_A_QP_T_DELTA_Y_QP_INF_NORM,
  REF_S_UB_QP_T_DELTA_Y_QP_PLUS,
  REF_S_LB_QP_T_DELTA_Y_QP_MINUS,
  REF_S_PRIMAL_INFEASIBILITY_COND_1,
  REF_S_PRIMAL_INFEASIBILITY_COND_2,
  REF_S_TAIL_SOLVE_ITER,
  REF_S_DELTA_FIS_INF_NORM,
  REF_S_DELTA_YIS_INF_NORM,
  REF_S_DELTA_W_INF_NORM,
  REF_S_DELTA_VIS_INF_NORM,
  REF_S_DELTA_NU_INF_NORM,
  REF_S_AV_INF_NORM,
  REF_S_NU_INF_NORM,
  REF_S_HREF_V_INF_NORM,
  REF_S_G_INF_NORM,
  REF_S_STF_PLUS_W_INF_NORM,
  REF_S_BIS_INF_NORM,
  REF_S_HV_INF_NORM,
  REF_S_COUNT
};
double ref_scalar(ref_solver *s, int which);
void ref_set_max_iter(ref_solver *s, int max_iter);
void ref_set_tols(ref_solver *s, double tol_abs, double tol_rel);
void ref_set_warm_start(ref_solver *s, int warm);

/*
 * Batch driver used ONLY as bench.py's cpu_baseline and by batch parity tests:
 * one solver object per thread (the reference object is stateful/non-reentrant,
 * loik-loid-optimized.hpp:762-765), instances split contiguously over `nthreads`.
 * All per-instance arrays are instance-major ([B][...]).  Each instance runs
 * Solve(q,H_ref,v_ref,ids,Ais,bis,lb,ub) cold (loik-loid-optimized.hpp:475-580).
 * `shared_mask` bit0: A shared ([nc][36]) ; bit1: bounds shared ([nv]).
 */
int ref_solve_batch(const ref_model *model, const ref_params *prm, int B, const double *q,
                    const double *H_ref, const double *v_ref, const int *c_ids, int nc,
                    const double *Ais, const double *bis, const double *lb, const double *ub,
                    int shared_mask, int nthreads,
                    double *z_out /*[B][nv]*/, double *nu_out /*[B][nv] or NULL*/,
                    int *iters_out /*[B]*/, int *flags_out /*[B] bit0 converged bit1 primal_inf*/,
                    double *res_out /*[B][2] or NULL*/);
/* the same with per-link references: SolveInit(...) ; UpdateReferences(H_refs [nj][36], v_refs [nj][6]) ; Solve() */
int ref_solve_batch_refs(const ref_model *model, const ref_params *prm, int B, const double *q,
                    const double *H_ref, const double *v_ref, const int *c_ids, int nc,
                    const double *Ais, const double *bis, const double *lb, const double *ub,
                    int shared_mask, int nthreads,
                    double *z_out /*[B][nv]*/, double *nu_out /*[B][nv] or NULL*/,
                    int *iters_out /*[B]*/, int *flags_out /*[B] bit0 converged bit1 primal_inf*/,
                    double *res_out /*[B][2] or NULL*/,
                         const double *H_refs, const double *v_refs);

#ifdef __cplusplus
}
#endif
#endif
