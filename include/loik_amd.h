/*
 * loik_amd.h -- C-ABI of the MI355X-native batched LoIK solver (libloik_amd.so).
 *
 * Drop-in boundary: the reference has no FFI layer; its public surface for this path is the C++ class
 * `loik::FirstOrderLoikOptimizedTpl<double>` together with the caller-owned data object
 * `loik::IkIdDataTypeOptimizedTpl<double>` (include/loik/loik-loid-optimized.hpp:22-808,
 * include/loik/loik-loid-data-optimized.hpp:62-379).  Every entry point below names the reference member
 * it replaces.  The only new concept is the batch: one handle solves `batch` independent problem instances
 * of the same kinematic tree; `batch = 1` reproduces the reference's single-instance semantics.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns an int status
 * (0 = ok, negative = one of the reference's `throw std::runtime_error` sites or a runtime failure), never
 * throws.  6-vectors are [linear; angular], 6x6 matrices are row-major, joint 0 is the universe -- exactly
 * Pinocchio's conventions.  Per-instance arrays are instance-major ("array of problems"):
 * q[batch][nq], bis[batch][nc][6], z[batch][nv] ...; the library transposes them to its wavefront-tiled
 * device layout (64 instances side by side per tile) on the GPU.
 */
#ifndef LOIK_AMD_H
#define LOIK_AMD_H

#include "loik_amd_models.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LOIKB_VERSION 602  /* round.minor: bumped whenever a struct or an entry point of this header changes */

/* ---- status codes -------------------------------------------------------------------------------- */
enum {
  LOIKB_OK = 0,
  /* reference throw sites */
  LOIKB_ERR_EQ_C_DIM = -1,        /* eq_c_dim != 6                 ik-id-description-optimized.hpp:41-44   */
  LOIKB_ERR_EQ_C_SIZE = -2,       /* #constraints != nc_eq_ (:132-145); also: no free slot for AddEqConstraint, more slots than bodies */
  LOIKB_ERR_INEQ_DIM = -3,        /* lb/ub size != nv              ik-id-description-optimized.hpp:328-335 */
  LOIKB_ERR_NO_SUCH_CONSTRAINT = -4, /* UpdateEqConstraint on an unknown link       ...hpp:184-186          */
  LOIKB_ERR_DUP_CONSTRAINT = -5,  /* same link listed twice                         ...hpp:197-199          */
  LOIKB_ERR_MU_STRATEGY = -6,     /* unknown mu update strategy (upstream also: OSQP, MAXEIGENVALUE)  loik-loid-optimized.hxx:632-640 */
  LOIKB_ERR_MODEL = -7,           /* unsupported joint type, inconsistent nq/nv/idx_q/idx_v, parents[i] >= i */
  LOIKB_ERR_REFS_SIZE = -8,       /* UpdateReferences: not one entry per joint      ...hpp:105-107          */
  /* runtime */
  LOIKB_ERR_ARG = -20,
  LOIKB_ERR_HIP = -21,            /* a HIP call failed: loikb_last_error() has the text */
  LOIKB_ERR_NO_DEVICE = -22,
  LOIKB_ERR_HREF_NOT_SYMMETRIC = -23, /* device path stores H_i as 21 symmetric entries (SE3actOn symmetrises
                                         implicitly upstream) */
  LOIKB_ERR_STATE = -24           /* Solve() before SolveInit() */
};

/* ---- ADMMPenaltyUpdateStrat, task-solver-base.hpp:13-18 -------------------------------------------- */
/* DEFAULT is the reference's rule.  OSQP is declared upstream but throws "not yet implemented" there (hxx:632-637); HERE it
 * is implemented -- OSQP's published penalty rule on LoIK's residuals and normalisers, see update_mu() in
 * loik_amd/csrc/loik_device.hpp and the identical expression in the CPU oracle -- as an extension, not a parity target: on the
 * headline workload it ends most of DEFAULT's mu limit cycles (instances hitting max_iter: 1.16 % -> 0.14 %).  mu is then
 * off the decade grid: fp64 robots of 17..64 joints run it on the flat engines (k_flat2 / k_flat1 with MUR = 1: the wavefront
 * builds W / Dinv itself at every change of mu, since round 5), everything else on k_solve / k_tail.
 * MAXEIGENVALUE, also declared and unimplemented upstream (hxx:635-637), is implemented HERE as a spectral initialisation of the
 * penalty followed by DEFAULT's decade steps: every solve starts at mu = the geometric mean of the extreme eigenvalues of the
 * links' cost blocks rho I + sym(H_ref,i) (all links that carry a cost), snapped to a quarter decade 10^(k/4), clipped to
 * [1e-6, 1e6]; the constructor's mu is not used.  (Reference fixture, H_ref = I, rho = 1e-5: mu starts at 1 instead of 1e-2 --
 * on the headline workload 89.8 % of the instances converge instead of 86.1 %, at the same mean iteration count: fewer spurious
 * infeasibility certificates.)  For the kernels it is DEFAULT with another mu0: every engine runs it.  Any other value returns
 * LOIKB_ERR_MU_STRATEGY like upstream's last branch (hxx:638-640). */
enum { LOIKB_MU_DEFAULT = 0, LOIKB_MU_OSQP = 1, LOIKB_MU_MAXEIGENVALUE = 3 };

enum { LOIKB_F64 = 0, LOIKB_F32 = 1 };

/* option flags */
enum {
  LOIKB_OPT_FIXED_ITERS = 1, /* run exactly max_iter-1 ADMM iterations, mu frozen, no stopping logic       */
  LOIKB_OPT_NO_H_CACHE = 2,  /* recompute H_i/UDinv/Dinv every iteration like upstream (default: reuse them
                                 while mu is unchanged -- bit-identical results, fewer HBM bytes)            */
  LOIKB_OPT_OWN_STREAM = 8,  /* the handle creates its own non-blocking HIP stream instead of launching on the null stream:
                                 two handles on one device then run concurrently (batches in flight from two host threads);
                                 loikb_set_stream still overrides it                                                  */
  LOIKB_OPT_NO_COMPACTION = 4, /* never repack live instances into dense wavefronts between launches (default:
                                 repack when at most 85 % of the slots are still iterating; results are
                                 bit-identical, but the inter-sweep temporaries His/pis/UDinv/Dinv/r of
                                 instances that moved are not retrievable afterwards)                         */
  LOIKB_OPT_ORDER_FROM_PREVIOUS = 32, /* The on-chip engines hand a handle's instances out longest first, by the iteration counts
                                 its previous solve had -- by default only while the inputs are the ones those counts were taken
                                 on (Solve() again: the counts are exact).  With this flag also after SolveInit / a tailored solve /
                                 UpdateEqConstraint / integrate changed them: for callers whose consecutive problems resemble each
                                 other (a planner tracking a slowly moving target).  The handle compares such launches with its
                                 last arrival-order launch and goes back to arrival order for a few solves when they are no shorter.
                                 The order only decides WHEN an instance runs: results are bit-identical either way.          */
  LOIKB_OPT_F32_ACCURATE = 16 /* LOIKB_F32 only (no fp32 exists upstream: src/loik-loid-optimized.cpp:10-13): the accuracy
                                 contract |z_f32 - z_f64|_inf <= tol_abs for 99 % of the instances that converge in both,
                                 for tol_abs >= 1e-3 (below that single precision does not resolve D_i = S^T H S + mu once mu
                                 has dropped to 1e-2: both fp32 paths end at p99 ~1e-3 -- use LOIKB_F64, +16 % time).  The
                                 link forces come from f = H v + p with the stored H of the decade (k_lean) instead of the
                                 force-balance recursion over the children, whose cancellation costs the fast path a decade
                                 (Panda-7, tol 1e-3: p99 1.8e-4 against 3.9e-3, at 1.9x the time).  Robots of more than 16
                                 joints run fp32 in k_lean either way.                                                  */
};

/* input flags of solve_init / solve_full / solve_tailored */
enum {
  LOIKB_IN_DEVICE = 1,      /* per-instance input pointers are device pointers (resident in HBM)           */
  LOIKB_A_SHARED = 2,       /* Ais is [nc][36]: one constraint matrix per constraint for the whole batch  */
  LOIKB_BOUNDS_SHARED = 4,  /* lb/ub are [nv]: one box for the whole batch                                 */
  LOIKB_B_SHARED = 8,       /* bis is [nc][6] (tailored/single-instance convenience)                        */
  LOIKB_Q_SHARED = 16       /* q is [nq]                                                                     */
};

/* output flags of loikb_get */
enum { LOIKB_OUT_DEVICE = 1 };

typedef struct loikb_options {
  /* the reference constructor's arguments, same order and meaning (loik-loid-optimized.hpp:129-134) */
  int max_iter;
  double tol_abs, tol_rel, tol_primal_inf, tol_dual_inf;
  double rho, mu, mu_equality_scale_factor;
  int mu_update_strat;
  int num_eq_c, eq_c_dim;
  int warm_start;
  double tol_tail_solve;
  int verbose, logging;
  /* batch / device */
  int batch;      /* number of problem instances                       */
  int device;     /* HIP device ordinal                                */
  int precision;  /* LOIKB_F64 (reference arithmetic) | LOIKB_F32      */
  int flags;      /* LOIKB_OPT_*                                       */
  int max_launch_iters; /* ADMM iterations per kernel launch, 0 = automatic                    */
  int compact_min_instances; /* stop compacting below this many slots, 0 = default (4096)     */
  int tail_max_instances;    /* hand the last N live instances to the cooperative tail kernel (one wavefront per
                                instance): 0 = default (2^20 when the lean kernel applies -- loikb_plan_string says
                                whether and why: whole batches run in it --, else 32768), < 0 = never           */
  int eq_c_capacity;         /* constraint slots to allocate, 0 = num_eq_c.  Room for loikb_add_eq_constraint: upstream sizes
                                yis/Aty for num_eq_c only, so its AddEqConstraint ("deactivated for now",
                                ik-id-description-optimized.hpp:242) has nowhere to put a new dual; a slot without a
                                constraint holds the null constraint (A = 0, b = 0, y = 0), which changes no number  */
} loikb_options;

typedef struct loikb_solver loikb_solver;

/* ctor: IkIdDataTypeOptimizedTpl(model,num_eq_c) + FirstOrderLoikOptimizedTpl(...)
 * (loik-loid-data-optimized.hxx:40-104, loik-loid-optimized.hpp:129-162).  The model is copied (hpp:762). */
int loikb_create(const loikb_model_desc *model, const loikb_options *opts, loikb_solver **out);
int loikb_destroy(loikb_solver *s);
/* launch on the caller's HIP stream (a hipStream_t); default is the null stream */
int loikb_set_stream(loikb_solver *s, void *hip_stream);

/* SolveInit(q,H_ref,v_ref,active_task_constraint_ids,Ais,bis,lb,ub), loik-loid-optimized.hpp:335-361 */
int loikb_solve_init(loikb_solver *s, const double *q, const double *H_ref /*[36]*/, const double *v_ref /*[6]*/,
                     const int *c_ids /*[nc]*/, int nc, const double *Ais, const double *bis, const double *lb,
                     const double *ub, int nbound, int in_flags);
/* Solve(), loik-loid-optimized.hpp:368-455: cold main loop on the problem set by SolveInit */
int loikb_solve(loikb_solver *s);
/* Solve(q,H_ref,v_ref,ids,Ais,bis,lb,ub), loik-loid-optimized.hpp:475-580 */
int loikb_solve_full(loikb_solver *s, const double *q, const double *H_ref, const double *v_ref, const int *c_ids,
                     int nc, const double *Ais, const double *bis, const double *lb, const double *ub, int nbound,
                     int in_flags);
/* Solve(q,c_id,Ai,bi), loik-loid-optimized.hpp:596-695: honours warm_start, keeps reference and bounds */
int loikb_solve_tailored(loikb_solver *s, const double *q, int c_id, const double *Ai, const double *bi,
                         int in_flags);

/* IkProblemFormulationOptimized's editing methods (ik-id-description-optimized.hpp).  Upstream the solver keeps `problem_`
 * protected (loik-loid-optimized.hpp:765), so they are reachable from a subclass only; here they are entry points.  They act
 * on the problem SolveInit set and stay in force for loikb_solve / loikb_solve_tailored until the next SolveInit
 * (whose UpdateReference / UpdateEqConstraints overwrite them, hpp:355-357).
 *   loikb_update_references     UpdateReferences(H_refs, v_refs), :103-121.  [n][36] row-major and [n][6], n = njoints incl. the
 *                               universe (else LOIKB_ERR_REFS_SIZE); entry 0 is carried and counts in Hv_inf_norm_, which this
 *                               call never resets (quirks kept).  Per-link references run in k_solve / k_tail, not in k_lean.
 *   loikb_update_eq_constraint  UpdateEqConstraint(c_id, Ai, bi) :178-218; Ai == NULL: the (c_id, bi) overload :224-238.
 *                               Ai: [36] with LOIKB_A_SHARED (must match SolveInit) else [B][36]; bi: [B][6] or [6] with
 *                               LOIKB_B_SHARED; LOIKB_IN_DEVICE as in solve_init.  bis_inf_norm_ only grows (quirk kept).
 *   loikb_add_eq_constraint     AddEqConstraint, :244-286: a link that has a constraint -> UpdateEqConstraint; else the
 *                               constraint is appended (its dual starts at zero).  LOIKB_ERR_EQ_C_SIZE without a free slot.
 *   loikb_remove_eq_constraint  RemoveEqConstraint, :292-319: the later constraints move down (with their duals),
 *                               bis_inf_norm_ is recomputed.  LOIKB_NOTHING_TO_REMOVE (= 1, not an error; upstream warns on
 *                               stderr) when the link has none.
 *   loikb_solve_tailored(s, q, -1, NULL, NULL, flags) solves on the edited set without updating a constraint (not upstream,
 *   whose tailored Solve always rewrites one).  SolveInit's count check is against the CURRENT number of constraints, like
 *   upstream's nc_eq_ (hpp:143).  loikb_get(YIS / ATY) return the active constraints, in active_ids order.               */
enum { LOIKB_NOTHING_TO_REMOVE = 1 };
int loikb_update_references(loikb_solver *s, const double *H_refs, const double *v_refs, int n);
int loikb_update_eq_constraint(loikb_solver *s, int c_id, const double *Ai, const double *bi, int in_flags);
int loikb_add_eq_constraint(loikb_solver *s, int c_id, const double *Ai, const double *bi, int in_flags);
int loikb_remove_eq_constraint(loikb_solver *s, int c_id);
int loikb_num_eq_c(const loikb_solver *s);        /* nc_eq_: constraints in force                                  */
int loikb_eq_c_capacity(const loikb_solver *s);   /* slots allocated                                               */
int loikb_active_constraint_ids(const loikb_solver *s, int *out, int cap); /* returns nc_eq_, writes min(nc_eq_, cap) ids */

/* SolverInfo / LoikSolverInfo (task-solver-base.hpp:25-52, loik-loid-optimized.hpp:47-127), filled when the handle was created
 * with options.logging = 1 ("logging residuals, should be disabled for speed", hpp:408): every Solve then runs on the plain
 * pass-by-pass implementation behind loikb_pass (or, for fp64 handles whose robot the flat engines take, on that engine's logging
 * build -- any reference weight on k_flat2 / k_flat1, H_ref = h I on k_flat) and records, per instance and main-loop iteration, what
 * upstream pushes after ComputeResiduals (hpp:406-420).  loikb_get returns the state that implementation left.
 *   out[b][k], k = iteration - 1 < rows[b]: the list of instance b ([B][out_rows_cap], zero beyond rows[b]); rows may be NULL.
 *   out_rows_cap = entries per instance the caller's buffer holds: loikb_solver_info_rows_cap() (= max_iter - 1 at the time of
 *   the solve) gives the full lists; a shorter buffer gets their first out_rows_cap entries.
 *   rows[b] = iterations of the main loop = size of every residual / mu list; iter_list_ is 1 .. get_iter(), the tail solve adds
 *   get_tail_solve_iter entries to iter_list_ / tail_solve_iter_list_ only (hpp:286-290).  The lists are those of the LAST solve
 *   (upstream never clears them in Solve(): unbounded growth, SURVEY 8(a) note 2, not replicated).                          */
enum {
  LOIKB_LOG_PRIMAL_RESIDUAL_TASK = 0, LOIKB_LOG_PRIMAL_RESIDUAL_SLACK, LOIKB_LOG_PRIMAL_RESIDUAL, LOIKB_LOG_DUAL_RESIDUAL_NU,
  LOIKB_LOG_DUAL_RESIDUAL_V, LOIKB_LOG_DUAL_RESIDUAL, LOIKB_LOG_MU, LOIKB_LOG_MU_EQ, LOIKB_LOG_MU_INEQ, LOIKB_LOG_NLIST
};
int loikb_get_solver_info(loikb_solver *s, int list, double *out, int out_rows_cap, int *rows);
int loikb_solver_info_rows_cap(const loikb_solver *s);
/* Instances of the last logged solve whose lists end before their last iteration: 0 except for a WARM-STARTED solve on the flat
 * engine's logging build in which an instance's mu left the ten precomputed decades (mu0 * 10^-2 .. 10^7) and the instance was
 * finished by the engine that writes no lists -- a cold solve in which that happens is repeated on the pass-by-pass implementation
 * instead, which logs everything; a warm start cannot be repeated (the iterates it began with are gone).                        */
int loikb_solver_info_truncated(const loikb_solver *s);

/* Outer loop on the device (the caller side of the path: a sampling planner / global IK iterates
 * solve -> integrate -> re-target, README.md:5 of the reference).  The configurations q stay
 * resident in HBM between steps: no per-step upload of q, FwdPassInit (loik-loid-optimized.hxx:253-283) runs from the
 * resident copy.
 *   loikb_integrate(s, dt):  q <- q (+) dt * z, z = the answer of the last solve (pinocchio::integrate: a plain sum
 *                            for 1-DoF and translation joints, the SE(3) / SO(3) exponential update with the
 *                            first-order re-normalised quaternion for free-flyer and spherical joints)
 *   loikb_solve_tailored(s, NULL, c_id, Ai, bi, flags):  q == NULL means "the resident q"                           */
int loikb_integrate(loikb_solver *s, double dt);
/* hipDeviceSynchronize() on the solver's device: every solve entry point already returns after its own stream drained;
 * this is the explicit bracket a timing harness (bench.py) or a caller with its own streams puts around them */
int loikb_synchronize(loikb_solver *s);

/* The PASS-LEVEL public methods of the reference (loik-loid-optimized.hpp:192-264), which its own component-wise test calls
 * one by one while reading the data object in between (tests/loik-loid.cpp:305-556).  The production kernels fuse the passes
 * and never hold the state in between, so these run on a separate, plain implementation (loik_amd/csrc/loik_passes.hpp: one
 * instance per thread, the reference's data object restated member by member) -- a debug path, and the second implementation
 * on the device the fused engines are checked against.  The first loikb_pass after SolveInit / a solve copies the solver's
 * state; from then until the next SolveInit / Solve call loikb_get serves the members from that copy, in the same layouts
 * (His, pis, r, Dinv, UDinv are exactly what the last pass left, as upstream).  Any model the solver accepts: a multi-DoF
 * joint is the chain of 1-DoF joints the engines use (per-link members: the body-carrying link; r, Dinv, UDinv: one column
 * of the chain's elimination per DoF).  The arithmetic is fp64 whatever the handle's precision.
 *   LOIKB_PASS_BEGIN_ITERATION = iter_++, ik_id_data.UpdatePrev(), ik_id_data.ResetInfNorms()   (hpp:381-388)          */
enum {
  LOIKB_PASS_BEGIN_ITERATION = 0, LOIKB_PASS_FWD_PASS1, LOIKB_PASS_BWD_PASS, LOIKB_PASS_FWD_PASS2, LOIKB_PASS_BOX_PROJ,
  LOIKB_PASS_DUAL_UPDATE, LOIKB_PASS_COMPUTE_RESIDUALS, LOIKB_PASS_CHECK_CONVERGENCE, LOIKB_PASS_CHECK_FEASIBILITY,
  LOIKB_PASS_UPDATE_MU
};
int loikb_pass(loikb_solver *s, int pass);

/* setters of IkIdSolverBaseTpl / the solver (task-solver-base.hpp:104-141, loik-loid-optimized.hpp:702-703).
 * Two deliberate differences from upstream, both flagged here because a drop-in must not surprise:
 *  - loikb_set_mu sets the INITIAL penalty mu0 of the following solves.  Upstream's set_mu writes mu_ only
 *    (task-solver-base.hpp:114), which ResetSolver() -- called by SolveInit and every Solve overload -- immediately
 *    overwrites with mu0_ (task-solver-base.hpp:73-84): upstream's setter cannot influence a solve at all.
 *  - loikb_set_tol(tol_abs, tol_rel) has no upstream counterpart (tol_abs_/tol_rel_ are constructor-only there;
 *    set_tol_primal / set_tol_dual write values that CheckConvergence recomputes every iteration, hxx:544-552).
 * set_tol_dual_inf is not exposed: dual infeasibility can never be flagged on this path.            */
int loikb_set_max_iter(loikb_solver *s, int max_iter);
int loikb_set_rho(loikb_solver *s, double rho);
int loikb_set_mu(loikb_solver *s, double mu);
int loikb_set_tol(loikb_solver *s, double tol_abs, double tol_rel);
int loikb_set_tol_primal_inf(loikb_solver *s, double tol);
int loikb_set_tol_tail_solve(loikb_solver *s, double tol);
int loikb_set_warm_start(loikb_solver *s, int warm_start);

/* results: the public members of the caller-owned IkIdData the reference's tests read
 * (tests/loik-loid.cpp:597-615) and the getters of the solver (task-solver-base.hpp:87-102,
 * loik-loid-optimized.hpp:698-755), one value per instance */
enum {
  /* double [batch][nv], Pinocchio's idx_v order (a free-flyer contributes 6 entries [linear; angular], ...) */
  LOIKB_F_Z = 0,      /* ik_id_data.z  -- the answer: box-projected joint velocity */
  LOIKB_F_NU,         /* ik_id_data.nu */
  LOIKB_F_W,          /* ik_id_data.w  */
  LOIKB_F_STF_PLUS_W, /* ik_id_data.Stf_plus_w */
  LOIKB_F_R,          /* ik_id_data.r (after the backward pass)             } inter-sweep temporaries, one entry per   */
  LOIKB_F_DINV,       /* JointData::Dinv                                    } DoF: upstream's values for 1-DoF joints; */
                      /*   for a multi-DoF joint the device eliminates its coordinates one at a time (same Schur   */
                      /*   complement), so r / Dinv / UDinv are those of that sequence, not upstream's blocks       */
  /* double [batch][nb][6], joints 1..nb */
  LOIKB_F_VIS,        /* ik_id_data.vis[i] */
  LOIKB_F_FIS,        /* ik_id_data.fis[i] */
  LOIKB_F_G,          /* ik_id_data.fis_diff_plus_Aty[i] */
  LOIKB_F_PIS,        /* ik_id_data.pis[i] */
  LOIKB_F_UDINV,      /* JointData::UDinv: [batch][nv][6], per DoF (see LOIKB_F_R) */
  /* double [batch][nb][21]: upper triangle, row-major */
  LOIKB_F_HIS,        /* ik_id_data.His[i] */
  /* double [batch][nb][12]: R row-major, t */
  LOIKB_F_LIMI,       /* ik_id_data.liMi[i] */
  /* double [batch][nc][6], nc = loikb_num_eq_c(): the constraints in force, in active_task_constraint_ids order */
  LOIKB_F_YIS,        /* ik_id_data.yis[c] */
  LOIKB_F_ATY,        /* ik_id_data.Aty[c] */
  /* int [batch] */
  LOIKB_F_ITER,               /* get_iter()                        */
  LOIKB_F_CONVERGED,          /* get_convergence_status()          */
  LOIKB_F_PRIMAL_INFEASIBLE,  /* get_primal_infeasibility_status() */
  LOIKB_F_STATUS,             /* raw bits: 1 converged, 2 primal infeasible, 4 tail solve ran, 8 finished */
  /* double [batch] */
  LOIKB_F_PRIMAL_RESIDUAL = 32, LOIKB_F_DUAL_RESIDUAL, LOIKB_F_PRIMAL_RESIDUAL_TASK, LOIKB_F_PRIMAL_RESIDUAL_SLACK,
  LOIKB_F_DUAL_RESIDUAL_V, LOIKB_F_DUAL_RESIDUAL_NU, LOIKB_F_TOL_PRIMAL, LOIKB_F_TOL_DUAL, LOIKB_F_MU, LOIKB_F_MU_EQ,
  LOIKB_F_MU_INEQ, LOIKB_F_DELTA_X_QP_INF_NORM, LOIKB_F_DELTA_Z_QP_INF_NORM, LOIKB_F_DELTA_Y_QP_INF_NORM,
  LOIKB_F_A_QP_T_DELTA_Y_QP_INF_NORM, LOIKB_F_UB_QP_T_DELTA_Y_QP_PLUS, LOIKB_F_LB_QP_T_DELTA_Y_QP_MINUS,
  LOIKB_F_DELTA_FIS_INF_NORM, LOIKB_F_DELTA_YIS_INF_NORM, LOIKB_F_DELTA_W_INF_NORM, LOIKB_F_DELTA_VIS_INF_NORM,
  LOIKB_F_DELTA_NU_INF_NORM, LOIKB_F_AV_INF_NORM, LOIKB_F_NU_INF_NORM, LOIKB_F_HREF_V_INF_NORM, LOIKB_F_G_INF_NORM,
  LOIKB_F_STF_PLUS_W_INF_NORM, LOIKB_F_PRIMAL_INFEASIBILITY_COND_1, LOIKB_F_PRIMAL_INFEASIBILITY_COND_2,
  LOIKB_F_TAIL_SOLVE_ITER,
  /* double [batch][nq]: the configurations resident on the device (SolveInit/Solve input, advanced by
     loikb_integrate) */
  LOIKB_F_Q = 96,
  /* int [batch]: how many times UpdateMu (optimized.hxx:613-641) changed mu in the last solve -- a diagnostic: the
     instances that run to max_iter are the ones that keep flipping mu between two decades */
  LOIKB_F_MU_UPDATES = 97,
  /* double [batch][6 nb + nv]: get_primal_residual_vec() / get_dual_residual_vec() (loik-loid-optimized.hpp:698-699)
     of the last iteration.  The hot path only forms their running maxima (the scalar residuals); these getters
     rebuild the vectors from the state: primal = (A_c v_c - b_c on the constrained links' rows, 0 elsewhere | nu - z),
     dual = (H_ref v_i - H_ref v_ref + fis_diff_plus_Aty[i] | Stf_plus_w) */
  LOIKB_F_PRIMAL_RESIDUAL_VEC = 98,
  LOIKB_F_DUAL_RESIDUAL_VEC = 99
};
/* copies one field for the whole batch into `out` (host pointer, or device pointer with LOIKB_OUT_DEVICE) */
int loikb_get(loikb_solver *s, int field, void *out, int out_flags);
/* The members of the reference's caller-owned data object a solve leaves behind (IkIdDataTypeOptimizedTpl, loik-loid-data-optimized.hpp:
 * nu :118, vis :124, fis :154, yis :162, w :170, z :178) in ONE call, for the whole batch, into host arrays laid out as loikb_get lays
 * them out ([batch][nv] for z / nu / w, [batch][njoints - 1][6] for vis / fis, [batch][active constraints][6] for yis).  `mask` selects
 * (LOIKB_RES_*); a selected member's pointer must not be NULL, the others are ignored.  One gather launch and one synchronisation when
 * the selected members of the batch fit 4 MiB (LOIKB_RESULTS_FUSED_BYTES) -- one problem per call, the reference's own use
 * (tests/loik-loid.cpp:987-1032), pays 0.02 ms for its results instead of 0.33 ms through six loikb_get calls --, field by field above that.
 * Same values as loikb_get's, bit for bit. */
#define LOIKB_RES_Z 1u
#define LOIKB_RES_NU 2u
#define LOIKB_RES_W 4u
#define LOIKB_RES_VIS 8u
#define LOIKB_RES_FIS 16u
#define LOIKB_RES_YIS 32u
#define LOIKB_RES_ALL 63u
/* ... and what a caller asks a solver after every solve -- get_iter(), get_convergence_status(), get_primal_infeasibility_status(), the
 * residuals and the other scalar getters (task-solver-base.hpp:87-141, loik-loid-optimized.hpp:698-755) -- in the same gather:
 * `scalars` [batch][LOIKB_RES_NSCALARS] doubles: entries 0 .. 29 the fields LOIKB_F_PRIMAL_RESIDUAL .. LOIKB_F_TAIL_SOLVE_ITER in the
 * enum's order, 30 the iteration count, 31 the status bits (LOIKB_F_STATUS: 1 converged, 2 primal infeasible, 4 tail solve ran,
 * 8 finished), 32 the number of mu updates.  (Each scalar through loikb_get costs a small batch 0.025 ms: three of them cost one
 * problem half its solve.) */
#define LOIKB_RES_SCALARS 64u
#define LOIKB_RES_NSCALARS 33
#define LOIKB_RES_SCALAR_ITER 30
#define LOIKB_RES_SCALAR_STATUS 31
#define LOIKB_RES_SCALAR_MU_UPDATES 32
#define LOIKB_RESULTS_FUSED_BYTES_DEFAULT (4u << 20)
int loikb_get_results(loikb_solver *s, unsigned int mask, double *z, double *nu, double *w, double *vis, double *fis, double *yis,
                      double *scalars);

/* measurement: what the last loikb_solve* call did */
typedef struct loikb_stats {
  unsigned long long instance_iterations; /* ADMM iterations summed over instances                */
  int launches;                           /* kernel launches (solve kernel + tail kernel)         */
  int n_unfinished;                       /* instances that hit max_iter without stopping         */
  int compactions;                        /* lane compactions performed                           */
  int tail_instances;                     /* instances finished by the cooperative tail kernel    */
  double tail_ms;                         /* HIP-event time of the tail kernel (part of kernel_ms)*/
  double kernel_ms;                       /* HIP-event time of all solve + tail launches          */
  double total_ms;                        /* HIP-event time of the whole call on the stream       */
  double bytes_per_instance_iteration;    /* algorithmic bytes, SURVEY.md 8(d): s*(203 nb+108 nc) */
  unsigned long long tail_instance_iterations; /* the part of instance_iterations run by the tail kernel */
  int tail_launches;                      /* launches of the tail kernel (part of `launches`)     */
  int team;                               /* wavefronts per tile used by the solve kernel         */
  int chunks;                             /* independent ranges of the batch solved concurrently (own stream each);
                                             kernel_ms / tail_ms sum the launches of all chunks, so with chunks > 1
                                             they can exceed total_ms                                            */
  double solve_busy_ms;                   /* time during which >= 1 solve-kernel launch was executing (HIP events;
                                             equals kernel_ms - tail_ms when chunks == 1)                        */
  double tail_busy_ms;                    /* same for the tail kernel                                            */
  int lean_launches;                      /* tail launches that ran the lean kernel (two wavefronts per SIMD, decade
                                             slots of H precomputed); the rest ran the one-wavefront-per-SIMD kernel   */
  int lean_escaped;                       /* instances whose mu left the precomputed decades in a lean launch and were
                                             finished by the other tail kernel                                         */
  double hslots_ms;                       /* HIP-event time of the decade-slot precomputation (part of tail_ms); 0 in
                                             the short sequence of a small batch unless LOIKB_SMALL_SLOT_EVENT=1       */
  int lean_requeues;                      /* time slices that ended with the instance going to the back of the lean kernel's
                                             work queue (round-robin among the instances waiting for a slot)                */
  int flat_launches;                      /* of lean_launches: those that ran the flat engine (k_fslots + k_flat: no loops over the
                                             tree levels, loik_amd/csrc/loik_flat.hpp) instead of k_hslots + k_lean             */
  double queue_dry_ms;                    /* flat engine: time from the start of its (last) launch until a lane group first found the
                                             work queue empty -- the bulk phase; the rest of the launch waits for its long runners */
  int flat_split_launches;                /* of flat_launches: those in the build with two lanes per joint (k_flat2,
                                             loik_amd/csrc/loik_flat2.hpp: robots of 17..64 joints, fp64)                        */
  int flat_ordered;                       /* of flat_launches: those that took their instances longest first, in the order the
                                             handle's previous solve left (LOIKB_FLAT_ORDER=0 turns that off)                   */
  int flat_built;                         /* decade slots (W / Dinv of one instance for one mu) built by the instance's own wavefront
                                             inside k_flat2: every change of mu under the OSQP rule, decades outside the table
                                             with LOIKB_FLAT_BUILD=1 (round 5)                                                  */
  int flat_probe_launches;                /* of flat_split_launches: those that ran as TWO launches -- every instance for
                                             LOIKB_FLAT_PROBE=p iterations at most, then the survivors to completion, longest
                                             predicted first (round 6; an experiment kept as an option: the default is one launch
                                             with round-robin time slices, which it does not beat)                              */
  double probe_ms;                        /* HIP-event time of those probe launches incl. the sort of the survivors (part of tail_ms) */
} loikb_stats;
int loikb_get_stats(loikb_solver *s, loikb_stats *out);
/* which kernels the solves of this handle use and why (the engine plan is made in one place, from (nb, nc, sharing mode of
 * A, children per joint, batch, options): at create and again at SolveInit) -- a human-readable line, valid until the next
 * call on this thread */
const char *loikb_plan_string(loikb_solver *s);
/* The flat engine's schedule of a kinematic tree (inspection / tests; loik_amd/csrc/loik_flat.hpp: the engine replaces the
 * level-by-level recursions of LoikBackwardStepVisitor / LoikForwardStep2Visitor, loik-loid-optimized.hxx:31-81, :102-163, by
 * sums over subtrees and root paths).  meta[5] = {applicable, lanes per instance G, ancestors per joint, scan steps, jump
 * rounds}; out = G records of 40 ints: depth, subtree size, jmp[5] (lane of the ancestor at distance 1, 2, 4, 8, 16), anc[16]
 * (lane of the ancestor at depth k + 1), red[8] (entries k * G + lane of the W tau products the lane sums), helper, part[8]
 * (lanes whose partial sums the lane collects); -1 = none.  Returns 0, or the number of ints `out` needs when cap is smaller;
 * not applicable (meta[0] == 0): loikb_last_error() says why. */
int loikb_flat_schedule(const int *parents, int njoints, int *out, int cap, int *meta);

/* introspection */
int loikb_batch(const loikb_solver *s);
int loikb_nv(const loikb_solver *s);
int loikb_njoints(const loikb_solver *s);
const char *loikb_last_error(void);
const char *loikb_status_string(int code);
int loikb_version(void);
/* number of visible HIP devices (0 when none / no driver) */
int loikb_device_count(void);

/* Host-side introspection of the sweep schedule (no device needed).  The solve kernel advances a tile of 64 instances
 * with a team of `team` wavefronts; the joints of a tree sweep are list-scheduled onto the wavefronts, one joint per
 * wavefront and step.  direction 0 = leaf->root (FwdPass1+BwdPass / BwdPass2 order), 1 = root->leaf (FwdPass2 order).
 * joint_out / flags_out: [team][steps_cap] row-major, joint 0 = idle step; flags: 1 = child contribution arrives in
 * registers, 2 = result stays in registers for the parent, 4 = result goes through an LDS slot (slot_out),
 * 8 = parent velocity still in registers.  Returns the number of steps (the sweep's critical path in joint visits),
 * or a negative status (LOIKB_ERR_ARG when steps_cap is too small or the tree is malformed).                          */
int loikb_sweep_schedule(const int *parents, int njoints, int team, int direction, int steps_cap, int *joint_out,
                         int *flags_out, int *slot_out, int *lds_slots_out);

#ifdef __cplusplus
}
#endif
#endif
