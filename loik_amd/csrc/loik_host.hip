// loik_host.hip -- host driver + C-ABI (include/loik_amd.h) of the batched LoIK solver for MI355X.
//
// Host-side mirror of the reference's orchestration (paths under /root/reference/):
//   loikb_create          <- IkIdDataTypeOptimizedTpl ctor + FirstOrderLoikOptimizedTpl ctor
//                            (include/loik/loik-loid-data-optimized.hxx:40-104, loik-loid-optimized.hpp:129-162)
//   loikb_solve_init      <- SolveInit          (loik-loid-optimized.hpp:335-361)
//   loikb_solve           <- Solve()            (loik-loid-optimized.hpp:368-455)
//   loikb_solve_full      <- Solve(q,H_ref,...) (loik-loid-optimized.hpp:475-580)
//   loikb_solve_tailored  <- Solve(q,c_id,A,b)  (loik-loid-optimized.hpp:596-695)
// Reset semantics follow IkIdDataTypeOptimizedTpl::Reset / ResetRecursion
// (loik-loid-data-optimized.hxx:114-154) and IkProblemFormulationOptimized (ik-id-description-optimized.hpp).
#include "loik_device.hpp"

#include "../../include/loik_amd.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace loikb;

static thread_local std::string g_last_error;

#define HIPCHK(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) {                                                                                \
      char _buf[512];                                                                                      \
      snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      g_last_error = _buf;                                                                                 \
      return LOIKB_ERR_HIP;                                                                                \
    }                                                                                                      \
  } while (0)

namespace {

struct DevMem {
  void* p = nullptr;
  size_t bytes = 0;
};

// type-erased device workspace; element size chosen at create time
struct loikb_solver_impl {
  // model (copied)
  int nj = 0, nb = 0, nq = 0, nv = 0;
  std::vector<int> parents, jtype, idx_q, idx_v;
  std::vector<JointDesc> jd;
  int stack_levels = 0;
  // options
  loikb_options opt{};
  int B = 0, ld = 0, nc = 0;
  size_t esz = 8;
  bool f32 = false;
  // problem (uniform part)
  double Href[36]{}, vref[6]{}, Hv[6]{};
  double Hv_inf_norm = 0.0;
  std::vector<int> active_ids;
  bool have_problem = false;
  bool a_shared = true, bnd_shared = true;
  // device
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  std::vector<DevMem> allocs;
  JointDesc* d_jd = nullptr;
  int* d_idx_q = nullptr;
  unsigned int* d_counters = nullptr;
  unsigned int* h_counters = nullptr;  // pinned
  void* d_stage = nullptr;             // staging for host<->device transposes (doubles)
  size_t stage_bytes = 0;
  // per-instance SoA arrays live in buffer sets: set 0 = home (slot == instance id, capacity ld);
  // sets 1,2 = compaction work sets (capacity ld/2), allocated on first use
  struct Set {
    int ld = 0;
    bool allocated = false;
    void *cs = nullptr, *v = nullptr, *f = nullptr, *g = nullptr, *nu = nullptr, *z = nullptr, *w = nullptr,
         *s = nullptr, *y = nullptr, *aty = nullptr, *H = nullptr, *p = nullptr, *ud = nullptr, *dinv = nullptr,
         *rr = nullptr, *A = nullptr, *AtA = nullptr, *b = nullptr, *Atb = nullptr, *lb = nullptr, *ub = nullptr,
         *bnorm = nullptr, *mu = nullptr, *mu_h = nullptr, *scal = nullptr;
    int *iter = nullptr, *status = nullptr, *map = nullptr, *wave_live = nullptr, *wave_off = nullptr;
  } set[3];
  std::vector<int> h_wave;  // host scratch for the compaction scan
  // stats of the last solve
  loikb_stats stats{};
};

int alloc_dev(loikb_solver_impl* S, void** out, size_t bytes)
{
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, bytes ? bytes : 16));
  HIPCHK(hipMemsetAsync(p, 0, bytes ? bytes : 16, S->stream));
  S->allocs.push_back({p, bytes});
  *out = p;
  return LOIKB_OK;
}

int ensure_stage(loikb_solver_impl* S, size_t bytes)
{
  if (bytes <= S->stage_bytes) return LOIKB_OK;
  if (S->d_stage) HIPCHK(hipFree(S->d_stage));
  S->d_stage = nullptr;
  S->stage_bytes = 0;
  HIPCHK(hipMalloc(&S->d_stage, bytes));
  S->stage_bytes = bytes;
  return LOIKB_OK;
}

// allocate every per-instance array of buffer set k with capacity `ld` slots
int alloc_set(loikb_solver_impl* S, int k, int ld_)
{
  loikb_solver_impl::Set& W = S->set[k];
  if (W.allocated) return LOIKB_OK;
  const size_t ld = ld_, e = S->esz, nb = S->nb, nc = S->nc > 0 ? S->nc : 1;
  W.ld = ld_;
  int rc;
#define A_(field, n) if ((rc = alloc_dev(S, &W.field, (size_t)(n) * ld * e))) return rc
  A_(cs, 2 * nb); A_(v, 6 * nb); A_(f, 6 * nb); A_(g, 6 * nb); A_(nu, nb); A_(z, nb); A_(w, nb); A_(s, nb);
  A_(y, 6 * nc); A_(aty, 6 * nc); A_(H, 21 * nb); A_(p, 6 * nb); A_(ud, 6 * nb); A_(dinv, nb); A_(rr, nb);
  A_(A, 36 * nc); A_(AtA, 21 * nc); A_(b, 6 * nc); A_(Atb, 6 * nc); A_(lb, nb); A_(ub, nb); A_(bnorm, 1);
  A_(mu, 1); A_(mu_h, 1); A_(scal, NSCAL);
#undef A_
  void* tmp = nullptr;
  if ((rc = alloc_dev(S, &tmp, sizeof(int) * ld))) return rc; W.iter = (int*)tmp;
  if ((rc = alloc_dev(S, &tmp, sizeof(int) * ld))) return rc; W.status = (int*)tmp;
  if ((rc = alloc_dev(S, &tmp, sizeof(int) * ld))) return rc; W.map = (int*)tmp;
  if ((rc = alloc_dev(S, &tmp, sizeof(int) * (ld / WAVE + 1)))) return rc; W.wave_live = (int*)tmp;
  if ((rc = alloc_dev(S, &tmp, sizeof(int) * (ld / WAVE + 1)))) return rc; W.wave_off = (int*)tmp;
  W.allocated = true;
  return LOIKB_OK;
}

inline dim3 grid1(size_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

// build the uniform per-joint schedule from the Pinocchio-style model
int build_schedule(loikb_solver_impl* S, const loikb_model_desc* m)
{
  const int nj = m->njoints;
  if (nj < 2 || m->nv != nj - 1 || m->nq != nj - 1) {
    g_last_error = "model not supported: only 1-DoF joints (nq == nv == njoints-1)";
    return LOIKB_ERR_MODEL;
  }
  S->nj = nj; S->nb = nj - 1; S->nq = m->nq; S->nv = m->nv;
  S->parents.assign(m->parents, m->parents + nj);
  S->jtype.assign(m->jtype, m->jtype + nj);
  S->idx_q.assign(m->idx_q, m->idx_q + nj);
  S->idx_v.assign(m->idx_v, m->idx_v + nj);
  std::vector<int> nchild(nj, 0), last_child(nj, -1), subtree_end(nj, 0);
  for (int i = 1; i < nj; ++i) {
    const int p = S->parents[i];
    if (p < 0 || p >= i) { g_last_error = "model: parents[i] must be < i"; return LOIKB_ERR_MODEL; }
    if (S->idx_v[i] != i - 1 || S->idx_q[i] != i - 1) {
      g_last_error = "model: idx_q/idx_v must equal joint index - 1 (all joints 1-DoF)";
      return LOIKB_ERR_MODEL;
    }
    if (S->jtype[i] < LOIKB_J_RX || S->jtype[i] > LOIKB_J_PU) {
      g_last_error = "model: unsupported joint type";
      return LOIKB_ERR_MODEL;
    }
    nchild[p]++;
    last_child[p] = i;  // increasing i: ends as the largest-index child
  }
  // depth-first numbering check: descendants of every joint are the contiguous range (i, subtree_end[i]]
  for (int i = nj - 1; i >= 0; --i) subtree_end[i] = i;
  for (int i = nj - 1; i >= 1; --i) {
    const int p = S->parents[i];
    if (subtree_end[i] > subtree_end[p]) subtree_end[p] = subtree_end[i];
  }
  for (int i = 1; i < nj; ++i) {
    // every joint in (i, subtree_end[i]] must have its parent inside [i, subtree_end[i]]
    for (int k = i + 1; k <= subtree_end[i]; ++k)
      if (S->parents[k] < i) { g_last_error = "model: joints are not numbered depth-first"; return LOIKB_ERR_MODEL; }
  }
  S->jd.assign(nj, JointDesc{});
  for (int i = 1; i < nj; ++i) {
    JointDesc& d = S->jd[i];
    for (int k = 0; k < 9; ++k) d.Rp[k] = m->placement[12 * i + k];
    for (int k = 0; k < 3; ++k) d.tp[k] = m->placement[12 * i + 9 + k];
    const int jt = S->jtype[i];
    double ax[3] = {0, 0, 0};
    int rot = ROT_NONE, flags = 0;
    switch (jt) {
    case LOIKB_J_RX: ax[0] = 1; rot = ROT_X; flags |= JF_REVOLUTE; break;
    case LOIKB_J_RY: ax[1] = 1; rot = ROT_Y; flags |= JF_REVOLUTE; break;
    case LOIKB_J_RZ: ax[2] = 1; rot = ROT_Z; flags |= JF_REVOLUTE; break;
    case LOIKB_J_PX: ax[0] = 1; break;
    case LOIKB_J_PY: ax[1] = 1; break;
    case LOIKB_J_PZ: ax[2] = 1; break;
    case LOIKB_J_RU: for (int k = 0; k < 3; ++k) ax[k] = m->axis[3 * i + k]; rot = ROT_U; flags |= JF_REVOLUTE; break;
    case LOIKB_J_PU: for (int k = 0; k < 3; ++k) ax[k] = m->axis[3 * i + k]; break;
    }
    for (int k = 0; k < 3; ++k) d.axis[k] = ax[k];
    d.parent = S->parents[i];
    if (nchild[i] == 0) flags |= JF_LEAF;
    if (d.parent == 0) flags |= JF_PARENT_ROOT;
    if (last_child[d.parent] == i) flags |= JF_LAST_CHILD;
    if (d.parent == i - 1) flags |= JF_NEXT_IS_PARENT;
    d.flags = flags;
    d.cslot = -1;
    d.rot = rot;
  }
  // LDS stack depth needed by the leaf->root sweeps
  int level = 0, maxlevel = 0;
  for (int i = nj - 1; i >= 1; --i) {
    const JointDesc& d = S->jd[i];
    if (d.flags & JF_PARENT_ROOT) continue;
    if (!(d.flags & JF_LAST_CHILD)) --level;
    if (!(d.flags & JF_NEXT_IS_PARENT)) { ++level; if (level > maxlevel) maxlevel = level; }
  }
  S->stack_levels = maxlevel;
  return LOIKB_OK;
}

template <typename T>
Bufs<T> make_bufs(loikb_solver_impl* S, int k = 0)
{
  const loikb_solver_impl::Set& W = S->set[k];
  const loikb_solver_impl::Set& H0 = S->set[0];
  Bufs<T> Bf{};
  Bf.cs = (const T*)W.cs; Bf.v = (T*)W.v; Bf.f = (T*)W.f; Bf.g = (T*)W.g; Bf.nu = (T*)W.nu; Bf.z = (T*)W.z;
  Bf.w = (T*)W.w; Bf.s = (T*)W.s; Bf.y = (T*)W.y; Bf.aty = (T*)W.aty; Bf.H = (T*)W.H; Bf.p = (T*)W.p;
  Bf.ud = (T*)W.ud; Bf.dinv = (T*)W.dinv; Bf.rr = (T*)W.rr;
  // shared inputs exist once (in the home set); per-instance inputs travel with the instance
  Bf.A = (const T*)(S->a_shared ? H0.A : W.A); Bf.AtA = (const T*)(S->a_shared ? H0.AtA : W.AtA);
  Bf.b = (const T*)W.b; Bf.Atb = (const T*)W.Atb;
  Bf.lb = (const T*)(S->bnd_shared ? H0.lb : W.lb); Bf.ub = (const T*)(S->bnd_shared ? H0.ub : W.ub);
  Bf.bnorm = (const T*)W.bnorm; Bf.mu = (T*)W.mu; Bf.mu_h = (T*)W.mu_h; Bf.iter = W.iter; Bf.status = W.status;
  Bf.scal = (T*)W.scal; Bf.counters = S->d_counters; Bf.wave_live = W.wave_live;
  return Bf;
}

template <typename T>
Params<T> make_params(loikb_solver_impl* S)
{
  Params<T> P{};
  for (int k = 0; k < 36; ++k) P.Href[k] = (T)S->Href[k];
  for (int k = 0; k < 6; ++k) P.Hv[k] = (T)S->Hv[k];
  P.Hv_inf_norm = (T)S->Hv_inf_norm;
  P.rho = (T)S->opt.rho; P.mu0 = (T)S->opt.mu; P.mu_scale = (T)S->opt.mu_equality_scale_factor;
  P.tol_abs = (T)S->opt.tol_abs; P.tol_rel = (T)S->opt.tol_rel; P.tol_primal_inf = (T)S->opt.tol_primal_inf;
  P.tol_tail_solve = (T)S->opt.tol_tail_solve;
  P.max_iter = S->opt.max_iter;
  int mode = 0;
  if (S->opt.flags & LOIKB_OPT_FIXED_ITERS) mode |= MODE_FIXED_ITERS;
  if (!(S->opt.flags & LOIKB_OPT_NO_H_CACHE)) mode |= MODE_CACHE_H;
  if (S->a_shared) mode |= MODE_A_SHARED;
  if (S->bnd_shared) mode |= MODE_BND_SHARED;
  P.mode = mode;
  P.nb = S->nb; P.nc = S->nc; P.B = S->B; P.ld = S->ld;
  P.max_launch_iters = S->opt.max_launch_iters > 0 ? S->opt.max_launch_iters : (S->opt.max_iter + 1);
  P.stack_levels = S->stack_levels;
  return P;
}

// memset a whole SoA field
int zero_field(loikb_solver_impl* S, void* p, size_t rows)
{
  HIPCHK(hipMemsetAsync(p, 0, rows * (size_t)S->ld * S->esz, S->stream));
  return LOIKB_OK;
}

template <typename T>
int fill_field(loikb_solver_impl* S, void* p, size_t n, double val)
{
  hipLaunchKernelGGL(k_fill<T>, grid1(n), dim3(256), 0, S->stream, (T*)p, n, (T)val);
  HIPCHK(hipGetLastError());
  return LOIKB_OK;
}

// IkIdDataTypeOptimizedTpl::Reset(warm_start), loik-loid-data-optimized.hxx:114-127
int data_reset(loikb_solver_impl* S, bool warm_start)
{
  if (warm_start) return LOIKB_OK;
  int rc;
  if ((rc = zero_field(S, S->set[0].w, S->nb))) return rc;
  if ((rc = zero_field(S, S->set[0].z, S->nb))) return rc;
  if ((rc = zero_field(S, S->set[0].nu, S->nb))) return rc;
  if ((rc = zero_field(S, S->set[0].v, 6 * (size_t)S->nb))) return rc;
  if ((rc = zero_field(S, S->set[0].f, 6 * (size_t)S->nb))) return rc;
  if ((rc = zero_field(S, S->set[0].g, 6 * (size_t)S->nb))) return rc;
  return LOIKB_OK;
}

// ResetRecursion(), loik-loid-data-optimized.hxx:138-154 (nu and Stf_plus_w are NOT reset upstream)
int data_reset_recursion(loikb_solver_impl* S)
{
  int rc;
  if ((rc = zero_field(S, S->set[0].w, S->nb))) return rc;
  if ((rc = zero_field(S, S->set[0].z, S->nb))) return rc;
  if ((rc = zero_field(S, S->set[0].v, 6 * (size_t)S->nb))) return rc;
  if ((rc = zero_field(S, S->set[0].f, 6 * (size_t)S->nb))) return rc;
  if ((rc = zero_field(S, S->set[0].g, 6 * (size_t)S->nb))) return rc;
  if ((rc = zero_field(S, S->set[0].y, 6 * (size_t)S->nc))) return rc;
  if ((rc = zero_field(S, S->set[0].aty, 6 * (size_t)S->nc))) return rc;
  return LOIKB_OK;
}

// ResetSolver(), loik-loid-optimized.hpp:168-186 + Base::Reset task-solver-base.hpp:73-84
int reset_solver(loikb_solver_impl* S)
{
  int rc;
  HIPCHK(hipMemsetAsync(S->set[0].iter, 0, sizeof(int) * (size_t)S->ld, S->stream));
  HIPCHK(hipMemsetAsync(S->set[0].status, 0, sizeof(int) * (size_t)S->ld, S->stream));
  if ((rc = zero_field(S, S->set[0].scal, NSCAL))) return rc;
  if (S->f32) rc = fill_field<float>(S, S->set[0].mu, S->ld, S->opt.mu);
  else rc = fill_field<double>(S, S->set[0].mu, S->ld, S->opt.mu);
  return rc;
}

int invalidate_h_cache(loikb_solver_impl* S)
{
  if (S->f32) return fill_field<float>(S, S->set[0].mu_h, S->ld, -1.0);
  return fill_field<double>(S, S->set[0].mu_h, S->ld, -1.0);
}

// bring a per-instance instance-major double array [B][n] (host or device) into SoA [n][ld] of T
int upload_aos(loikb_solver_impl* S, const double* src, int n, void* dst, bool src_device, bool shared)
{
  const size_t count = shared ? (size_t)n : (size_t)S->B * n;
  const double* dsrc = src;
  if (!src_device) {
    int rc = ensure_stage(S, count * sizeof(double));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(S->d_stage, src, count * sizeof(double), hipMemcpyHostToDevice, S->stream));
    dsrc = (const double*)S->d_stage;
  }
  if (shared) {
    // [n] -> [n] (T); a 1-"instance" transpose with ld = 1
    if (S->f32) hipLaunchKernelGGL(k_aos_to_soa<float>, dim3(1), dim3(64), 0, S->stream, dsrc, n, 1, 1, (float*)dst);
    else hipLaunchKernelGGL(k_aos_to_soa<double>, dim3(1), dim3(64), 0, S->stream, dsrc, n, 1, 1, (double*)dst);
  } else {
    if (S->f32)
      hipLaunchKernelGGL(k_aos_to_soa<float>, grid1(S->B), dim3(256), 0, S->stream, dsrc, n, S->B, S->ld, (float*)dst);
    else
      hipLaunchKernelGGL(k_aos_to_soa<double>, grid1(S->B), dim3(256), 0, S->stream, dsrc, n, S->B, S->ld, (double*)dst);
  }
  HIPCHK(hipGetLastError());
  if (!src_device) HIPCHK(hipStreamSynchronize(S->stream));  // staging buffer is reused
  return LOIKB_OK;
}

// a per-instance array given once ([n], host) and replicated to every instance: SoA rows filled with a constant
int upload_broadcast(loikb_solver_impl* S, const double* src, int n, void* dst)
{
  for (int k = 0; k < n; ++k) {
    int rc;
    if (S->f32) rc = fill_field<float>(S, (float*)dst + (size_t)k * S->ld, S->ld, src[k]);
    else rc = fill_field<double>(S, (double*)dst + (size_t)k * S->ld, S->ld, src[k]);
    if (rc) return rc;
  }
  return LOIKB_OK;
}

// FwdPassInit(q), loik-loid-optimized.hxx:253-283
int fwd_pass_init(loikb_solver_impl* S, const double* q, int in_flags)
{
  const bool dev = in_flags & LOIKB_IN_DEVICE;
  const double* dq = q;
  std::vector<double> rep;
  if (in_flags & LOIKB_Q_SHARED) {
    // replicate on the host (single-instance convenience path)
    rep.resize((size_t)S->B * S->nq);
    std::vector<double> hq(S->nq);
    if (dev) {
      HIPCHK(hipMemcpyAsync(hq.data(), q, sizeof(double) * S->nq, hipMemcpyDeviceToHost, S->stream));
      HIPCHK(hipStreamSynchronize(S->stream));
    } else {
      memcpy(hq.data(), q, sizeof(double) * S->nq);
    }
    for (int b = 0; b < S->B; ++b) memcpy(&rep[(size_t)b * S->nq], hq.data(), sizeof(double) * S->nq);
    q = rep.data();
  }
  if (!dev || (in_flags & LOIKB_Q_SHARED)) {
    const size_t bytes = (size_t)S->B * S->nq * sizeof(double);
    int rc = ensure_stage(S, bytes);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(S->d_stage, q, bytes, hipMemcpyHostToDevice, S->stream));
    dq = (const double*)S->d_stage;
  }
  if (S->f32)
    hipLaunchKernelGGL(k_fk_init<float>, grid1(S->B), dim3(256), 0, S->stream, dq, S->nq, S->d_jd, S->d_idx_q, S->nb,
                       S->B, S->ld, (float*)S->set[0].cs);
  else
    hipLaunchKernelGGL(k_fk_init<double>, grid1(S->B), dim3(256), 0, S->stream, dq, S->nq, S->d_jd, S->d_idx_q, S->nb,
                       S->B, S->ld, (double*)S->set[0].cs);
  HIPCHK(hipGetLastError());
  if (dq == S->d_stage) HIPCHK(hipStreamSynchronize(S->stream));
  // H/UDinv/Dinv cache depends on liMi
  int rc = invalidate_h_cache(S);
  if (rc) return rc;
  // cold start: yis = 0, Aty = 0 (hxx:270-278)
  if (!S->opt.warm_start) {
    if ((rc = zero_field(S, S->set[0].y, 6 * (size_t)S->nc))) return rc;
    if ((rc = zero_field(S, S->set[0].aty, 6 * (size_t)S->nc))) return rc;
  }
  return LOIKB_OK;
}

int upload_jd(loikb_solver_impl* S)
{
  HIPCHK(hipMemcpyAsync(S->d_jd, S->jd.data(), sizeof(JointDesc) * S->nj, hipMemcpyHostToDevice, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  return LOIKB_OK;
}

int constraint_products(loikb_solver_impl* S, int c_lo, int c_hi, bool grow_only)
{
  if (S->f32)
    hipLaunchKernelGGL(k_constraint_products<float>, grid1(S->B), dim3(256), 0, S->stream, (const float*)S->set[0].A,
                       (const float*)S->set[0].b, S->nc, c_lo, c_hi, (int)S->a_shared, S->B, S->ld, (float*)S->set[0].AtA,
                       (float*)S->set[0].Atb, (float*)S->set[0].bnorm, (int)grow_only);
  else
    hipLaunchKernelGGL(k_constraint_products<double>, grid1(S->B), dim3(256), 0, S->stream, (const double*)S->set[0].A,
                       (const double*)S->set[0].b, S->nc, c_lo, c_hi, (int)S->a_shared, S->B, S->ld, (double*)S->set[0].AtA,
                       (double*)S->set[0].Atb, (double*)S->set[0].bnorm, (int)grow_only);
  HIPCHK(hipGetLastError());
  return LOIKB_OK;
}

// shared A: AtA computed once on the host (ik-id-description-optimized.hpp:162)
int upload_shared_AtA(loikb_solver_impl* S, const double* A, int c)
{
  double AtA[21];
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      double a = 0.0;
      for (int k = 0; k < 6; ++k) a += A[6 * k + i] * A[6 * k + j];
      AtA[sym(i, j)] = a;
    }
  if (S->f32) {
    float tmp[21];
    for (int k = 0; k < 21; ++k) tmp[k] = (float)AtA[k];
    HIPCHK(hipMemcpyAsync((float*)S->set[0].AtA + 21 * c, tmp, sizeof(tmp), hipMemcpyHostToDevice, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
  } else {
    HIPCHK(hipMemcpyAsync((double*)S->set[0].AtA + 21 * c, AtA, sizeof(AtA), hipMemcpyHostToDevice, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
  }
  return LOIKB_OK;
}

// problem_.UpdateReference / UpdateIneqConstraints / UpdateEqConstraints (ik-id-description-optimized.hpp:78-171,
// :325-339) with the batch layouts of loik_amd.h
int set_problem(loikb_solver_impl* S, const double* H_ref, const double* v_ref, const int* c_ids, int nc,
                const double* Ais, const double* bis, const double* lb, const double* ub, int nbound, int in_flags)
{
  if (nbound != S->nv) { g_last_error = "lb/ub dimension differs from model.nv"; return LOIKB_ERR_INEQ_DIM; }
  if (nc != S->nc) { g_last_error = "number of equality constraints doesn't match initialization"; return LOIKB_ERR_EQ_C_SIZE; }
  for (int i = 0; i < 6; ++i)
    for (int j = i + 1; j < 6; ++j)
      if (std::fabs(H_ref[6 * i + j] - H_ref[6 * j + i]) > 1e-14 * (1.0 + std::fabs(H_ref[6 * i + j]))) {
        g_last_error = "H_ref must be symmetric";
        return LOIKB_ERR_HREF_NOT_SYMMETRIC;
      }
  for (int c = 0; c < nc; ++c) {
    if (c_ids[c] < 1 || c_ids[c] >= S->nj) { g_last_error = "constraint link id out of range"; return LOIKB_ERR_ARG; }
    for (int c2 = 0; c2 < c; ++c2)
      if (c_ids[c2] == c_ids[c]) { g_last_error = "multiple constraints on the same link"; return LOIKB_ERR_DUP_CONSTRAINT; }
  }
  const bool dev = in_flags & LOIKB_IN_DEVICE;
  // UpdateReference: Hv = H_ref v_ref, Hv_inf_norm_ (hpp:85-96)
  memcpy(S->Href, H_ref, sizeof(S->Href));
  memcpy(S->vref, v_ref, sizeof(S->vref));
  S->Hv_inf_norm = 0.0;
  for (int i = 0; i < 6; ++i) {
    double a = 0.0;
    for (int k = 0; k < 6; ++k) a += H_ref[6 * i + k] * v_ref[k];
    S->Hv[i] = a;
    if (std::fabs(a) > S->Hv_inf_norm) S->Hv_inf_norm = std::fabs(a);
  }
  // UpdateIneqConstraints
  S->bnd_shared = in_flags & LOIKB_BOUNDS_SHARED;
  int rc;
  if ((rc = upload_aos(S, lb, S->nv, S->set[0].lb, dev && !S->bnd_shared, S->bnd_shared))) return rc;
  if ((rc = upload_aos(S, ub, S->nv, S->set[0].ub, dev && !S->bnd_shared, S->bnd_shared))) return rc;
  // UpdateEqConstraints
  S->active_ids.assign(c_ids, c_ids + nc);
  for (int i = 1; i < S->nj; ++i) S->jd[i].cslot = -1;
  for (int c = 0; c < nc; ++c) S->jd[c_ids[c]].cslot = c;
  if ((rc = upload_jd(S))) return rc;
  S->a_shared = in_flags & LOIKB_A_SHARED;
  if ((rc = upload_aos(S, Ais, 36 * nc, S->set[0].A, dev && !S->a_shared, S->a_shared))) return rc;
  if (S->a_shared)
    for (int c = 0; c < nc; ++c)
      if ((rc = upload_shared_AtA(S, Ais + 36 * c, c))) return rc;
  if (in_flags & LOIKB_B_SHARED) rc = upload_broadcast(S, bis, 6 * nc, S->set[0].b);
  else rc = upload_aos(S, bis, 6 * nc, S->set[0].b, dev, false);
  if (rc) return rc;
  if ((rc = constraint_products(S, 0, nc, false))) return rc;
  S->have_problem = true;
  return LOIKB_OK;
}

// list of the per-instance arrays that travel with an instance when it changes buffer set.  Inter-sweep
// temporaries (H, p, UDinv, Dinv, r) stay behind: they are rebuilt by the first sweep after the move (mu_h = -1).
template <typename T>
void fill_move_plan(loikb_solver_impl* S, int src, int dst, MovePlan& M)
{
  const loikb_solver_impl::Set &A = S->set[src], &D = S->set[dst], &H0 = S->set[0];
  const int nb = S->nb, nc = S->nc, e = (int)sizeof(T);
  int k = 0;
  auto add = [&](const void* s, void* dl, void* dh, int rows, int esz) {
    M.f[k].src = s; M.f[k].dst_live = dl; M.f[k].dst_home = src == 0 ? nullptr : dh; M.f[k].rows = rows; M.f[k].esz = esz;
    ++k;
  };
  add(A.cs, D.cs, H0.cs, 2 * nb, e); add(A.v, D.v, H0.v, 6 * nb, e); add(A.f, D.f, H0.f, 6 * nb, e);
  add(A.g, D.g, H0.g, 6 * nb, e); add(A.nu, D.nu, H0.nu, nb, e); add(A.z, D.z, H0.z, nb, e);
  add(A.w, D.w, H0.w, nb, e); add(A.s, D.s, H0.s, nb, e); add(A.y, D.y, H0.y, 6 * nc, e);
  add(A.aty, D.aty, H0.aty, 6 * nc, e); add(A.b, D.b, H0.b, 6 * nc, e); add(A.Atb, D.Atb, H0.Atb, 6 * nc, e);
  add(A.bnorm, D.bnorm, H0.bnorm, 1, e); add(A.mu, D.mu, H0.mu, 1, e); add(A.scal, D.scal, H0.scal, NSCAL, e);
  add(A.iter, D.iter, H0.iter, 1, 4); add(A.status, D.status, H0.status, 1, 4);
  if (!S->a_shared) { add(A.A, D.A, H0.A, 36 * nc, e); add(A.AtA, D.AtA, H0.AtA, 21 * nc, e); }
  if (!S->bnd_shared) { add(A.lb, D.lb, H0.lb, nb, e); add(A.ub, D.ub, H0.ub, nb, e); }
  M.nfields = k;
  M.ld_src = A.ld; M.ld_dst = D.ld; M.ld_home = H0.ld;
  M.status = A.status;
  M.map_src = src == 0 ? nullptr : A.map;
  M.map_dst = D.map;
  M.wave_off = A.wave_off;
  M.force_home = 0;
}

// move the live instances of set `src` (n_src slots) to the first slots of set `dst`; finished ones go home
template <typename T>
int compact(loikb_solver_impl* S, int src, int dst, int n_src, int* n_dst_out)
{
  loikb_solver_impl::Set& A = S->set[src];
  const int nw = (n_src + WAVE - 1) / WAVE;
  int rc;
  if (dst >= 0 && (rc = alloc_set(S, dst, ((S->ld / 2 + WAVE - 1) / WAVE) * WAVE))) return rc;
  // exclusive scan of the per-wavefront live counts on the host (nw <= B/64 ints)
  S->h_wave.resize(2 * (size_t)nw + 2);
  int* cnt = S->h_wave.data();
  int* off = cnt + nw + 1;
  HIPCHK(hipMemcpyAsync(cnt, A.wave_live, sizeof(int) * nw, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  int total = 0;
  for (int w = 0; w < nw; ++w) { off[w] = total; total += dst >= 0 ? cnt[w] : 0; }
  HIPCHK(hipMemcpyAsync(A.wave_off, off, sizeof(int) * nw, hipMemcpyHostToDevice, S->stream));
  MovePlan M{};
  fill_move_plan<T>(S, src, dst >= 0 ? dst : 0, M);
  M.n_src = n_src;
  M.force_home = dst < 0;
  hipLaunchKernelGGL(k_move, dim3(nw), dim3(WAVE), 0, S->stream, M);
  HIPCHK(hipGetLastError());
  if (dst >= 0) {
    // the H/UDinv/Dinv cache did not travel
    if ((rc = fill_field<T>(S, S->set[dst].mu_h, total, -1.0))) return rc;
  }
  HIPCHK(hipStreamSynchronize(S->stream));  // h_wave is reused
  if (n_dst_out) *n_dst_out = total;
  return LOIKB_OK;
}

template <typename T>
int run_main_loop_t(loikb_solver_impl* S)
{
  Params<T> P = make_params<T>(S);
  const size_t lds = (size_t)(S->stack_levels > 0 ? S->stack_levels : 1) * 27 * WAVE * sizeof(T);
  S->stats = loikb_stats{};
  S->stats.bytes_per_instance_iteration = (double)sizeof(T) * (203.0 * S->nb + 108.0 * S->nc);
  double kernel_ms = 0.0;
  HIPCHK(hipEventRecord(S->ev_t0, S->stream));
  // main-loop bound: at most max_iter-1 iterations, tail solve may reach max_iter (hpp:377, :276)
  const int max_total = S->opt.max_iter + 1;
  const bool can_compact = !(S->opt.flags & LOIKB_OPT_NO_COMPACTION) && !(S->opt.flags & LOIKB_OPT_FIXED_ITERS);
  // compaction pays only while the launch is bandwidth-bound (many wavefronts); below ~COMPACT_MIN_WAVES
  // wavefronts an ADMM iteration costs the same single-wavefront latency however few lanes are live
  const int compact_min = S->opt.compact_min_instances > 0 ? S->opt.compact_min_instances : 64 * WAVE;
  int cur = 0, n_cur = S->B;
  int done_iters = 0;
  unsigned long long inst_iters = 0;
  unsigned int n_live = 0;
  while (true) {
    const bool may_compact_later = can_compact && n_cur > compact_min;
    int launch_iters = S->opt.max_launch_iters > 0 ? S->opt.max_launch_iters : (may_compact_later ? 8 : max_total);
    if (launch_iters > max_total - done_iters) launch_iters = max_total - done_iters;
    P.B = n_cur;
    P.ld = S->set[cur].ld;
    P.max_launch_iters = launch_iters;
    Bufs<T> Bf = make_bufs<T>(S, cur);
    const dim3 grid((unsigned)((n_cur + WAVE - 1) / WAVE)), block(WAVE);
    HIPCHK(hipMemsetAsync(S->d_counters, 0, 2 * sizeof(unsigned int), S->stream));
    HIPCHK(hipEventRecord(S->ev_k0, S->stream));
    hipLaunchKernelGGL(k_solve<T>, grid, block, lds, S->stream, P, Bf, (const JointDesc*)S->d_jd);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(S->ev_k1, S->stream));
    HIPCHK(hipMemcpyAsync(S->h_counters, S->d_counters, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, S->ev_k0, S->ev_k1));
    kernel_ms += ms;
    S->stats.launches++;
    inst_iters += S->h_counters[1];
    n_live = S->h_counters[0];
    done_iters += launch_iters;
    if (n_live == 0 || done_iters >= max_total) break;
    if (may_compact_later && 2 * (long long)n_live <= n_cur) {
      const int dst = cur == 1 ? 2 : 1;
      int n_new = 0;
      int rc = compact<T>(S, cur, dst, n_cur, &n_new);
      if (rc) return rc;
      cur = dst;
      n_cur = n_new;
      S->stats.compactions++;
    }
  }
  if (cur != 0) {
    int rc = compact<T>(S, cur, -1, n_cur, nullptr);  // everything that is still in a work set goes home
    if (rc) return rc;
  }
  HIPCHK(hipEventRecord(S->ev_t1, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  float tms = 0.f;
  HIPCHK(hipEventElapsedTime(&tms, S->ev_t0, S->ev_t1));
  S->stats.instance_iterations = inst_iters;
  S->stats.n_unfinished = (int)n_live;
  S->stats.kernel_ms = kernel_ms;
  S->stats.total_ms = tms;
  return LOIKB_OK;
}

int run_main_loop(loikb_solver_impl* S)
{
  // UpdateMu's throw sites (hxx:632-640)
  if (S->opt.mu_update_strat != LOIKB_MU_DEFAULT && !(S->opt.flags & LOIKB_OPT_FIXED_ITERS)) {
    g_last_error = "[FirstOrderLoikOptimizedTpl::UpdateMu]: mu update strategy not yet implemented";
    return LOIKB_ERR_MU_STRATEGY;
  }
  return S->f32 ? run_main_loop_t<float>(S) : run_main_loop_t<double>(S);
}

template <typename T>
__global__ void k_limi(const T* __restrict__ cs, const JointDesc* __restrict__ jd, int nb, int B, int ldm,
                       double* __restrict__ out)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int i = 1; i <= nb; ++i) {
    T R[9], t[3];
    make_liMi<T>(jd[i], cs[(size_t)(2 * (i - 1)) * ldm + b], cs[(size_t)(2 * (i - 1) + 1) * ldm + b], R, t);
    double* o = out + ((size_t)b * nb + (i - 1)) * 12;
    for (int k = 0; k < 9; ++k) o[k] = (double)R[k];
    for (int k = 0; k < 3; ++k) o[9 + k] = (double)t[k];
  }
}

__global__ void k_status_extract(const int* __restrict__ status, int B, int mask, int* __restrict__ out)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = mask ? ((status[b] & mask) ? 1 : 0) : status[b];
}

}  // namespace

struct loikb_solver : loikb_solver_impl {};

extern "C" {

int loikb_version(void) { return LOIKB_VERSION; }

const char* loikb_last_error(void) { return g_last_error.c_str(); }

const char* loikb_status_string(int code)
{
  switch (code) {
  case LOIKB_OK: return "ok";
  case LOIKB_ERR_EQ_C_DIM:
    return "[IkProblemFormulation::IkProblemFormulation]: equality constraint dimension is not 6, problem formulation "
           "not supported !!!";
  case LOIKB_ERR_EQ_C_SIZE:
    return "[IkProblemFormulation::UpdateEqConstraints]: number of equality constraints doesn't match initialization!!!";
  case LOIKB_ERR_INEQ_DIM:
    return "IkProblemFormulation::UpdateIneqConstraints]: inequality constraint dimension has changed, this is not "
           "supported currently!!!";
  case LOIKB_ERR_NO_SUCH_CONSTRAINT:
    return "[IkProblemFormulation::UpdateEqConstraint]: constraint doesn't yet exist at link 'c_id' !!! ";
  case LOIKB_ERR_DUP_CONSTRAINT:
    return "[IkProblemFormulation::UpdateEqConstraint]: multiple constraint specification for the same link id, not "
           "supported, terminating !!!";
  case LOIKB_ERR_MU_STRATEGY: return "[FirstOrderLoikOptimizedTpl::UpdateMu]: mu update strategy not supported";
  case LOIKB_ERR_MODEL:
    return "[IkProblemFormulation::IkProblemFormulation]: nb does not equal to nj - 1, robot model not supported !!!";
  case LOIKB_ERR_ARG: return "invalid argument";
  case LOIKB_ERR_HIP: return "HIP runtime error";
  case LOIKB_ERR_NO_DEVICE: return "no HIP device";
  case LOIKB_ERR_HREF_NOT_SYMMETRIC: return "H_ref must be symmetric";
  case LOIKB_ERR_STATE: return "Solve() called before SolveInit()";
  default: return "unknown";
  }
}

int loikb_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int loikb_create(const loikb_model_desc* model, const loikb_options* opts, loikb_solver** out)
{
  if (!model || !opts || !out) return LOIKB_ERR_ARG;
  if (opts->eq_c_dim != 6) return LOIKB_ERR_EQ_C_DIM;
  if (opts->batch < 1 || opts->num_eq_c < 0) { g_last_error = "batch must be >= 1"; return LOIKB_ERR_ARG; }
  loikb_solver* S = new loikb_solver();
  int rc = build_schedule(S, model);
  if (rc) { delete S; return rc; }
  S->opt = *opts;
  S->B = opts->batch;
  S->ld = ((S->B + WAVE - 1) / WAVE) * WAVE;
  S->nc = opts->num_eq_c;
  S->f32 = opts->precision == LOIKB_F32;
  S->esz = S->f32 ? 4 : 8;
  S->device = opts->device;
  if (loikb_device_count() <= S->device) {
    g_last_error = "no HIP device available for the requested ordinal";
    delete S;
    return LOIKB_ERR_NO_DEVICE;
  }
  auto fail = [&](int code) { loikb_destroy(S); return code; };
#define TRY(x) do { int _rc = (x); if (_rc) return fail(_rc); } while (0)
#define HIPTRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) { g_last_error = std::string(#x) + ": " + hipGetErrorString(_e); return fail(LOIKB_ERR_HIP); } } while (0)
  HIPTRY(hipSetDevice(S->device));
  HIPTRY(hipEventCreate(&S->ev_k0));
  HIPTRY(hipEventCreate(&S->ev_k1));
  HIPTRY(hipEventCreate(&S->ev_t0));
  HIPTRY(hipEventCreate(&S->ev_t1));
  HIPTRY(hipHostMalloc((void**)&S->h_counters, 2 * sizeof(unsigned int)));
  void* tmp = nullptr;
  TRY(alloc_dev(S, &tmp, sizeof(JointDesc) * S->nj)); S->d_jd = (JointDesc*)tmp;
  TRY(alloc_dev(S, &tmp, sizeof(int) * S->nj)); S->d_idx_q = (int*)tmp;
  TRY(alloc_dev(S, &tmp, 2 * sizeof(unsigned int))); S->d_counters = (unsigned int*)tmp;
  TRY(alloc_set(S, 0, S->ld));
  HIPTRY(hipMemcpyAsync(S->d_idx_q, S->idx_q.data(), sizeof(int) * S->nj, hipMemcpyHostToDevice, S->stream));
  TRY(upload_jd(S));
  TRY(reset_solver(S));
  TRY(invalidate_h_cache(S));
  HIPTRY(hipStreamSynchronize(S->stream));
#undef TRY
#undef HIPTRY
  *out = S;
  return LOIKB_OK;
}

int loikb_destroy(loikb_solver* S)
{
  if (!S) return LOIKB_OK;
  (void)hipSetDevice(S->device);
  for (auto& a : S->allocs) (void)hipFree(a.p);
  if (S->d_stage) (void)hipFree(S->d_stage);
  if (S->h_counters) (void)hipHostFree(S->h_counters);
  if (S->ev_k0) (void)hipEventDestroy(S->ev_k0);
  if (S->ev_k1) (void)hipEventDestroy(S->ev_k1);
  if (S->ev_t0) (void)hipEventDestroy(S->ev_t0);
  if (S->ev_t1) (void)hipEventDestroy(S->ev_t1);
  delete S;
  return LOIKB_OK;
}

int loikb_set_stream(loikb_solver* S, void* hip_stream)
{
  if (!S) return LOIKB_ERR_ARG;
  S->stream = (hipStream_t)hip_stream;
  return LOIKB_OK;
}

int loikb_solve_init(loikb_solver* S, const double* q, const double* H_ref, const double* v_ref, const int* c_ids,
                     int nc, const double* Ais, const double* bis, const double* lb, const double* ub, int nbound,
                     int in_flags)
{
  if (!S || !q || !H_ref || !v_ref || (nc > 0 && (!c_ids || !Ais || !bis)) || !lb || !ub) return LOIKB_ERR_ARG;
  HIPCHK(hipSetDevice(S->device));
  int rc;
  // problem_.Reset(); ik_id_data_.Reset(warm_start); ResetSolver()  (hpp:345-352)
  if ((rc = data_reset(S, S->opt.warm_start))) return rc;
  if ((rc = reset_solver(S))) return rc;
  if ((rc = set_problem(S, H_ref, v_ref, c_ids, nc, Ais, bis, lb, ub, nbound, in_flags))) return rc;
  if ((rc = fwd_pass_init(S, q, in_flags))) return rc;
  HIPCHK(hipStreamSynchronize(S->stream));
  return LOIKB_OK;
}

int loikb_solve(loikb_solver* S)
{
  if (!S) return LOIKB_ERR_ARG;
  if (!S->have_problem) { g_last_error = "Solve() before SolveInit()"; return LOIKB_ERR_STATE; }
  HIPCHK(hipSetDevice(S->device));
  int rc;
  if ((rc = data_reset_recursion(S))) return rc;
  if ((rc = reset_solver(S))) return rc;
  return run_main_loop(S);
}

int loikb_solve_full(loikb_solver* S, const double* q, const double* H_ref, const double* v_ref, const int* c_ids,
                     int nc, const double* Ais, const double* bis, const double* lb, const double* ub, int nbound,
                     int in_flags)
{
  int rc = loikb_solve_init(S, q, H_ref, v_ref, c_ids, nc, Ais, bis, lb, ub, nbound, in_flags);
  if (rc) return rc;
  return run_main_loop(S);
}

int loikb_solve_tailored(loikb_solver* S, const double* q, int c_id, const double* Ai, const double* bi, int in_flags)
{
  if (!S || !q || !Ai || !bi) return LOIKB_ERR_ARG;
  if (!S->have_problem) { g_last_error = "tailored Solve() before SolveInit()"; return LOIKB_ERR_STATE; }
  HIPCHK(hipSetDevice(S->device));
  int rc;
  if ((rc = data_reset(S, S->opt.warm_start))) return rc;
  if ((rc = reset_solver(S))) return rc;
  // problem_.UpdateEqConstraint(c_id, Ai, bi), ik-id-description-optimized.hpp:178-218
  int found = -1, count = 0;
  for (int c = 0; c < S->nc; ++c)
    if (S->active_ids[c] == c_id) { if (found < 0) found = c; ++count; }
  if (found < 0) { g_last_error = loikb_status_string(LOIKB_ERR_NO_SUCH_CONSTRAINT); return LOIKB_ERR_NO_SUCH_CONSTRAINT; }
  if (count > 1) { g_last_error = loikb_status_string(LOIKB_ERR_DUP_CONSTRAINT); return LOIKB_ERR_DUP_CONSTRAINT; }
  const bool dev = in_flags & LOIKB_IN_DEVICE;
  const bool a_shared_in = in_flags & LOIKB_A_SHARED;
  if (a_shared_in != S->a_shared) {
    g_last_error = "tailored solve: A sharing mode must match SolveInit";
    return LOIKB_ERR_ARG;
  }
  const size_t ld = S->ld, e = S->esz;
  void* Adst = (char*)S->set[0].A + (a_shared_in ? (size_t)36 * found * e : (size_t)36 * found * ld * e);
  if ((rc = upload_aos(S, Ai, 36, Adst, dev && !a_shared_in, a_shared_in))) return rc;
  if (a_shared_in && (rc = upload_shared_AtA(S, Ai, found))) return rc;
  void* bdst = (char*)S->set[0].b + (size_t)6 * found * ld * e;
  if (in_flags & LOIKB_B_SHARED) rc = upload_broadcast(S, bi, 6, bdst);
  else rc = upload_aos(S, bi, 6, bdst, dev, false);
  if (rc) return rc;
  if ((rc = constraint_products(S, found, found + 1, true))) return rc;
  if ((rc = fwd_pass_init(S, q, in_flags))) return rc;
  return run_main_loop(S);
}

int loikb_set_max_iter(loikb_solver* S, int v) { if (!S) return LOIKB_ERR_ARG; S->opt.max_iter = v; return LOIKB_OK; }
int loikb_set_rho(loikb_solver* S, double v)
{
  if (!S) return LOIKB_ERR_ARG;
  S->opt.rho = v;
  return invalidate_h_cache(S);
}
int loikb_set_mu(loikb_solver* S, double v) { if (!S) return LOIKB_ERR_ARG; S->opt.mu = v; return LOIKB_OK; }
int loikb_set_tol(loikb_solver* S, double a, double r)
{
  if (!S) return LOIKB_ERR_ARG;
  S->opt.tol_abs = a; S->opt.tol_rel = r;
  return LOIKB_OK;
}
int loikb_set_tol_primal_inf(loikb_solver* S, double v) { if (!S) return LOIKB_ERR_ARG; S->opt.tol_primal_inf = v; return LOIKB_OK; }
int loikb_set_tol_tail_solve(loikb_solver* S, double v) { if (!S) return LOIKB_ERR_ARG; S->opt.tol_tail_solve = v; return LOIKB_OK; }
int loikb_set_warm_start(loikb_solver* S, int v) { if (!S) return LOIKB_ERR_ARG; S->opt.warm_start = v; return LOIKB_OK; }

int loikb_batch(const loikb_solver* S) { return S ? S->B : 0; }
int loikb_nv(const loikb_solver* S) { return S ? S->nv : 0; }
int loikb_njoints(const loikb_solver* S) { return S ? S->nj : 0; }

int loikb_get_stats(loikb_solver* S, loikb_stats* out)
{
  if (!S || !out) return LOIKB_ERR_ARG;
  *out = S->stats;
  return LOIKB_OK;
}

int loikb_get(loikb_solver* S, int field, void* out, int out_flags)
{
  if (!S || !out) return LOIKB_ERR_ARG;
  HIPCHK(hipSetDevice(S->device));
  const bool to_dev = out_flags & LOIKB_OUT_DEVICE;
  const void* src = nullptr;
  int n = 0;
  bool is_int = false;
  int mask = 0;
  switch (field) {
  case LOIKB_F_Z: src = S->set[0].z; n = S->nb; break;
  case LOIKB_F_NU: src = S->set[0].nu; n = S->nb; break;
  case LOIKB_F_W: src = S->set[0].w; n = S->nb; break;
  case LOIKB_F_STF_PLUS_W: src = S->set[0].s; n = S->nb; break;
  case LOIKB_F_R: src = S->set[0].rr; n = S->nb; break;
  case LOIKB_F_DINV: src = S->set[0].dinv; n = S->nb; break;
  case LOIKB_F_VIS: src = S->set[0].v; n = 6 * S->nb; break;
  case LOIKB_F_FIS: src = S->set[0].f; n = 6 * S->nb; break;
  case LOIKB_F_G: src = S->set[0].g; n = 6 * S->nb; break;
  case LOIKB_F_PIS: src = S->set[0].p; n = 6 * S->nb; break;
  case LOIKB_F_UDINV: src = S->set[0].ud; n = 6 * S->nb; break;
  case LOIKB_F_HIS: src = S->set[0].H; n = 21 * S->nb; break;
  case LOIKB_F_YIS: src = S->set[0].y; n = 6 * S->nc; break;
  case LOIKB_F_ATY: src = S->set[0].aty; n = 6 * S->nc; break;
  case LOIKB_F_LIMI: n = 12 * S->nb; break;
  case LOIKB_F_ITER: is_int = true; break;
  case LOIKB_F_STATUS: is_int = true; break;
  case LOIKB_F_CONVERGED: is_int = true; mask = ST_CONVERGED; break;
  case LOIKB_F_PRIMAL_INFEASIBLE: is_int = true; mask = ST_PRIMAL_INF; break;
  case LOIKB_F_MU: src = S->set[0].mu; n = 1; break;  // per-instance mu_ (== mu0 right after ResetSolver)
  default:
    static_assert(LOIKB_F_TAIL_SOLVE_ITER - LOIKB_F_PRIMAL_RESIDUAL + 1 == NSCAL, "scalar field ids out of sync");
    if (field >= LOIKB_F_PRIMAL_RESIDUAL && field <= LOIKB_F_TAIL_SOLVE_ITER) {
      const int row = field - LOIKB_F_PRIMAL_RESIDUAL;  // same order as the SC_* enum
      src = (const char*)S->set[0].scal + (size_t)row * S->ld * S->esz;
      n = 1;
    } else {
      return LOIKB_ERR_ARG;
    }
  }
  if (is_int) {
    int* dst = (int*)out;
    if (!to_dev) {
      int rc = ensure_stage(S, sizeof(int) * (size_t)S->B);
      if (rc) return rc;
      dst = (int*)S->d_stage;
    }
    if (field == LOIKB_F_ITER)
      HIPCHK(hipMemcpyAsync(dst, S->set[0].iter, sizeof(int) * (size_t)S->B, hipMemcpyDeviceToDevice, S->stream));
    else {
      hipLaunchKernelGGL(k_status_extract, grid1(S->B), dim3(256), 0, S->stream, S->set[0].status, S->B, mask, dst);
      HIPCHK(hipGetLastError());
    }
    if (!to_dev) HIPCHK(hipMemcpyAsync(out, dst, sizeof(int) * (size_t)S->B, hipMemcpyDeviceToHost, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
    return LOIKB_OK;
  }
  const size_t bytes = sizeof(double) * (size_t)S->B * n;
  double* dst = (double*)out;
  if (!to_dev) {
    int rc = ensure_stage(S, bytes);
    if (rc) return rc;
    dst = (double*)S->d_stage;
  }
  if (field == LOIKB_F_LIMI) {
    if (S->f32)
      hipLaunchKernelGGL(k_limi<float>, grid1(S->B), dim3(256), 0, S->stream, (const float*)S->set[0].cs, S->d_jd, S->nb, S->B,
                         S->ld, dst);
    else
      hipLaunchKernelGGL(k_limi<double>, grid1(S->B), dim3(256), 0, S->stream, (const double*)S->set[0].cs, S->d_jd, S->nb, S->B,
                         S->ld, dst);
  } else if (S->f32) {
    hipLaunchKernelGGL(k_soa_to_aos<float>, grid1(S->B), dim3(256), 0, S->stream, (const float*)src, n, S->B, S->ld, dst);
  } else {
    hipLaunchKernelGGL(k_soa_to_aos<double>, grid1(S->B), dim3(256), 0, S->stream, (const double*)src, n, S->B, S->ld, dst);
  }
  HIPCHK(hipGetLastError());
  if (!to_dev) HIPCHK(hipMemcpyAsync(out, dst, bytes, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  return LOIKB_OK;
}

}  // extern "C"
