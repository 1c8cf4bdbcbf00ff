// loik_tail.hpp -- cooperative "tail" kernel: one problem instance per lane GROUP, ONE joint per lane.
//
// Why it exists: the ADMM iteration counts of a batch are heavy-tailed (on the Talos workload the median is ~26
// iterations, 1.2 % of the instances run to max_iter = 1000 because the reference's DEFAULT penalty update keeps
// flipping mu between two decades).  With one instance per lane (k_solve) an iteration costs 30-55 us of HBM round
// trips however few instances are left, so the stragglers set the batch time.
// Here a group of G = 8/16/32/64 lanes (G >= nb, 64/G instances per wavefront) splits ONE instance by joint: the
// whole ADMM state of a joint lives in the registers / LDS of its lane for as long as the group works on the instance
// (zero HBM traffic per iteration).  Only the two true recursions over the tree (p leaf -> root, (nu, v) root -> leaf)
// are loops over the tree depth (Talos: 10 steps instead of 32 joint visits), and they are branch-free: every lane
// recomputes its value from its children's / parent's exchange rows at every step.  Everything else is per-joint work
// done once by all lanes; inf-norms and dot products are folded through LDS.  ~10 us per iteration.
// H_i / UDinv / Dinv are cached for the TWO most recent values of mu, so an instance whose penalty flips between
// two decades (the typical straggler) does not repeat the H-recursion.
// The live instances are handed out through an atomic queue: a group that finishes one takes the next.
//
// The arithmetic per joint is the same as in loik_device.hpp (same helpers, same reference citations); only
// the order in which children contributions / norm maxima are combined differs, so results agree with k_solve
// and the CPU oracle to rounding (not bit for bit).  Used when nb <= 64.
// loik_lean.hpp holds the variant of this kernel that runs two wavefronts per SIMD (the default engine for whole
// batches); this one rebuilds H itself and serves what the lean kernel cannot: more than two task constraints, more than four
// children per joint, LOIKB_OPT_NO_H_CACHE, fewer than 64 instances, and instances whose mu leaves the lean kernel's
// precomputed decades.
#pragma once

#include "loik_device.hpp"

namespace loikb {

struct TailTopo {
  int depth;        // 1 for children of the universe
  int child_start;  // into child_list (children in DECREASING joint index, the order of the reference's sweep)
  int nchild;
  int pad;
};

constexpr int XS = 30;  // scalars per lane in the exchange buffer: 21 (H) + 1 pad + 6 (p / v / f) + 2 pad.  The row stride is
                        // an ODD number of 16-byte slots (15): b128 accesses of consecutive lanes do not collide
                        // (with 28 = 14 slots the PMC counters showed 36 % of the LDS cycles in bank conflicts)
constexpr int XC = 22;  // first of the 6 exchange columns for p / v (16-byte aligned: read as three b128)
constexpr int XROWS = WAVE + 1;  // one exchange row per lane + a row of zeros ("no parent" / "no such child")
constexpr int HS = 22;  // scalars per lane and mu-slot in the H store: H[21], Dinv  (4 wavefronts/CU fit in 160 KiB)
constexpr int TAIL_WAVES = 4;  // wavefronts per workgroup of the tail kernel (one per SIMD of a CU; fewer if their LDS does not fit)
constexpr int CD = 82;  // per-constraint LDS block: A[36] AtA[21] pad b[6] Atb[6] y[6] aty[6]
enum : int { CD_A = 0, CD_ATA = 36, CD_PAD = 57 /* group lane of the constrained joint */, CD_B = 58, CD_ATB = 64, CD_Y = 70,
             CD_ATY = 76 };

// The kernel runs one wavefront per workgroup: LDS operations of a wavefront execute in program order, so an exchange
// between lanes needs no hardware barrier -- only a fence that keeps the compiler from reordering the LDS accesses.
__device__ __forceinline__ void tail_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// reductions over an aligned group of G lanes (G a power of two <= 64)
template <typename T>
__device__ __forceinline__ T group_max(T x, int G)
{
  for (int off = G >> 1; off > 0; off >>= 1) x = tmax(x, __shfl_xor(x, off));
  return x;
}
template <typename T>
__device__ __forceinline__ T group_sum(T x, int G)
{
  for (int off = G >> 1; off > 0; off >>= 1) x += __shfl_xor(x, off);
  return x;
}

// LDS of ONE wavefront of the tail kernel (rounded to 16 B: the next wavefront's slice starts behind it)
template <typename T>
__host__ __device__ __forceinline__ size_t tail_lds_bytes(int nc, int G)
{
  return ((((size_t)XROWS * XS + 2 * (size_t)WAVE * HS + (size_t)(WAVE / G) * nc * CD) * sizeof(T)) + 15) & ~(size_t)15;
}

// Phase timeline of one wavefront (diagnostic build only: -DLOIKB_TAIL_PROF, scripts/tail_phase_profile.py)
#ifdef LOIKB_TAIL_PROF
// (a profile build of TWO translation units -- the flat kernels with their own code generation, as shipped -- has two copies of these: the
//  flat unit's are its own (static), read through loikb_flat_prof_read, loik_flat_kernels.hip; the debug entry points take whichever copy
//  holds the last launch)
#ifdef LOIKB_FLAT_KERNELS_TU
#define LOIKB_PROF_LINKAGE static
#else
#define LOIKB_PROF_LINKAGE
#endif
LOIKB_PROF_LINKAGE __device__ unsigned long long g_tail_prof_all[32];  // the same phases summed over ALL wavefronts of the launch ([8]: iterations, [9]: wavefronts)
LOIKB_PROF_LINKAGE __device__ unsigned long long g_tail_prof[32];  // [0..8) phases, [8] iterations, [9] clock, [10..14) finer phases of k_lean / k_flat2, [14..22) k_flat2: an instance's load / store
// per wavefront of the last lean launch: [0] wall clock (100 MHz) at start, [1] when the wavefront last had an instance,
// [2] at exit, [3] wavefront-iterations, [4] wavefront-iterations with both groups active, [5] instance switches
LOIKB_PROF_LINKAGE __device__ unsigned long long g_wave_dbg[4096][6];
#define TAIL_TP(k) { const unsigned long long tn_ = clock64(); prof_[k] += tn_ - tprev_; tprev_ = tn_; }
#else
#define TAIL_TP(k)
#endif

// acc[0..6) += the 6-vectors at column `off` of the exchange rows chl[0..NCH): all LDS reads are issued before the
// first add (a per-child "if" makes the compiler wait for each child's rows in turn -- three LDS round trips per tree
// level on a humanoid); the adds keep the child order.  A missing child is the zero row.
// a . b over a 6-vector as the sum of its linear and angular halves: two independent chains of three multiply-adds
// instead of one of six (a wavefront runs alone on its SIMD in this kernel: the dependent fp64 latency is exposed)
template <typename T>
__device__ __forceinline__ T dot6_halves(const T* a, const T* b)
{
  const T lin = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  const T ang = a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
  return lin + ang;
}

template <typename T, int NCH>
__device__ __forceinline__ void gather_rows(const T* xch, const int* chl, int off, T* acc)
{
  T x[NCH][6];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) x[c][k] = xch[chl[c] * XS + off + k];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] += x[c][k];
}
template <typename T>
__device__ __forceinline__ void gather_rows_n(int n, const T* xch, const int* chl, int off, T* acc)
{
  switch (n) {  // uniform
  case 0: break;
  case 1: gather_rows<T, 1>(xch, chl, off, acc); break;
  case 2: gather_rows<T, 2>(xch, chl, off, acc); break;
  case 3: gather_rows<T, 3>(xch, chl, off, acc); break;
  default: gather_rows<T, 4>(xch, chl, off, acc); break;
  }
}

template <typename T, bool HDIAG>
__global__ void __launch_bounds__(WAVE * TAIL_WAVES)
k_tail(const Params<T> P, const Bufs<T> Bf, const JointDesc* __restrict__ jd, const TailTopo* __restrict__ topo,
       const int* __restrict__ child_list, int maxdepth, int maxchild, const int* __restrict__ slots, int nslots, int G)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Layout& L = P.L;
  // TAIL_WAVES independent wavefronts per workgroup (they never talk to each other: the workgroup only makes the
  // kernel claim whole CUs, so that solve-kernel workgroups of another stream are not locked out of a CU by a
  // stray tail wavefront); every wavefront has its own slice of the LDS
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t wave_lds = tail_lds_bytes<T>(L.nc, G);
  T* xch = reinterpret_cast<T*>(smem_raw + wv * wave_lds);  // [WAVE][XS]  exchange between a joint and its parent / children
  T* hst = xch + XROWS * XS;                // [2][WAVE][HS]   this joint's H (pre-projection) for two values of mu
  T* cd = hst + 2 * WAVE * HS;              // [64/G][nc][CD]  constraint data of every instance of the wavefront
  const int lane = threadIdx.x & (WAVE - 1);
  const int sub = lane / G, jlane = lane % G, gbase = sub * G;
  const bool isj_lane = jlane < L.nb;
  const int jl = isj_lane ? jlane : 0;
  T* cdi = cd + (size_t)sub * L.nc * CD;

  // ---- per-lane joint description (VGPRs: every lane owns a different joint) ------------------------------
  const JointDesc d = jd[jl + 1];
  const TailTopo tp = topo[jl + 1];
  const int depth = isj_lane ? tp.depth : 0;
  const bool rev = d.flags & JF_REVOLUTE;
  const T mass = (d.flags & JF_MASSLESS) ? T(0) : T(1);  // chain link of a multi-DoF joint: no cost of its own
  // per-link references (UpdateReferences): this lane's row of the table, else the broadcast pair in P (uniform branch)
  const T* const hrow = P.href_tab ? P.href_tab + (size_t)(jl + 1) * HREF_ROW : nullptr;
  const bool has_parent = !(d.flags & JF_PARENT_ROOT);
  const int plane = gbase + d.parent - 1;  // parent's lane
  const T ax0 = (T)d.axis[0], ax1 = (T)d.axis[1], ax2 = (T)d.axis[2];
  // rows (lanes) of this joint's children in the exchange buffer: kept in registers, the sweeps must not go to
  // global memory for them (children beyond NCH_REG are looked up in child_list)
  constexpr int NCH_REG = 4;
  int chl[NCH_REG];
#pragma unroll
  for (int c = 0; c < NCH_REG; ++c) chl[c] = c < tp.nchild ? gbase + child_list[tp.child_start + c] : WAVE;  // WAVE: the zero row
  const int prow = has_parent ? plane : WAVE;  // exchange row of the parent's velocity (zero row under the universe)
  // S_i as a 6-vector: v' + S nu and S^T p become plain multiply-adds, no branch on the joint type in the loops
  T Sv[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) { Sv[k] = rev ? T(0) : (T)d.axis[k]; Sv[3 + k] = rev ? (T)d.axis[k] : T(0); }
  // helical joints (S = [pitch a; a]): a wavefront-uniform switch, so that models without one pay nothing
  const bool hel = (d.flags & JF_HELICAL) != 0;
  const bool hel_any = __any(hel ? 1 : 0) != 0;
  const T ph = hel ? (T)d.pitch : T(0);
  if (hel) {
#pragma unroll
    for (int k = 0; k < 3; ++k) Sv[k] = ph * (T)d.axis[k];
  }
  if (lane < XS) xch[WAVE * XS + lane] = T(0);

  // ---- the instance a group works on (all of this is reloaded when the group takes the next one) -----------------
  bool has_inst = false, isj = false, done = true, any_iter = false;
  int slot = 0;
  char *ip = Bf.tiles, *rec = Bf.tiles, *srec = Bf.tiles;
  T R[9], t[3], v[6], f[6], g[6], UD[6], UDo[6], p[6];
  T w = T(0), z = T(0), nu = T(0), s = T(0), r = T(0), dinv = T(0), lbi = T(0), ubi = T(0);
  T mu = T(1), tg_in = T(-1), mu_h = T(-1), mu_o = T(-1), mu_v = T(-1), bnorm = T(0), st_y = T(0);
  int kexp = 0, hsl = 0, iter = 0, status = ST_DONE, tail_iter = 0, c1 = 0, c2 = 0, nflip = 0;
  T tol_p = T(0), tol_d = T(0), dyqp = T(0), atdy = T(0), ubp = T(0), lbm = T(0);
  unsigned int my_iters = 0;
  // last-iteration scalars (for the final dump)
  T primal = T(0), dual = T(0), pr_task = T(0), pr_slack = T(0), dual_v = T(0), stf_w_inf = T(0), dx = T(0), dz = T(0);
  T n_dfis = T(0), n_dyis = T(0), n_dw = T(0), n_dvis = T(0), n_dnu = T(0), n_av = T(0), n_nu = T(0), n_hrefv = T(0),
    n_g = T(0);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = T(0);
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) { v[k] = f[k] = g[k] = UD[k] = UDo[k] = p[k] = T(0); }
  unsigned int n_wave_iters = 0, n_h_iters = 0;  // diagnostics: wavefront-iterations, those with an H rebuild

  // The live instances are handed out through an atomic queue head: a group that finishes its instance stores it and
  // takes the next one from the list, so every lane group stays busy until the list is empty although instances
  // finish at very different iterations (1.2 % run 1000 iterations, the median is 26), and every instance is loaded
  // and stored exactly once.
  auto fetch = [&]() -> int {
    int nx = 0;
    if (jlane == 0) nx = (int)atomicAdd(&Bf.counters[7], 1u);
    return __shfl(nx, gbase);
  };
  // joint j of instance list[idx] -> lane j of the group
  auto load_instance = [&](int idx) {
    has_inst = idx < nslots;
    isj = has_inst && isj_lane;
    slot = slots[has_inst ? idx : 0];
    ip = lane_ptr<T>(Bf.tiles, L, slot);  // the instance's lane pointer: identical within the group
    rec = ip + (size_t)jl * JREC * pair_bytes<T>();
    srec = ip + (size_t)L.off_s * pair_bytes<T>();
    {
      const typename Vec2<T>::type cs = ldp<T>(rec, JP_CS), wz = ldp<T>(rec, JP_WZ), nus = ldp<T>(rec, JP_NUS);
      joint_xform<T>(d, rec, cs.x, cs.y, R, t);  // liMi is kept in registers for the whole solve of the instance
      ld6<T>(rec, JP_V, v);
      ld6<T>(rec, JP_F, f);
      ld6<T>(rec, JP_G, g);
      w = wz.x; z = wz.y; nu = nus.x; s = nus.y;
      if (P.mode & MODE_BND_SHARED) {
        lbi = Bf.uni[L.nc * 57 + jl];
        ubi = Bf.uni[L.nc * 57 + L.nb + jl];
      } else {
        const typename Vec2<T>::type lu = ldp<T>(rec, JP_LBUB);
        lbi = lu.x; ubi = lu.y;
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) { UD[k] = T(0); UDo[k] = T(0); p[k] = T(0); }
      r = T(0); dinv = T(0);
    }
    // constraint blocks -> LDS (the lanes of a group cooperate)
    for (int c = 0; c < L.nc; ++c) {
      const char* crec = ip + (size_t)(L.off_c + c * L.crec) * pair_bytes<T>();
      for (int e = jlane; e < CD; e += G) {
        T val = T(0);
        if (e < 36) {
          val = (P.mode & MODE_A_SHARED) ? Bf.uni[c * 36 + e]
                                         : *reinterpret_cast<const T*>(crec + (size_t)(CP_A + e / 2) * pair_bytes<T>() + (e & 1) * sizeof(T));
        } else if (e < 57) {
          const int q = e - 36;
          val = (P.mode & MODE_A_SHARED) ? Bf.uni[L.nc * 36 + c * 21 + q]
                                         : *reinterpret_cast<const T*>(crec + (size_t)(CP_ATA + q / 2) * pair_bytes<T>() + (q & 1) * sizeof(T));
        } else if (e >= CD_B) {
          const int q = e - CD_B;
          const int which = q / 6, k = q % 6;
          const int pair = which == 0 ? CP_B : which == 1 ? CP_ATB : which == 2 ? CP_Y : CP_ATY;
          val = *reinterpret_cast<const T*>(crec + (size_t)(pair + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T));
        }
        if (e != CD_PAD) cdi[c * CD + e] = val;
      }
    }
    if (isj_lane && d.cslot >= 0) cdi[d.cslot * CD + CD_PAD] = (T)jlane;  // which lane of the group owns the constrained joint
    // per-instance solver scalars: every lane of the group reads the same words
    const typename Vec2<T>::type mu2 = ldp<T>(srec, SP_MU), bi2 = ldp<T>(srec, SP_BI), st2 = ldp<T>(srec, SP_ST);
    mu = mu2.x;
    kexp = (int)mu2.y;         // mu = mu0 * 10^kexp
    tg_in = ldp<T>(srec, SP_TAG).x;
    nflip = (int)ldp<T>(srec, SP_FLIP).x;
    mu_h = T(-1); mu_o = T(-1);  // mu of the current / the other H slot: nothing cached yet
    mu_v = T(-1);                // mu of the victim slot (third level of the cache, see the main loop)
    hsl = 0;                     // current H slot
    // (H_i is not kept in HBM -- k_solve gets f_i from the force-balance recursion -- so the first iteration on an
    //  instance rebuilds its H cache; this kernel keeps f_i = H_i v_i + p_i: H_i sits in LDS anyway and the extra
    //  recursion over the tree levels would cost more than the 36 multiply-adds per joint.)
    bnorm = bi2.x;
    iter = (int)bi2.y;
    status = has_inst ? (int)st2.x : ST_DONE;
    st_y = st2.y;
    tol_p = ld_scal<T>(srec, SC_TOL_PRIMAL); tol_d = ld_scal<T>(srec, SC_TOL_DUAL);
    tail_iter = (int)ld_scal<T>(srec, SC_TAIL_ITER);
    dyqp = ld_scal<T>(srec, SC_DELTA_Y_QP); atdy = ld_scal<T>(srec, SC_AT_DELTA_Y_QP);
    ubp = ld_scal<T>(srec, SC_UB_DY_PLUS); lbm = ld_scal<T>(srec, SC_LB_DY_MINUS);
    c1 = (int)ld_scal<T>(srec, SC_COND1); c2 = (int)ld_scal<T>(srec, SC_COND2);
    tail_sync();
    done = (status & ST_DONE) != 0;
    if (!done && !(status & ST_TAIL) && iter + 1 >= P.max_iter) { done = true; status |= ST_DONE; }
    my_iters = 0;
    any_iter = false;
  };
  // write the group's instance back (same slot)
  auto store_instance = [&]() {
    if (isj) {
      st6<T>(rec, JP_V, v);
      st6<T>(rec, JP_F, f);
      st6<T>(rec, JP_G, g);
      stp<T>(rec, JP_WZ, w, z);
      stp<T>(rec, JP_NUS, nu, s);
      if (any_iter) {
        // inter-sweep temporaries of the LAST iteration (pis, UDinv, Dinv, r), as upstream leaves them: here the
        // accumulated p_i itself (flagged ST_PFULL below; k_solve stores p_i^base in that slot)
        st6<T>(rec, JP_P, p);
        st6<T>(rec, JP_UD, UD);
        stp<T>(rec, JP_R, r, dinv);
      }
    }
    tail_sync();
    if (has_inst) {
      for (int c = 0; c < L.nc; ++c) {
        char* crec = ip + (size_t)(L.off_c + c * L.crec) * pair_bytes<T>();
        if (jlane < 6) {
          const int k = jlane;
          *reinterpret_cast<T*>(crec + (size_t)(CP_Y + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T)) = cdi[c * CD + CD_Y + k];
          *reinterpret_cast<T*>(crec + (size_t)(CP_ATY + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T)) = cdi[c * CD + CD_ATY + k];
        }
      }
      if (jlane == 0) {
        stp<T>(srec, SP_MU, mu, (T)kexp);
        // the UDinv / Dinv written above belong to mu_h, the mu of the last executed iteration
        stp<T>(srec, SP_TAG, any_iter ? mu_h : tg_in, T(0));
        stp<T>(srec, SP_BI, bnorm, (T)iter);
      stp<T>(srec, SP_FLIP, (T)nflip, T(0));
        stp<T>(srec, SP_ST, (T)(any_iter ? (status | ST_PFULL) : status), any_iter ? mu_h : st_y);
        if (any_iter) {
          stp<T>(srec, SP_SCAL + 0, primal, dual);
          stp<T>(srec, SP_SCAL + 1, pr_task, pr_slack);
          stp<T>(srec, SP_SCAL + 2, dual_v, stf_w_inf);
          stp<T>(srec, SP_SCAL + 3, tol_p, tol_d);
          stp<T>(srec, SP_SCAL + 4, mu, P.mu_scale * mu);
          stp<T>(srec, SP_SCAL + 5, mu, dx);
          stp<T>(srec, SP_SCAL + 6, dz, dyqp);
          stp<T>(srec, SP_SCAL + 7, atdy, ubp);
          stp<T>(srec, SP_SCAL + 8, lbm, n_dfis);
          stp<T>(srec, SP_SCAL + 9, n_dyis, n_dw);
          stp<T>(srec, SP_SCAL + 10, n_dvis, n_dnu);
          stp<T>(srec, SP_SCAL + 11, n_av, n_nu);
          stp<T>(srec, SP_SCAL + 12, n_hrefv, n_g);
          stp<T>(srec, SP_SCAL + 13, stf_w_inf, (T)c1);
          stp<T>(srec, SP_SCAL + 14, (T)c2, (T)tail_iter);
        }
        if (my_iters) atomicAdd(&Bf.counters[1], my_iters);
      }
    }
  };

  // the loops below stay in wavefront-uniform control flow (LDS exchanges inside); a group without work only masks its
  // updates with `act`
  load_instance(fetch());
#ifdef LOIKB_TAIL_PROF
  unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = clock64();
#endif
  while (__any(!done || has_inst)) {  // (done && has_inst: an instance that was already finished when it was fetched)
    const bool act = !done;
    const T mu_eq = P.mu_scale * mu, mu_in = mu;
    if (act) { ++iter; ++my_iters; any_iter = true; }

    // ---- H cache: two slots (two most recent mu); a flip back to the previous mu costs a register swap ----------
    if (act && (P.mode & MODE_CACHE_H) && (mu_h != mu)) {
      { const T tmp = mu_h; mu_h = mu_o; mu_o = tmp; }
      hsl ^= 1;
      dinv = hst[((size_t)hsl * WAVE + lane) * HS + 21];  // Dinv of the slot that becomes current
#pragma unroll
      for (int k = 0; k < 6; ++k) { const T tmp = UD[k]; UD[k] = UDo[k]; UDo[k] = tmp; }
    }
    T* hcur = hst + ((size_t)hsl * WAVE + lane) * HS;
    // ---- third level of the H cache: a victim slot in HBM.  Most long runners alternate between two decades of mu
    // (both LDS slots hit), but some cycle through three (k, k+1, k+2, k+1, k ...): with two slots every other move
    // is a rebuild -- a masked level loop with the H recursion, several times the cost of a plain iteration, paid by
    // both instances of the wavefront -- and exactly these instances run to max_iter and decide when the launch ends.
    // The joint's record in the tile is scratch while the instance lives in this kernel (store_instance rewrites
    // pairs JP_V .. JP_P): the slot that a rebuild would overwrite is parked there, and fetched back on a hit.
    if (act && (P.mode & MODE_CACHE_H) && mu_h != mu) {
      constexpr int VP = JP_V;  // 14 pairs: H (21) + Dinv, UDinv (6)
      static_assert(JP_P + 3 - JP_V == 14 && JP_F == JP_V + 3 && JP_G == JP_F + 3 && JP_WZ == JP_G + 3 && JP_NUS == JP_WZ + 1 &&
                    JP_P == JP_NUS + 1, "victim slot: pairs JP_V .. JP_P + 2 must be contiguous");
      if (mu_v == mu) {
        // hit: exchange the current slot with the victim, seven pairs at a time (loads before the stores they alias)
        if (isj) {
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            typename Vec2<T>::type in[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) in[k] = ldp<T>(rec, VP + 7 * half + k);
#pragma unroll
            for (int k = 0; k < 7; ++k) {
              const int e = 14 * half + 2 * k;  // entries e, e+1 of (H[0..21], UD[0..5])
              const T o0 = e < 22 ? hcur[e] : UD[e - 22], o1 = e + 1 < 22 ? hcur[e + 1] : UD[e + 1 - 22];
              stp<T>(rec, VP + 7 * half + k, o0, o1);
              if (e < 22) hcur[e] = in[k].x; else UD[e - 22] = in[k].x;
              if (e + 1 < 22) hcur[e + 1] = in[k].y; else UD[e + 1 - 22] = in[k].y;
            }
          }
          dinv = hcur[21];
        }
        { const T tmp = mu_h; mu_h = mu_v; mu_v = tmp; }
      } else if (mu_h >= T(0)) {
        // miss with a valid slot about to be overwritten: park it
        if (isj) {
#pragma unroll
          for (int k = 0; k < 14; ++k) {
            const int e = 2 * k;
            const T o0 = e < 22 ? hcur[e] : UD[e - 22], o1 = e + 1 < 22 ? hcur[e + 1] : UD[e + 1 - 22];
            stp<T>(rec, VP + k, o0, o1);
          }
        }
        mu_v = mu_h;
      }
    }
    const bool need_h = act && (!(P.mode & MODE_CACHE_H) || (mu_h != mu));
    ++n_wave_iters;
    n_h_iters += __any(need_h) ? 1u : 0u;

    // ================= leaf -> root: FwdPass1 + BwdPass (hxx:290-338, :31-81) =================================
    T hh[22];
    if (need_h) {
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b2 = a; b2 < 6; ++b2)
          hh[sym(a, b2)] = mass * ((a == b2 ? P.rho : T(0)) +
                                   ((HDIAG && a != b2) ? T(0) : (hrow ? hrow[6 * a + b2] : P.Href[6 * a + b2])));
    }
    if (act) {
#pragma unroll
      for (int k = 0; k < 6; ++k) p[k] = mass * (-P.rho * v[k] - (hrow ? hrow[36 + k] : P.Hv[k]));
      if (isj && d.cslot >= 0) {
        const T* c_ = cdi + d.cslot * CD;
        if (need_h) {
#pragma unroll
          for (int k = 0; k < 21; ++k) hh[k] += mu_eq * c_[CD_ATA + k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) p[k] += c_[CD_ATY + k] - mu_eq * c_[CD_ATB + k];
      }
    }
    // Two forms of the leaf -> root recursion.  While some instance of the wavefront rebuilds its H cache: the masked
    // level loop (only the lanes of a level work, the 21 entries of H travel with p).  Otherwise (95-98 % of the
    // iterations) the LEAN loop: every lane recomputes p_i = p_i^base + sum of its children's exchange rows at every
    // level -- a joint of height h is final after h levels and simply recomputes the same value afterwards -- so the
    // body is branch-free and select-free, all lanes active, no per-level bookkeeping of who is "at" the level.
    const bool lean = !__any(need_h) && maxchild <= NCH_REG;
    TAIL_TP(0)
    if (lean) {
      T pl[6], rl = T(0);
#pragma unroll
      for (int k = 0; k < 6; ++k) pl[k] = p[k];
      for (int lev = maxdepth; lev >= 1; --lev) {
#pragma unroll
        for (int k = 0; k < 6; ++k) pl[k] = p[k];  // p still holds p^base
        gather_rows_n<T>(maxchild, xch, chl, XC, pl);
        const T Stp = dot6_halves(Sv, pl);
        rl = (w - mu_in * z) + Stp;
        T pa[6], pc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) pa[k] = pl[k] - UD[k] * rl;
        act_force(R, t, pa, pc);
        tail_sync();  // every lane has read its children's rows of the previous level
#pragma unroll
        for (int k = 0; k < 6; ++k) xch[lane * XS + XC + k] = pc[k];
        tail_sync();
      }
      if (act) {
#pragma unroll
        for (int k = 0; k < 6; ++k) p[k] = pl[k];
        r = rl;
      }
    } else
    for (int lev = maxdepth; lev >= 1; --lev) {
      if (act && depth == lev) {
        // children contributions (deposited one level deeper), largest child index first as upstream
        for (int c = 0; c < maxchild; ++c) {
          if (c < tp.nchild) {
            const int crow = c == 0 ? chl[0] : c == 1 ? chl[1] : c == 2 ? chl[2] : c == 3 ? chl[3]
                                                                     : gbase + child_list[tp.child_start + c];
            const T* x = xch + crow * XS;
            if (need_h) {
#pragma unroll
              for (int k = 0; k < 21; ++k) hh[k] += x[k];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) p[k] += x[XC + k];
          }
        }
        T U[6];
        if (need_h) {
          if (rev) {
#pragma unroll
            for (int k = 0; k < 6; ++k) U[k] = hh[sym(k, 3)] * ax0 + hh[sym(k, 4)] * ax1 + hh[sym(k, 5)] * ax2;
            dinv = T(1) / ((ax0 * U[3] + ax1 * U[4] + ax2 * U[5]) + mu_in);
          } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) U[k] = hh[sym(k, 0)] * ax0 + hh[sym(k, 1)] * ax1 + hh[sym(k, 2)] * ax2;
            dinv = T(1) / ((ax0 * U[0] + ax1 * U[1] + ax2 * U[2]) + mu_in);
          }
          if (hel_any && hel) {
#pragma unroll
            for (int k = 0; k < 6; ++k) U[k] += ph * (hh[sym(k, 0)] * ax0 + hh[sym(k, 1)] * ax1 + hh[sym(k, 2)] * ax2);
            dinv = T(1) / (((ax0 * U[3] + ax1 * U[4] + ax2 * U[5]) + ph * (ax0 * U[0] + ax1 * U[1] + ax2 * U[2])) + mu_in);
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) UD[k] = U[k] * dinv;
#pragma unroll
          for (int k = 0; k < 21; ++k) hcur[k] = hh[k];  // pre-projection H for the forward sweep
          hcur[21] = dinv;
        }
        T Stp = rev ? (ax0 * p[3] + ax1 * p[4] + ax2 * p[5]) : (ax0 * p[0] + ax1 * p[1] + ax2 * p[2]);
        if (hel_any && hel) Stp += ph * (ax0 * p[0] + ax1 * p[1] + ax2 * p[2]);
        r = (w - mu_in * z) + Stp;
        if (has_parent) {
          T* x = xch + lane * XS;
          if (need_h) {
            T part[21];
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
              for (int b2 = a; b2 < 6; ++b2) hh[sym(a, b2)] -= UD[a] * U[b2];
            congr_sym(R, t, hh, part);
#pragma unroll
            for (int k = 0; k < 21; ++k) x[k] = part[k];
          }
          T pa[6], pc[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) pa[k] = p[k] - UD[k] * r;
          act_force(R, t, pa, pc);
#pragma unroll
          for (int k = 0; k < 6; ++k) x[XC + k] = pc[k];
        }
      }
      tail_sync();
    }
    if (need_h) mu_h = mu;
    TAIL_TP(1)

    // ================= root -> leaf: FwdPass2 + BoxProj + DualUpdate (hxx:102-163, :384-461) ==================
    // Only nu_i / v_i form a recursion over the tree: the level loop carries just that.  Everything else of the pass
    // (f_i = H_i v_i + p_i, the projections, the dual updates, the norms) is per-joint work, done ONCE by all lanes.
    // Lean form as above: every lane recomputes (nu_i, v_i) from its parent's exchange row at every level; a joint at
    // depth d is final after d levels.
    T vi[6], nui = T(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) vi[k] = T(0);
    for (int lev = 1; lev <= maxdepth; ++lev) {
      T vpar[6], vp[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) vpar[k] = xch[prow * XS + XC + k];
      actinv_motion(R, t, vpar, vp);  // hxx:125
      const T udv = dot6_halves(UD, vp);
      nui = -udv - dinv * r;          // hxx:127
#pragma unroll
      for (int k = 0; k < 6; ++k) vi[k] = vp[k] + Sv[k] * nui;  // hxx:133-134
      tail_sync();  // every lane has read its parent's row of the previous level
#pragma unroll
      for (int k = 0; k < 6; ++k) xch[lane * XS + XC + k] = vi[k];
      tail_sync();
    }
    TAIL_TP(2)
    T l_nu = T(0), l_dfis = T(0), l_hrefv = T(0), l_dvis = T(0), l_dnu = T(0), l_dz = T(0), l_dw = T(0), l_dyis = T(0),
      l_av = T(0), l_prt = T(0), l_prs = T(0), l_up = T(0), l_lm = T(0);
    if (act && isj) {
      T fi[6], hl[21];
#pragma unroll
      for (int k = 0; k < 21; ++k) hl[k] = hcur[k];
      l_nu = tabs(nui);
      symv(hl, vi, fi);  // hxx:139-140
      T df[6], dv6[6], hrv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        fi[k] += p[k];
        df[k] = fi[k] - f[k];
        dv6[k] = vi[k] - v[k];
      }
      l_dfis = mass * inf6(df);  // (a massless chain link is not a body of the model: no f_i of its own upstream)
      href_mul_link<T, HDIAG>(P, hrow, vi, hrv);
      l_hrefv = mass * inf6(hrv);  // a massless chain link is not a body of the model
      l_dvis = mass * inf6(dv6);
      l_dnu = tabs(nui - nu);
      const T x = nui + (T(1) / mu_in) * w;
      const T zi = tmin(ubi, tmax(lbi, x));
      l_dz = tabs(zi - z);
      l_prs = tabs(nui - zi);
      const T dwi = mu_in * (nui - zi);
      l_dw = tabs(dwi);
      l_up = ubi * tmax(dwi, T(0));
      l_lm = lbi * tmin(dwi, T(0));
      w = w + dwi; z = zi; nu = nui;
#pragma unroll
      for (int k = 0; k < 6; ++k) { v[k] = vi[k]; f[k] = fi[k]; }
    }
    TAIL_TP(3)
    // DualUpdate of the task constraints (hxx:410-451), spread over six lanes of the group: lane k < 6 owns row k of
    // A v_c - b and of A^T y (the constrained joint's own lane used to do all 72 multiply-adds, i.e. the whole
    // wavefront paid for them).  v_c is still in the joint's exchange row from the forward recursion.
    for (int c = 0; c < L.nc; ++c) {
      T* c_ = cdi + c * CD;
      if (act && jlane < 6) {
        const int k = jlane;
        const T* vc = xch + (gbase + (int)c_[CD_PAD]) * XS + XC;
        T avk = c_[CD_A + 6 * k] * vc[0];
#pragma unroll
        for (int j = 1; j < 6; ++j) avk += c_[CD_A + 6 * k + j] * vc[j];
        const T bk = c_[CD_B + k];
        const T ek = avk - bk;
        const T dy = mu_eq * ek;
        const T yk = c_[CD_Y + k] + dy;
        l_dyis = tmax(l_dyis, tabs(dy));
        l_up += bk * tmax(dy, T(0));
        l_lm += bk * tmin(dy, T(0));
        l_prt = tmax(l_prt, tabs(ek));
        l_av = tmax(l_av, tabs(avk));
        c_[CD_Y + k] = yk;
      }
      tail_sync();
      if (act && jlane < 6) {
        const int k = jlane;
        T at = c_[CD_A + k] * c_[CD_Y];
#pragma unroll
        for (int j = 1; j < 6; ++j) at += c_[CD_A + 6 * j + k] * c_[CD_Y + j];
        c_[CD_ATY + k] = at;
      }
      tail_sync();
    }

    TAIL_TP(4)
    // ================= BwdPass2 + dual residual (hxx:185-241, :468-487) =========================================
    // g_i = Aty_c + sum_children act(f_j) - f_i needs the children's f only (no recursion): one exchange, no levels.
    T l_dg = T(0), l_g = T(0), l_dualv = T(0), l_stf = T(0), l_dstf = T(0);
    if (act && isj && has_parent) {
      T pc[6];
      act_force(R, t, f, pc);  // hxx:212
#pragma unroll
      for (int k = 0; k < 6; ++k) xch[lane * XS + k] = pc[k];
    }
    tail_sync();
    if (act && isj) {
      T gi[6];
      if (d.cslot >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) gi[k] = cdi[d.cslot * CD + CD_ATY + k];
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) gi[k] = T(0);
      }
      if (maxchild <= NCH_REG) {  // uniform; missing children read the zero row: no branch, no select
        gather_rows_n<T>(maxchild, xch, chl, 0, gi);
      } else {
        for (int c = 0; c < maxchild; ++c) {
          if (c < tp.nchild) {
            const int crow = c == 0 ? chl[0] : c == 1 ? chl[1] : c == 2 ? chl[2] : c == 3 ? chl[3]
                                                                     : gbase + child_list[tp.child_start + c];
            const T* x = xch + crow * XS;
#pragma unroll
            for (int k = 0; k < 6; ++k) gi[k] += x[k];
          }
        }
      }
      T dg[6], dvr[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        gi[k] += -f[k];
        dg[k] = gi[k] - g[k];
        g[k] = gi[k];
      }
      l_dg = inf6(dg);
      l_g = inf6(gi);
      href_mul_link<T, HDIAG>(P, hrow, v, dvr);
#pragma unroll
      for (int a = 0; a < 6; ++a) dvr[a] = mass * (dvr[a] - (hrow ? hrow[36 + a] : P.Hv[a])) + gi[a];
      l_dualv = inf6(dvr);
      T stf = rev ? (ax0 * f[3] + ax1 * f[4] + ax2 * f[5]) : (ax0 * f[0] + ax1 * f[1] + ax2 * f[2]);
      if (hel_any && hel) stf += ph * (ax0 * f[0] + ax1 * f[1] + ax2 * f[2]);
      const T si = stf + w;
      l_stf = tabs(si);
      l_dstf = tabs(si - s);
      s = si;
    }
    tail_sync();

    TAIL_TP(5)
    // ================= lane-group reductions of the running norms, then the scalar epilogue ======================
    // Through LDS: every lane deposits its NRED scalars in its exchange row, lane q of a group folds scalar q over the
    // group's rows (max for the inf-norms, sum for the two dot products) and publishes it in column XS-1 of row q.
    constexpr int NRED = 18, NRMAX = 16;
    {
      T* row = xch + lane * XS;
      row[0] = l_prt; row[1] = l_prs; row[2] = l_dualv; row[3] = l_stf; row[4] = l_dvis; row[5] = l_dnu;
      row[6] = l_dz; row[7] = l_dfis; row[8] = l_dyis; row[9] = l_dw; row[10] = l_av; row[11] = l_nu;
      row[12] = l_hrefv; row[13] = l_g; row[14] = l_dg; row[15] = l_dstf; row[16] = l_up; row[17] = l_lm;
    }
    tail_sync();
    for (int q = jlane; q < NRED; q += G) {
      // eight rows per trip (G is 8, 16, 32 or 64): the LDS reads of a trip are in flight together instead of one
      // read-wait-fold round trip per row; the sums keep the joint order of upstream's dot products (hxx:587-590)
      const T* col = xch + gbase * XS + q;
      T red = T(0);
      for (int l = 0; l < G; l += 8) {
        T a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = col[(l + u) * XS];
        if (q < NRMAX) {
          red = tmax(red, tmax(tmax(tmax(a[0], a[1]), tmax(a[2], a[3])), tmax(tmax(a[4], a[5]), tmax(a[6], a[7]))));
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) red += a[u];
        }
      }
      xch[(gbase + q % G) * XS + (XS - 1) - q / G] = red;
    }
    tail_sync();
    T rr[NRED];
#pragma unroll
    for (int q = 0; q < NRED; ++q) rr[q] = xch[(gbase + q % G) * XS + (XS - 1) - q / G];
    tail_sync();
    TAIL_TP(6)
    const T r_prt = rr[0], r_prs = rr[1], r_dualv = rr[2], r_stf = rr[3], r_dvis = rr[4], r_dnu = rr[5], r_dz = rr[6],
            r_dfis = rr[7], r_dyis = rr[8], r_dw = rr[9], r_av = rr[10], r_nu = rr[11], r_hrefv = rr[12], r_g = rr[13],
            r_dg = rr[14], r_dstf = rr[15], r_up = rr[16], r_lm = rr[17];
    if (act) {
      pr_task = r_prt; pr_slack = r_prs; dual_v = r_dualv; stf_w_inf = r_stf;
      primal = tmax(pr_task, pr_slack);
      dual = tmax(dual_v, stf_w_inf);
      n_dvis = r_dvis; n_dnu = r_dnu;
      dx = tmax(n_dvis, n_dnu);
      dz = r_dz;
      n_dfis = r_dfis; n_dyis = r_dyis; n_dw = r_dw; n_av = r_av; n_nu = r_nu; n_hrefv = r_hrefv; n_g = r_g;
      if (P.mode & MODE_FIXED_ITERS) {
        if (iter + 1 >= P.max_iter) { status |= ST_DONE; done = true; }
      } else if (!(status & ST_TAIL)) {
        tol_p = P.tol_abs + P.tol_rel * tmax(tmax(n_av, n_nu), tmax(bnorm, n_nu));
        tol_d = P.tol_abs + P.tol_rel * tmax(tmax(n_hrefv, tmax(n_g, stf_w_inf)), P.Hv_inf_norm);
        const bool conv = (primal < tol_p) && (dual < tol_d);
        bool infeas = false;
        if (iter > 1) {
          dyqp = tmax(n_dfis, tmax(n_dyis, n_dw));
          atdy = tmax(r_dg, r_dstf);
          c1 = atdy <= P.tol_primal_inf * dyqp;
          ubp = r_up;
          lbm = r_lm;
          c2 = (ubp + lbm) <= P.tol_primal_inf * dyqp;
          infeas = c1 && c2;
        }
        if (conv) {
          status |= ST_CONVERGED | ST_DONE;
          if (infeas) status |= ST_PRIMAL_INF;
          done = true;
        } else if (infeas) {
          status |= ST_PRIMAL_INF | ST_TAIL;
          tail_iter = 0;
          if (!(dx >= P.tol_tail_solve || dz >= P.tol_tail_solve) || iter >= P.max_iter) { status |= ST_DONE; done = true; }
        } else {
          if (update_mu<T>(P.mode, primal, dual, tmax(tmax(n_av, n_nu), bnorm),
                           tmax(tmax(n_hrefv, tmax(n_g, stf_w_inf)), P.Hv_inf_norm), mu, kexp))
            ++nflip;
          if (iter + 1 >= P.max_iter) { status |= ST_DONE; done = true; }
        }
      } else {
        tail_iter += 1;
        if (!(dx >= P.tol_tail_solve || dz >= P.tol_tail_solve) || iter >= P.max_iter) { status |= ST_DONE; done = true; }
      }
    }
  

    // a group whose instance just stopped stores it and takes the next one from the list
    if (done && has_inst) {
      store_instance();
      load_instance(fetch());
    }
    TAIL_TP(7)
  }
#ifdef LOIKB_TAIL_PROF
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int k = 0; k < 8; ++k) g_tail_prof[k] = prof_[k];
    g_tail_prof[8] = n_wave_iters;
  }
#endif
  // instances that were already finished when they were fetched (nothing to store), diagnostics
  if (lane == 0) {
    atomicAdd(&Bf.counters[5], n_wave_iters);
    atomicAdd(&Bf.counters[6], n_h_iters);
  }
}

// slot indices of the live instances of a set, dense, in slot order (same scan as k_move).
// (Serving the instances with the largest residual/tolerance ratio first -- it predicts the iterations still to come
//  with correlation 0.88 -- was measured: fewer wavefront-iterations (464 k vs 513 k per chunk) but a LONGER launch,
//  16.9 vs 15.8 ms: every wavefront then starts with two long runners and the short instances queue up behind them.
//  Other predictors of the remaining iterations (mu updates so far / recently, residual decay rate, stagnation) rank
//  even worse in a queue simulation on the real remaining-iteration counts; a first launch with a quantum of 100-400
//  iterations per instance followed by a run-out launch of the survivors: 35.1-35.8 vs 34.8 ms/step.)
template <typename T>
__global__ void __launch_bounds__(WAVE) k_list_live(char* tiles, Layout L, int n, const int* __restrict__ wave_off,
                                                    int* __restrict__ list)
{
  const int lane = threadIdx.x;
  const int b = blockIdx.x * WAVE + lane;
  const bool inb = b < n;
  char* sp = lane_ptr<T>(tiles, L, inb ? b : 0);
  const int status = inb ? (int)ldp<T>(sp + (size_t)L.off_s * pair_bytes<T>(), SP_ST).x : ST_DONE;
  const bool live = inb && !(status & ST_DONE);
  const unsigned long long mask = __ballot(live);
  if (live) list[wave_off[blockIdx.x] + __popcll(mask & ((1ull << lane) - 1ull))] = b;
}

// the identity list: every slot of a set (a batch small enough to go to the tail kernel from its first iteration)
#ifndef LOIKB_FLAT_KERNELS_TU   // (not a template: defined in the host translation unit only)
__global__ void k_list_iota(int* __restrict__ list, int n)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n) list[b] = b;
}
#endif

}  // namespace loikb
