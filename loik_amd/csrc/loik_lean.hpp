// loik_lean.hpp -- the tail kernel at TWO wavefronts per SIMD.
//
// k_tail (loik_tail.hpp) keeps an instance's whole ADMM state in registers/LDS, one joint per lane, and is bound by
// fp64 issue latency: one wavefront per SIMD (512 registers, 39 KB of LDS) executes its dependent chains along the tree
// levels at ~13 cycles per instruction while the SIMD could issue one every 4 (scripts/ubench/fp64_issue.hip: a second
// wavefront on the SIMD halves the time per instruction at this amount of instruction-level parallelism).
// k_lean is the same iteration at <= 256 registers and <= 20 KB of LDS per wavefront, so that two wavefronts share a
// SIMD.  What had to go:
//   * the H rebuild (the masked level loop that carries the 21 entries of H_i up the tree: ~150 registers on its own).
//     H_i / Dinv_i / UDinv_i depend on q and mu only, and mu only ever moves by decades (UpdateMu, hxx:613-641):
//     k_hslots below precomputes them for the decades mu0 * 10^(kexp_lo .. kexp_lo+ndec-1) into HBM "decade slots"
//     (indexed by the instance's slot in the tile set, 224 contiguous bytes per joint and decade) before the lean
//     launch; on a change of mu a lane fetches its 14 pairs.  An instance whose mu leaves the precomputed decades is
//     written back unfinished ("escapes") and is finished by k_tail.  An instance can also be sent back after a
//     bounded number of iterations (P.max_launch_iters: optional rounds, run_tail in loik_host.hip).
//   * the second H slot in LDS and the 22 exchange columns the rebuild needs (22.5 -> 10.5 KB, 15.6 -> 7.1 KB)
//   * the ~60 registers of per-instance scalars every lane carried redundantly (tolerances, the 14 norms kept for the
//     getters, flags): they live once per instance in LDS; every lane reads what the epilogue needs.
// Arithmetic per joint is k_tail's lean path; the only numerical difference is that a decade slot was built for
// mu0 * 10^k (repeated multiplication by 10) while the solver's mu may have reached the decade through x0.1 steps --
// a relative 1e-16 in H, far inside the parity tolerance.
#pragma once

#include "loik_tail.hpp"

namespace loikb {

constexpr int LXS = 14;  // scalars per exchange row: A = cols 0..5 (p messages up, act(f) for g), B = cols 6..11 (v down),
                         // 2 pad; the stride is 7 (odd) 16-byte slots -> conflict-free b128 rows; norms are deposited in
                         // cols 0..11 once both exchanges of the iteration are over
constexpr int LXA = 0, LXB = 6;
constexpr int LHS = 21;  // H_i per lane (Dinv_i stays in a register)
constexpr int HSLOT_PAIRS = 11;  // decade slot of a joint: H[21], Dinv (UDinv = H S Dinv is recomputed by the reader: 21 % fewer bytes)
// per-instance scalars in LDS (T each, integers included: they are small and exact)
enum : int { IS_MU = 0, IS_KEXP, IS_ITER, IS_STATUS, IS_TAILIT, IS_C1, IS_C2, IS_NFLIP, IS_TOLP, IS_TOLD, IS_DYQP, IS_ATDY,
             IS_UBP, IS_LBM, IS_BNORM, IS_PRIMAL, IS_DUAL, IS_PRT, IS_PRS, IS_DUALV, IS_STF, IS_DX, IS_DZ, IS_DFIS,
             IS_DYIS, IS_DW, IS_DVIS, IS_DNU, IS_AV, IS_NU, IS_HREFV, IS_G, IS_TGIN, IS_STY, IS_MULAST,
             IS_RED /* results of the folds: 12 */, ISC = IS_RED + 12 };

// constraint block of an instance in LDS: lane of the constrained joint, b, A^T b, y, A^T y (+ A when it is per instance;
// a shared A is kept once per wavefront).  AtA is not needed: H is never rebuilt here.
constexpr int LCD = 26, LCA = 36;
// counters of a lean launch (Bufs::counters): [1] instance-iterations, [2] escaped instances, [5] wavefront-iterations,
// [6] decade-slot loads, and the work queue:
constexpr int LEAN_Q_HEAD = 7, LEAN_Q_TAIL = 8, LEAN_Q_REQUEUES = 9, LEAN_Q_RETIRED = 10;
constexpr int LEAN_DECADES_SEEN = 11;  // bit d: some instance loaded the slot of decade kexp_lo + d in this launch
constexpr int NCOUNTERS = 96;   // (32..63: k_flat2's per-decade slot counts, FLAT_COUNTERS_DEC; 64..95: the decades the instances ENDED in, ORDER_DEC_HIST)
constexpr int ORDER_DEC_HIST = 64;
#ifndef LOIKB_POLL_MASK
#define LOIKB_POLL_MASK 3u
#endif
enum : int { LC_PAD = 0, LC_B = 1, LC_ATB = 7, LC_Y = 13, LC_ATY = 19 };

// ---- accesses to an instance's record that are coherent across the chip WITHOUT fences.  The lean kernel's work queue lets
// an instance migrate between lane groups -- i.e. between CUs of different XCDs, each with its own L2 -- within ONE launch.
// An agent-scope release/acquire pair would make that safe, but on gfx950 it costs a write-back + invalidate of a whole
// 4 MB L2 (measured: 20 us per instance switch, 60x the kernel's run time).  Instead every load/store of the mutable part
// of a record is an agent-scope relaxed atomic: the access itself carries the scope bits (sc1: stores write through, loads
// do not hit possibly stale L2/L1 lines), nothing else is flushed.  Ordering between the record and the queue entry that
// publishes it is by completion: the pusher waits for its stores (s_waitcnt vmcnt(0)) before it writes the ring entry,
// the popper issues its loads after it has read the entry.
template <typename T>
#ifdef LOIKB_PLAIN_RECORDS
__device__ __forceinline__ T cld(const char* p) { return *reinterpret_cast<const T*>(p); }
#else
__device__ __forceinline__ T cld(const char* p) { return __hip_atomic_load(reinterpret_cast<const T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
template <typename T>
#ifdef LOIKB_PLAIN_RECORDS
__device__ __forceinline__ void cst(char* p, T v) { *reinterpret_cast<T*>(p) = v; }
#else
__device__ __forceinline__ void cst(char* p, T v) { __hip_atomic_store(reinterpret_cast<T*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
template <typename T>
__device__ __forceinline__ typename Vec2<T>::type cldp(const char* rec, int p)
{
  typename Vec2<T>::type v;
  v.x = cld<T>(rec + (size_t)p * pair_bytes<T>());
  v.y = cld<T>(rec + (size_t)p * pair_bytes<T>() + sizeof(T));
  return v;
}
template <typename T>
__device__ __forceinline__ void cstp(char* rec, int p, T x, T y)
{
  cst<T>(rec + (size_t)p * pair_bytes<T>(), x);
  cst<T>(rec + (size_t)p * pair_bytes<T>() + sizeof(T), y);
}
template <typename T>
__device__ __forceinline__ void cld6(const char* rec, int p, T* x)
{
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const typename Vec2<T>::type v = cldp<T>(rec, p + k);
    x[2 * k] = v.x; x[2 * k + 1] = v.y;
  }
}
template <typename T>
__device__ __forceinline__ void cst6(char* rec, int p, const T* x)
{
#pragma unroll
  for (int k = 0; k < 3; ++k) cstp<T>(rec, p + k, x[2 * k], x[2 * k + 1]);
}
template <typename T>
__device__ __forceinline__ T cld_scal(const char* srec, int idx)
{
  return cld<T>(srec + (size_t)(SP_SCAL + idx / 2) * pair_bytes<T>() + (idx & 1) * sizeof(T));
}

// record accessors of the lean kernel: coherent when instances migrate between lane groups (SLICED), plain otherwise
template <typename T, bool COH>
__device__ __forceinline__ typename Vec2<T>::type rldp(const char* rec, int p) { if constexpr (COH) return cldp<T>(rec, p); else return ldp<T>(rec, p); }
template <typename T, bool COH>
__device__ __forceinline__ void rstp(char* rec, int p, T x, T y) { if constexpr (COH) cstp<T>(rec, p, x, y); else stp<T>(rec, p, x, y); }
template <typename T, bool COH>
__device__ __forceinline__ void rld6(const char* rec, int p, T* x) { if constexpr (COH) cld6<T>(rec, p, x); else ld6<T>(rec, p, x); }
template <typename T, bool COH>
__device__ __forceinline__ void rst6(char* rec, int p, const T* x) { if constexpr (COH) cst6<T>(rec, p, x); else st6<T>(rec, p, x); }
template <typename T, bool COH>
__device__ __forceinline__ T rld_scal(const char* srec, int idx) { if constexpr (COH) return cld_scal<T>(srec, idx); else return ld_scal<T>(srec, idx); }
template <typename T, bool COH>
__device__ __forceinline__ T rld(const char* p) { if constexpr (COH) return cld<T>(p); else return *reinterpret_cast<const T*>(p); }
template <typename T, bool COH>
__device__ __forceinline__ void rst(char* p, T v) { if constexpr (COH) cst<T>(p, v); else *reinterpret_cast<T*>(p) = v; }

template <typename T>
__host__ __device__ __forceinline__ size_t lean_lds_bytes(int nc, int G, bool a_shared)
{
  const size_t per_inst = (size_t)nc * (LCD + (a_shared ? 0 : LCA)) + ISC;
  return ((((size_t)XROWS * LXS + (size_t)WAVE * LHS + (a_shared ? (size_t)nc * LCA : 0) + (size_t)(WAVE / G) * per_inst) *
           sizeof(T)) + 15) & ~(size_t)15;
}

// pair k of lane j of the decade slot (instance slot idx, decade d).  The 11 pairs of a joint are contiguous (176 B):
// the builder writes a joint's slot from the few lanes that sit at one tree level at a time -- with the pairs of
// different joints interleaved every store touched a quarter of a 64-byte line (4.2 ms for the headline's table,
// write-bound); whole lines per lane bring it to the cost of the arithmetic.
__device__ __forceinline__ size_t hslot_pair(int idx, int ndec, int dsl, int G, int k, int jlane)
{
  return (((size_t)idx * ndec + dsl) * G + jlane) * HSLOT_PAIRS + k;
}

// fold 12 columns of the group's exchange rows (columns < nmax: max, the others: sum in lane order) into isc[IS_RED..]
template <typename T>
__device__ __forceinline__ void lean_fold(const T* xch, T* isc, int gbase, int jlane, int G, int ncol, int nmax)
{
  for (int q = jlane; q < ncol; q += G) {
    const T* col = xch + gbase * LXS + q;
    T red = T(0);
    for (int l = 0; l < G; l += 8) {
      T a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = col[(l + u) * LXS];
      if (q < nmax) {
        red = tmax(red, tmax(tmax(tmax(a[0], a[1]), tmax(a[2], a[3])), tmax(tmax(a[4], a[5]), tmax(a[6], a[7]))));
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) red += a[u];
      }
    }
    isc[IS_RED + q] = red;
  }
}

// acc += the 6-vectors at column `off` of the exchange rows chl[0..n): see gather_rows in loik_tail.hpp (row stride LXS)
// (one child at a time: here registers are scarce and the LDS latency is covered by the SIMD's other wavefront)
template <typename T, int NCH>
__device__ __forceinline__ void lean_gather(const T* xch, const int* chl, int off, T* acc)
{
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const T* x = xch + chl[c] * LXS + off;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] += x[k];
  }
}
template <typename T>
__device__ __forceinline__ void lean_gather_n(int n, const T* xch, const int* chl, int off, T* acc)
{
  switch (n) {  // uniform
  case 0: break;
  case 1: lean_gather<T, 1>(xch, chl, off, acc); break;
  case 2: lean_gather<T, 2>(xch, chl, off, acc); break;
  case 3: lean_gather<T, 3>(xch, chl, off, acc); break;
  default: lean_gather<T, 4>(xch, chl, off, acc); break;
  }
}

// PERLINK: the links' (H_ref_i, H_ref_i v_ref_i) come from the table UpdateReferences left in HBM (Params::href_tab, one row per
// joint: 10.5 KiB for Talos, read through L1 / L2 -- 42 loads per joint and iteration) instead of the one pair every link
// shares.  An instantiation of its own: as a uniform branch in the shared-reference kernel it cost the headline 2-3 %.
template <typename T, bool HDIAG, bool SLICED, bool PERLINK = false>
__global__ void __launch_bounds__(WAVE * TAIL_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_lean(const Params<T> P, const Bufs<T> Bf, const JointDesc* __restrict__ jd, const TailTopo* __restrict__ topo,
       const int* __restrict__ child_list, int maxdepth, int maxchild, int* __restrict__ ring, int ring_mask, int nslots, int G,
       const T* __restrict__ hslots, int kexp_lo, int ndec, int quantum, int multi_from)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Layout& L = P.L;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool a_shared = P.mode & MODE_A_SHARED;
  const int cs = LCD + (a_shared ? 0 : LCA);                // scalars per constraint block of an instance
  const size_t wave_lds = lean_lds_bytes<T>(L.nc, G, a_shared);
  T* xch = reinterpret_cast<T*>(smem_raw + wv * wave_lds);  // [WAVE + 1][LXS]
  T* hst = xch + XROWS * LXS;                               // [WAVE][LHS]
  T* ash = hst + WAVE * LHS;                                // [nc][36]        the shared A (if it is shared)
  T* cd = ash + (a_shared ? L.nc * LCA : 0);                // [64/G][nc][cs]
  T* iscb = cd + (size_t)(WAVE / G) * L.nc * cs;            // [64/G][ISC]
  const int lane = threadIdx.x & (WAVE - 1);
  const int sub = lane / G, jlane = lane % G, gbase = sub * G;
  const bool isj_lane = jlane < L.nb;
  const int jl = isj_lane ? jlane : 0;
  T* cdi = cd + (size_t)sub * L.nc * cs;
  T* isc = iscb + (size_t)sub * ISC;
  T* hcur = hst + (size_t)lane * LHS;

  const JointDesc d = jd[jl + 1];
  const TailTopo tp = topo[jl + 1];
  const bool rev = d.flags & JF_REVOLUTE;
  const T mass = (d.flags & JF_MASSLESS) ? T(0) : T(1);
  static_assert(!(PERLINK && HDIAG), "per-link references are general 6x6 blocks");
  const T* const hrow = PERLINK ? P.href_tab + (size_t)(jl + 1) * HREF_ROW : nullptr;
  auto hv_ref = [&](int k) -> T { return PERLINK ? hrow[36 + k] : P.Hv[k]; };      // (H_ref_i v_ref_i)_k
  auto href_times = [&](const T* x, T* o) {                                        // o = H_ref_i x
    if (PERLINK) href_mul<T, false>(hrow, x, o);
    else href_mul<T, HDIAG>(P.Href, x, o);
  };
  const bool has_parent = !(d.flags & JF_PARENT_ROOT);
  constexpr int NCH_REG = 4;
  int chl[NCH_REG];
#pragma unroll
  for (int c = 0; c < NCH_REG; ++c) chl[c] = c < tp.nchild ? gbase + child_list[tp.child_start + c] : WAVE;
  const int prow = has_parent ? gbase + d.parent - 1 : WAVE;
  T Sv[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) { Sv[k] = rev ? T(0) : (T)d.axis[k]; Sv[3 + k] = rev ? (T)d.axis[k] : T(0); }
  if (d.flags & JF_HELICAL) {  // S = [pitch a; a]: every use of S in this kernel is the 6-vector
#pragma unroll
    for (int k = 0; k < 3; ++k) Sv[k] = (T)d.pitch * (T)d.axis[k];
  }
  if (lane < LXS) xch[WAVE * LXS + lane] = T(0);
  if (a_shared)
    for (int e = lane; e < L.nc * LCA; e += WAVE) ash[e] = Bf.uni[e];

  bool has_inst = false, isj = false, done = true, any_iter = false;
  int lidx = 0;  // the instance's slot in the set: also the index of its decade slots
  char *ip = Bf.tiles, *rec = Bf.tiles;
  T R[9], t[3], v[6], f[6], g[6], UD[6], p[6];
  T w = T(0), z = T(0), nu = T(0), s = T(0), r = T(0), dinv = T(0), lbi = T(0), ubi = T(0), mu = T(1);
  int kexp = 0, kslot = -(1 << 30);
  unsigned int my_iters = 0, slice_iters = 0;  // (slice_iters, requeue, ticket, fin: SLICED only)
  unsigned int n_wave_iters = 0, n_slot_loads = 0;
  bool requeue = false;

  // ---- work queue: a ring of instance slots in HBM, tickets on both sides (LEAN_Q_HEAD = tickets handed to groups that want
  // an instance, LEAN_Q_TAIL = entries pushed; ticket t is served by entry t -- one atomicAdd per pop or push, no CAS loop:
  // thousands of groups pop at the same moment at the start and at the first time-slice boundaries).  The ring starts out
  // holding every listed instance; a lane group takes one, iterates it for at most `quantum` iterations and, if other
  // instances are still waiting for a slot (tail > head), writes it back and pushes it to the BACK of the ring: round-robin
  // time slicing.  Iteration counts are heavy-tailed (median 26, 1.2 % run all 1000) and unpredictable, and a
  // 1000-iteration instance is a serial chain of ~10 ms -- run to completion in arrival order such instances are fetched
  // late and the launch ends ~9 ms after the queue ran dry; time-sliced, every long runner advances from the start and
  // the machine stays full until shortly before the end (scripts/r02/sim_sched.py: 22.1 -> 14.1 ms on the headline).
  // An instance whose quantum expires while nothing waits simply continues.  A group holding a ticket beyond the tail
  // polls until its entry appears or every listed instance has retired (LEAN_Q_RETIRED == nslots): finished, or written
  // back because its mu left the decade slots.
  unsigned int* q_head = Bf.counters + LEAN_Q_HEAD;
  unsigned int* q_tail = Bf.counters + LEAN_Q_TAIL;
  unsigned int* q_retired = Bf.counters + LEAN_Q_RETIRED;
  unsigned int ticket = 0;
  bool fin = false;  // this group will get no more work
  auto q_waiting = [&]() -> bool {
    int wtg = 0;
    if (jlane == 0)
      wtg = (int)(__hip_atomic_load(q_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                  __hip_atomic_load(q_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) > 0;
    return __shfl(wtg, gbase) != 0;
  };
  auto q_ticket = [&]() {
    unsigned int tk = 0;
    if (jlane == 0) tk = atomicAdd(q_head, 1u);
    ticket = (unsigned int)__shfl((int)tk, gbase);
  };
  auto q_poll = [&]() -> int {  // the instance slot of this group's ticket; -1: not there yet; -2: nothing will come any more
    int got = -1;
    if (jlane == 0) {
      int* e = ring + (ticket & (unsigned int)ring_mask);
      const int v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= 0) {
        __hip_atomic_store(e, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        got = v;
      } else if (__hip_atomic_load(q_retired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned int)nslots) {
        got = -2;
      }
    }
    return __shfl(got, gbase);
  };
  auto q_push = [&](int slot) {  // (after store_instance: every store of the record has completed before the entry appears)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): stores count in vmcnt on gfx9
    __builtin_amdgcn_wave_barrier();
    if (jlane == 0) {
      const unsigned int pos = atomicAdd(q_tail, 1u);
      int* e = ring + (pos & (unsigned int)ring_mask);
      // (the ring has more entries than instances + groups: the entry of the previous lap was consumed long ago)
      while (__hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(e, slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto load_instance = [&](int slot_in) {
    has_inst = slot_in >= 0;
    isj = has_inst && isj_lane;
    const int slot = has_inst ? slot_in : 0;
    lidx = slot;
    ip = lane_ptr<T>(Bf.tiles, L, slot);
    rec = ip + (size_t)jl * JREC * pair_bytes<T>();
    const char* srec = ip + (size_t)L.off_s * pair_bytes<T>();
    {
      const typename Vec2<T>::type cs = ldp<T>(rec, JP_CS), wz = rldp<T, SLICED>(rec, JP_WZ), nus = rldp<T, SLICED>(rec, JP_NUS);
      joint_xform<T>(d, rec, cs.x, cs.y, R, t);
      rld6<T, SLICED>(rec, JP_V, v);
      rld6<T, SLICED>(rec, JP_F, f);
      rld6<T, SLICED>(rec, JP_G, g);
      w = wz.x; z = wz.y; nu = nus.x; s = nus.y;
      if (P.mode & MODE_BND_SHARED) {
        lbi = Bf.uni[L.nc * 57 + jl];
        ubi = Bf.uni[L.nc * 57 + L.nb + jl];
      } else {
        const typename Vec2<T>::type lu = ldp<T>(rec, JP_LBUB);
        lbi = lu.x; ubi = lu.y;
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) { UD[k] = T(0); p[k] = T(0); }
      r = T(0); dinv = T(0);
    }
    for (int c = 0; c < L.nc; ++c) {
      const char* crec = ip + (size_t)(L.off_c + c * L.crec) * pair_bytes<T>();
      for (int e = jlane + 1; e < cs; e += G) {  // (entry 0: the lane of the constrained joint, set below)
        T val;
        if (e < LCD) {
          const int q = e - 1;
          if (q >= 24) continue;  // pad
          const int which = q / 6, k = q % 6;
          const int pair = which == 0 ? CP_B : which == 1 ? CP_ATB : which == 2 ? CP_Y : CP_ATY;
          val = rld<T, SLICED>(crec + (size_t)(pair + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T));  // (y, A^T y change in the kernel)
        } else {
          const int q = e - LCD;
          val = *reinterpret_cast<const T*>(crec + (size_t)(CP_A + q / 2) * pair_bytes<T>() + (q & 1) * sizeof(T));
        }
        cdi[c * cs + e] = val;
      }
    }
    if (isj_lane && d.cslot >= 0) cdi[d.cslot * cs + LC_PAD] = (T)jlane;
    const typename Vec2<T>::type mu2 = rldp<T, SLICED>(srec, SP_MU), bi2 = rldp<T, SLICED>(srec, SP_BI), st2 = rldp<T, SLICED>(srec, SP_ST);
    mu = mu2.x;
    kexp = (int)mu2.y;
    kslot = -(1 << 30);
    int status = has_inst ? (int)st2.x : ST_DONE;
    const int iter = (int)bi2.y;
    done = (status & ST_DONE) != 0;
    if (!done && !(status & ST_TAIL) && iter + 1 >= P.max_iter) { done = true; status |= ST_DONE; }
    if (jlane == 0) {
      isc[IS_MU] = mu; isc[IS_KEXP] = (T)kexp; isc[IS_ITER] = (T)iter; isc[IS_STATUS] = (T)status;
      isc[IS_TAILIT] = rld_scal<T, SLICED>(srec, SC_TAIL_ITER);
      isc[IS_C1] = rld_scal<T, SLICED>(srec, SC_COND1); isc[IS_C2] = rld_scal<T, SLICED>(srec, SC_COND2);
      isc[IS_NFLIP] = rldp<T, SLICED>(srec, SP_FLIP).x;
      isc[IS_TOLP] = rld_scal<T, SLICED>(srec, SC_TOL_PRIMAL); isc[IS_TOLD] = rld_scal<T, SLICED>(srec, SC_TOL_DUAL);
      isc[IS_DYQP] = rld_scal<T, SLICED>(srec, SC_DELTA_Y_QP); isc[IS_ATDY] = rld_scal<T, SLICED>(srec, SC_AT_DELTA_Y_QP);
      isc[IS_UBP] = rld_scal<T, SLICED>(srec, SC_UB_DY_PLUS); isc[IS_LBM] = rld_scal<T, SLICED>(srec, SC_LB_DY_MINUS);
      isc[IS_BNORM] = bi2.x;
      isc[IS_TGIN] = rldp<T, SLICED>(srec, SP_TAG).x; isc[IS_STY] = st2.y; isc[IS_MULAST] = T(-1);
    }
    tail_sync();
    my_iters = 0;
    slice_iters = 0;
    any_iter = false;
  };
  auto store_instance = [&]() {
    char* srec = ip + (size_t)L.off_s * pair_bytes<T>();
    if (isj) {
      rst6<T, SLICED>(rec, JP_V, v);
      rst6<T, SLICED>(rec, JP_F, f);
      rst6<T, SLICED>(rec, JP_G, g);
      rstp<T, SLICED>(rec, JP_WZ, w, z);
      rstp<T, SLICED>(rec, JP_NUS, nu, s);
      if (any_iter) {
        rst6<T, SLICED>(rec, JP_P, p);
        rst6<T, SLICED>(rec, JP_UD, UD);
        rstp<T, SLICED>(rec, JP_R, r, dinv);
      }
    }
    tail_sync();
    if (has_inst) {
      for (int c = 0; c < L.nc; ++c) {
        char* crec = ip + (size_t)(L.off_c + c * L.crec) * pair_bytes<T>();
        if (jlane < 6) {
          const int k = jlane;
          rst<T, SLICED>(crec + (size_t)(CP_Y + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T), cdi[c * cs + LC_Y + k]);
          rst<T, SLICED>(crec + (size_t)(CP_ATY + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T), cdi[c * cs + LC_ATY + k]);
        }
      }
      if (jlane == 0) {
        const T mu_s = isc[IS_MU], mu_last = isc[IS_MULAST];
        const int status = (int)isc[IS_STATUS];
        rstp<T, SLICED>(srec, SP_MU, mu_s, isc[IS_KEXP]);
        rstp<T, SLICED>(srec, SP_TAG, any_iter ? mu_last : isc[IS_TGIN], T(0));
        rstp<T, SLICED>(srec, SP_BI, isc[IS_BNORM], isc[IS_ITER]);
        rstp<T, SLICED>(srec, SP_FLIP, isc[IS_NFLIP], T(0));
        rstp<T, SLICED>(srec, SP_ST, (T)(any_iter ? (status | ST_PFULL) : status), any_iter ? mu_last : isc[IS_STY]);
        if (any_iter) {
          rstp<T, SLICED>(srec, SP_SCAL + 0, isc[IS_PRIMAL], isc[IS_DUAL]);
          rstp<T, SLICED>(srec, SP_SCAL + 1, isc[IS_PRT], isc[IS_PRS]);
          rstp<T, SLICED>(srec, SP_SCAL + 2, isc[IS_DUALV], isc[IS_STF]);
          rstp<T, SLICED>(srec, SP_SCAL + 3, isc[IS_TOLP], isc[IS_TOLD]);
          rstp<T, SLICED>(srec, SP_SCAL + 4, mu_s, P.mu_scale * mu_s);
          rstp<T, SLICED>(srec, SP_SCAL + 5, mu_s, isc[IS_DX]);
          rstp<T, SLICED>(srec, SP_SCAL + 6, isc[IS_DZ], isc[IS_DYQP]);
          rstp<T, SLICED>(srec, SP_SCAL + 7, isc[IS_ATDY], isc[IS_UBP]);
          rstp<T, SLICED>(srec, SP_SCAL + 8, isc[IS_LBM], isc[IS_DFIS]);
          rstp<T, SLICED>(srec, SP_SCAL + 9, isc[IS_DYIS], isc[IS_DW]);
          rstp<T, SLICED>(srec, SP_SCAL + 10, isc[IS_DVIS], isc[IS_DNU]);
          rstp<T, SLICED>(srec, SP_SCAL + 11, isc[IS_AV], isc[IS_NU]);
          rstp<T, SLICED>(srec, SP_SCAL + 12, isc[IS_HREFV], isc[IS_G]);
          rstp<T, SLICED>(srec, SP_SCAL + 13, isc[IS_STF], isc[IS_C1]);
          rstp<T, SLICED>(srec, SP_SCAL + 14, isc[IS_C2], isc[IS_TAILIT]);
        }
        if (my_iters) atomicAdd(&Bf.counters[1], my_iters);
      }
    }
    tail_sync();
  };

  // SLICED = false: the plain work list -- ring[0 .. nslots) handed out once by one atomic counter, every instance runs to
  // completion in the group that fetched it (no migration: plain loads/stores of the records)
  auto fetch_plain = [&]() -> int {
    int nx = 0;
    if (jlane == 0) nx = (int)atomicAdd(q_head, 1u);
    nx = __shfl(nx, gbase);
    return nx < nslots ? ring[nx] : -1;
  };
  if constexpr (SLICED) {
    q_ticket();  // (has_inst = false, done = true: the group starts out idle, waiting for the entry of its first ticket)
  } else {
    load_instance(fetch_plain());
  }
#ifdef LOIKB_TAIL_PROF
  unsigned long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = clock64();
  const unsigned long long wall0_ = wall_clock64(), clk0_ = tprev_;
#endif
  unsigned int poll_skip = 0;
#ifdef LOIKB_TAIL_PROF
  unsigned long long dbg_last_ = wall0_, dbg_both_ = 0, dbg_sw_ = 0;
#endif
  bool want_poll = true;
  while (SLICED ? true : __any(!done || has_inst)) {
    if constexpr (SLICED) {
      // a group without an instance looks for the entry of its ticket: right after it took the ticket (the ring starts out
      // full: the first ticket of every group is served at once), then every 4th wavefront-iteration while the wavefront's
      // other group keeps iterating (a poll is a global round trip), or after a nap when the whole wavefront is idle
      if (!has_inst && !fin && want_poll) {
        const int got = q_poll();
        if (got >= 0) load_instance(got);  // (the record's loads are issued after the entry was read, and are coherent)
        else fin = got == -2;
      }
      if (!__any(has_inst)) {
        if (__all(fin)) break;
        __builtin_amdgcn_s_sleep(32);
        want_poll = true;
        continue;
      }
    }
    // ---- decade slot of the current mu (H_i, Dinv_i, UDinv_i) -------------------------------------------------------
    if (!done && (int)my_iters >= P.max_launch_iters) {
      done = true;  // this launch's share of iterations is used up: back to the list, the host relaunches (run_tail)
    }
    if (SLICED && quantum > 0 && __any(!done && (int)slice_iters >= quantum)) {
      const bool expired = !done && (int)slice_iters >= quantum;
      const bool waiting = q_waiting();
      if (expired) {
        if (waiting) { done = true; requeue = true; }  // time slice used up and others are waiting: to the back of the ring
        else slice_iters = 0;                           // nobody is waiting for this slot: carry on
      }
    }
    if (!done && kexp != kslot) {
      const int dsl = kexp - kexp_lo;
      if (dsl < 0 || dsl >= ndec) {
        done = true;  // mu left the precomputed decades: written back unfinished, k_tail takes over
        if (jlane == 0) atomicAdd(&Bf.counters[2], 1u);
      } else {
        if (isj) {
          const typename Vec2<T>::type* hp = reinterpret_cast<const typename Vec2<T>::type*>(hslots);
#ifndef LOIKB_SLOT_TWO_PARTS
          {
            typename Vec2<T>::type in[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) in[k] = hp[hslot_pair(lidx, ndec, dsl, G, k, jlane)];
#pragma unroll
            for (int k = 0; k < 10; ++k) { hcur[2 * k] = in[k].x; hcur[2 * k + 1] = in[k].y; }
            hcur[20] = in[10].x;
            dinv = in[10].y;
          }
#else  // (the round-1 form, kept for A/B builds: two dependent round trips)
          {
            typename Vec2<T>::type in[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) in[k] = hp[hslot_pair(lidx, ndec, dsl, G, k, jlane)];
#pragma unroll
            for (int k = 0; k < 6; ++k) { hcur[2 * k] = in[k].x; hcur[2 * k + 1] = in[k].y; }
          }
          {
            typename Vec2<T>::type in[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) in[k] = hp[hslot_pair(lidx, ndec, dsl, G, 6 + k, jlane)];
#pragma unroll
            for (int k = 0; k < 4; ++k) { hcur[12 + 2 * k] = in[k].x; hcur[13 + 2 * k] = in[k].y; }
            hcur[20] = in[4].x;
            dinv = in[4].y;
          }
#endif
          // UDinv = (H S) Dinv  (calc_aba, hxx:60-63) from the slot just written to LDS
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            T u = T(0);
#pragma unroll
            for (int j = 0; j < 6; ++j) u += hcur[sym(k, j)] * Sv[j];
            UD[k] = u * dinv;
          }
        }
        kslot = kexp;
        n_slot_loads = (n_slot_loads + 0x10000u) | (1u << dsl);  // count in the upper half, the decades visited in the lower
      }
    }
    TAIL_TP(8)
    const bool act = !done;
#ifdef LOIKB_TAIL_PROF
    dbg_last_ = wall_clock64();
    if (__popcll(__ballot(act && jlane == 0)) * G == WAVE) ++dbg_both_;
#endif
    const T mu_eq = P.mu_scale * mu, mu_in = mu;
    if (act) { ++my_iters; ++slice_iters; any_iter = true; }
    ++n_wave_iters;

    // ================= leaf -> root: FwdPass1 + BwdPass, p only (hxx:290-338, :31-81) =================================
    if (act) {
#pragma unroll
      for (int k = 0; k < 6; ++k) p[k] = mass * (-P.rho * v[k] - hv_ref(k));
      if (isj && d.cslot >= 0) {
        const T* c_ = cdi + d.cslot * cs;
#pragma unroll
        for (int k = 0; k < 6; ++k) p[k] += c_[LC_ATY + k] - mu_eq * c_[LC_ATB + k];
      }
    }
    TAIL_TP(0)
    {
      T pl[6], rl = T(0);
#pragma unroll
      for (int k = 0; k < 6; ++k) pl[k] = p[k];
      for (int lev = maxdepth; lev >= 1; --lev) {
#pragma unroll
        for (int k = 0; k < 6; ++k) pl[k] = p[k];
        // (a joint with several children is final only from step `multi_from` on -- the smallest height of such a joint,
        //  loik_host.hip: before that every lane that matters has at most its first child, and the further child rows
        //  are not read: 18 instructions and three 16-byte LDS reads per skipped child, level and wavefront)
        lean_gather_n<T>((maxdepth - lev + 1 >= multi_from) ? maxchild : (maxchild > 0 ? 1 : 0), xch, chl, LXA, pl);
        const T Stp = dot6_halves(Sv, pl);
        rl = (w - mu_in * z) + Stp;
        T pa[6], pc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) pa[k] = pl[k] - UD[k] * rl;
        act_force(R, t, pa, pc);
        tail_sync();
#pragma unroll
        for (int k = 0; k < 6; ++k) xch[lane * LXS + LXA + k] = pc[k];
        tail_sync();
      }
      if (act) {
#pragma unroll
        for (int k = 0; k < 6; ++k) p[k] = pl[k];
        r = rl;
      }
    }

    TAIL_TP(1)
    // ================= root -> leaf: FwdPass2 (hxx:102-163) ==============================================================
    T vi[6], nui = T(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) vi[k] = T(0);
    for (int lev = 1; lev <= maxdepth; ++lev) {
      T vpar[6], vp[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) vpar[k] = xch[prow * LXS + LXB + k];
      actinv_motion(R, t, vpar, vp);
      const T udv = dot6_halves(UD, vp);
      nui = -udv - dinv * r;
#pragma unroll
      for (int k = 0; k < 6; ++k) vi[k] = vp[k] + Sv[k] * nui;
      tail_sync();
#pragma unroll
      for (int k = 0; k < 6; ++k) xch[lane * LXS + LXB + k] = vi[k];
      tail_sync();
    }
    TAIL_TP(2)
    // per-lane norms of this iteration (folded over the group below)
    T l_nu = T(0), l_dfis = T(0), l_hrefv = T(0), l_dvis = T(0), l_dnu = T(0), l_dz = T(0), l_dw = T(0), l_dyis = T(0),
      l_av = T(0), l_prt = T(0), l_prs = T(0), l_up = T(0), l_lm = T(0);
    if (act && isj) {
      T fi[6];
      l_nu = tabs(nui);
      // f = H v straight from the LDS slot, entry by entry (no 21-entry copy in registers)
#pragma unroll
      for (int k = 0; k < 6; ++k) fi[k] = T(0);
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b2 = a; b2 < 6; ++b2) {
          const T h = hcur[sym(a, b2)];
          fi[a] += h * vi[b2];
          if (a != b2) fi[b2] += h * vi[a];
        }
      T df[6], dv6[6], hrv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        fi[k] += p[k];
        df[k] = fi[k] - f[k];
        dv6[k] = vi[k] - v[k];
      }
      l_dfis = mass * inf6(df);  // (a massless chain link is not a body of the model: no f_i of its own upstream)
      href_times(vi, hrv);
      l_hrefv = mass * inf6(hrv);
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
    // DualUpdate of the task constraints (hxx:410-451), six lanes of the group
    for (int c = 0; c < L.nc; ++c) {
      T* c_ = cdi + c * cs;
      const T* A_ = a_shared ? ash + c * LCA : c_ + LCD;
      if (act && jlane < 6) {
        const int k = jlane;
        const T* vc = xch + (gbase + (int)c_[LC_PAD]) * LXS + LXB;
        T avk = A_[6 * k] * vc[0];
#pragma unroll
        for (int j = 1; j < 6; ++j) avk += A_[6 * k + j] * vc[j];
        const T bk = c_[LC_B + k];
        const T ek = avk - bk;
        const T dy = mu_eq * ek;
        const T yk = c_[LC_Y + k] + dy;
        l_dyis = tmax(l_dyis, tabs(dy));
        l_up += bk * tmax(dy, T(0));
        l_lm += bk * tmin(dy, T(0));
        l_prt = tmax(l_prt, tabs(ek));
        l_av = tmax(l_av, tabs(avk));
        c_[LC_Y + k] = yk;
      }
      tail_sync();
      if (act && jlane < 6) {
        const int k = jlane;
        T at = A_[k] * c_[LC_Y];
#pragma unroll
        for (int j = 1; j < 6; ++j) at += A_[6 * j + k] * c_[LC_Y + j];
        c_[LC_ATY + k] = at;
      }
      tail_sync();
    }

    TAIL_TP(4)
    // ================= BwdPass2 + dual residual (hxx:185-241, :468-487) =================================================
    T l_dg = T(0), l_g = T(0), l_dualv = T(0), l_stf = T(0), l_dstf = T(0);
    {
      T pc[6];
      act_force(R, t, f, pc);
#pragma unroll
      for (int k = 0; k < 6; ++k) xch[lane * LXS + LXA + k] = (act && isj && has_parent) ? pc[k] : T(0);
    }
    tail_sync();
    if (act && isj) {
      T gi[6];
      if (d.cslot >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) gi[k] = cdi[d.cslot * cs + LC_ATY + k];
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) gi[k] = T(0);
      }
      lean_gather_n<T>(maxchild, xch, chl, LXA, gi);
      T dg[6], dvr[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        gi[k] += -f[k];
        dg[k] = gi[k] - g[k];
        g[k] = gi[k];
      }
      l_dg = inf6(dg);
      l_g = inf6(gi);
      href_times(v, dvr);
#pragma unroll
      for (int a = 0; a < 6; ++a) dvr[a] = mass * (dvr[a] - hv_ref(a)) + gi[a];
      l_dualv = inf6(dvr);
      const T si = dot6_halves(Sv, f) + w;
      l_stf = tabs(si);
      l_dstf = tabs(si - s);
      s = si;
    }
    tail_sync();

    TAIL_TP(5)
    // ================= the ten scalars the stopping logic needs, folded over the group ================================
    {
      T* row = xch + lane * LXS;
      row[0] = tmax(l_prt, l_prs); row[1] = tmax(l_dualv, l_stf); row[2] = tmax(l_dvis, l_dnu); row[3] = l_dz;
      row[4] = tmax(l_av, l_nu); row[5] = tmax(tmax(l_hrefv, l_g), l_stf); row[6] = tmax(l_dfis, tmax(l_dyis, l_dw));
      row[7] = tmax(l_dg, l_dstf); row[8] = l_up; row[9] = l_lm;
    }
    tail_sync();
    lean_fold<T>(xch, isc, gbase, jlane, G, 10, 8);
    tail_sync();
    TAIL_TP(6)
    bool finishing = false;
    if (act) {
      // (only what the logic needs is read from the instance's scalars, only what changed is written back)
      const T primal = isc[IS_RED + 0], dual = isc[IS_RED + 1], dx = isc[IS_RED + 2], dz = isc[IS_RED + 3];
      int status = (int)isc[IS_STATUS];
      const int iter = (int)isc[IS_ITER] + 1;
      const T mu_used = mu;
      bool flipped = false, feas_checked = false, tail_mode = false, tol_computed = false;
      int tail_iter = 0, c1 = 0, c2 = 0;
      T tol_p = T(0), tol_d = T(0), dyqp = T(0), atdy = T(0), ubp = T(0), lbm = T(0);
      if (P.mode & MODE_FIXED_ITERS) {
        if (iter + 1 >= P.max_iter) { status |= ST_DONE; done = true; }
      } else if (!(status & ST_TAIL)) {
        tol_computed = true;
        tol_p = P.tol_abs + P.tol_rel * tmax(isc[IS_RED + 4], isc[IS_BNORM]);
        tol_d = P.tol_abs + P.tol_rel * tmax(isc[IS_RED + 5], P.Hv_inf_norm);
        const bool conv = (primal < tol_p) && (dual < tol_d);
        bool infeas = false;
        if (iter > 1) {
          feas_checked = true;
          dyqp = isc[IS_RED + 6];
          atdy = isc[IS_RED + 7];
          c1 = atdy <= P.tol_primal_inf * dyqp;
          ubp = isc[IS_RED + 8];
          lbm = isc[IS_RED + 9];
          c2 = (ubp + lbm) <= P.tol_primal_inf * dyqp;
          infeas = c1 && c2;
        }
        if (conv) {
          status |= ST_CONVERGED | ST_DONE;
          if (infeas) status |= ST_PRIMAL_INF;
          done = true;
        } else if (infeas) {
          status |= ST_PRIMAL_INF | ST_TAIL;
          tail_mode = true;
          tail_iter = 0;
          if (!(dx >= P.tol_tail_solve || dz >= P.tol_tail_solve) || iter >= P.max_iter) { status |= ST_DONE; done = true; }
        } else {
          if (primal > T(10) * dual) { mu *= T(10); ++kexp; flipped = true; }
          else if (dual > T(10) * primal) { mu *= T(0.1); --kexp; flipped = true; }
          if (iter + 1 >= P.max_iter) { status |= ST_DONE; done = true; }
        }
      } else {
        tail_mode = true;
        tail_iter = (int)isc[IS_TAILIT] + 1;
        if (!(dx >= P.tol_tail_solve || dz >= P.tol_tail_solve) || iter >= P.max_iter) { status |= ST_DONE; done = true; }
      }
      finishing = done;
      tail_sync();  // every lane has read the instance's scalars before lane 0 rewrites them
      if (jlane == 0) {
        isc[IS_ITER] = (T)iter; isc[IS_STATUS] = (T)status;
        isc[IS_PRIMAL] = primal; isc[IS_DUAL] = dual; isc[IS_DX] = dx; isc[IS_DZ] = dz; isc[IS_MULAST] = mu_used;
        if (flipped) { isc[IS_MU] = mu; isc[IS_KEXP] = (T)kexp; isc[IS_NFLIP] = isc[IS_NFLIP] + T(1); }
        if (tail_mode) isc[IS_TAILIT] = (T)tail_iter;
        if (tol_computed) { isc[IS_TOLP] = tol_p; isc[IS_TOLD] = tol_d; }
        if (feas_checked) {
          isc[IS_C1] = (T)c1; isc[IS_C2] = (T)c2; isc[IS_DYQP] = dyqp; isc[IS_ATDY] = atdy; isc[IS_UBP] = ubp; isc[IS_LBM] = lbm;
        }
      }
    } else {
      tail_sync();
    }
    // an escaped instance (mu left the precomputed decades) is written back as it is
    const bool leaving = done && has_inst;
    // ---- the norms the getters report (13 more maxima): only when some instance of the wavefront stops ------------------
    if (__any(finishing)) {
      tail_sync();
      {
        T* row = xch + lane * LXS;
        row[0] = l_prt; row[1] = l_prs; row[2] = l_dualv; row[3] = l_stf; row[4] = l_dvis; row[5] = l_dnu; row[6] = l_dfis;
        row[7] = l_dyis; row[8] = l_dw; row[9] = l_av; row[10] = l_nu; row[11] = l_hrefv;
      }
      tail_sync();
      lean_fold<T>(xch, isc, gbase, jlane, G, 12, 12);
      tail_sync();
      if (finishing && jlane == 0) {
        isc[IS_PRT] = isc[IS_RED + 0]; isc[IS_PRS] = isc[IS_RED + 1]; isc[IS_DUALV] = isc[IS_RED + 2]; isc[IS_STF] = isc[IS_RED + 3];
        isc[IS_DVIS] = isc[IS_RED + 4]; isc[IS_DNU] = isc[IS_RED + 5]; isc[IS_DFIS] = isc[IS_RED + 6]; isc[IS_DYIS] = isc[IS_RED + 7];
        isc[IS_DW] = isc[IS_RED + 8]; isc[IS_AV] = isc[IS_RED + 9]; isc[IS_NU] = isc[IS_RED + 10]; isc[IS_HREFV] = isc[IS_RED + 11];
      }
      tail_sync();
      xch[lane * LXS + 0] = l_g;
      tail_sync();
      lean_fold<T>(xch, isc, gbase, jlane, G, 1, 1);
      tail_sync();
      if (finishing && jlane == 0) isc[IS_G] = isc[IS_RED + 0];
      tail_sync();
    }
#ifdef LOIKB_TAIL_PROF
    if (__any(leaving)) ++dbg_sw_;
#endif
    if constexpr (SLICED) {
      if (leaving) {
        store_instance();
        if (requeue) {
          q_push(lidx);
          if (jlane == 0) atomicAdd(&Bf.counters[LEAN_Q_REQUEUES], 1u);
          requeue = false;
        } else if (jlane == 0) {
          atomicAdd(q_retired, 1u);
        }
        has_inst = false; isj = false; done = true; my_iters = 0; slice_iters = 0; any_iter = false;
        q_ticket();
      }
      want_poll = leaving || (poll_skip++ & LOIKB_POLL_MASK) == 0u;
    } else {
      if (leaving) {
#ifdef LOIKB_TAIL_PROF
        __builtin_amdgcn_s_waitcnt(0); TAIL_TP(7)
        store_instance();
        __builtin_amdgcn_s_waitcnt(0); TAIL_TP(9)
        const int nx_ = fetch_plain();
        __builtin_amdgcn_s_waitcnt(0); TAIL_TP(10)
        load_instance(nx_);
        __builtin_amdgcn_s_waitcnt(0); TAIL_TP(11)
#else
        store_instance();
        load_instance(fetch_plain());
#endif
      }
    }
    TAIL_TP(7)
  }
#ifdef LOIKB_TAIL_PROF
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int k = 0; k < 8; ++k) g_tail_prof[k] = prof_[k];
    for (int k = 8; k < 12; ++k) g_tail_prof[2 + k] = prof_[k];
    g_tail_prof[8] = n_wave_iters;
    // shader clock in kHz: clock64 ticks per wall_clock64 tick (100 MHz)
    g_tail_prof[9] = (clock64() - clk0_) * 100000ull / (wall_clock64() - wall0_ + 1);
  }
  if (lane == 0) {
    const int wid = blockIdx.x * TAIL_WAVES + wv;
    if (wid < 4096) {
      g_wave_dbg[wid][0] = wall0_; g_wave_dbg[wid][1] = dbg_last_; g_wave_dbg[wid][2] = wall_clock64();
      g_wave_dbg[wid][3] = n_wave_iters; g_wave_dbg[wid][4] = dbg_both_; g_wave_dbg[wid][5] = dbg_sw_;
    }
  }
#endif
  if (lane == 0) {
    atomicAdd(&Bf.counters[5], n_wave_iters);
  }
  if (jlane == 0) {  // (per lane group: each group loads the slots of its own instances)
    atomicAdd(&Bf.counters[6], n_slot_loads >> 16);
    atomicOr(&Bf.counters[LEAN_DECADES_SEEN], n_slot_loads & 0xFFFFu);
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Decade slots: H_i (accumulated, pre-projection), Dinv_i, UDinv_i of every listed instance for mu = mu0 * 10^(kexp_lo + d),
// d = 0 .. ndec-1.  One joint per lane like k_tail, the masked leaf -> root level loop of its H rebuild (hxx:290-338,
// :31-81, H part only), once per decade.
// ------------------------------------------------------------------------------------------------------------------------
template <typename T, bool HDIAG, bool PERLINK = false>
__global__ void __launch_bounds__(WAVE)
k_hslots(const Params<T> P, const Bufs<T> Bf, const JointDesc* __restrict__ jd, const TailTopo* __restrict__ topo,
         const int* __restrict__ child_list, int maxdepth, int maxchild, const int* __restrict__ slots, int nslots, int G,
         T* __restrict__ hslots, int kexp_lo, int ndec)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Layout& L = P.L;
  T* xch = reinterpret_cast<T*>(smem_raw);  // [WAVE + 1][22]: the projected, transported H of a child
  constexpr int HX = 22;
  const int lane = threadIdx.x & (WAVE - 1);
  const int ipw = WAVE / G;
  const int sub = lane / G, jlane = lane % G, gbase = sub * G;
  const int idx = blockIdx.x * ipw + sub;
  const bool has_inst = idx < nslots;
  const bool isj = has_inst && jlane < L.nb;
  const int jl = jlane < L.nb ? jlane : 0;
  const JointDesc d = jd[jl + 1];
  const TailTopo tp = topo[jl + 1];
  const int depth = jlane < L.nb ? tp.depth : 0;
  const bool rev = d.flags & JF_REVOLUTE;
  const T mass = (d.flags & JF_MASSLESS) ? T(0) : T(1);
  const T* const hrow = PERLINK ? P.href_tab + (size_t)(jl + 1) * HREF_ROW : nullptr;  // (this link's H_ref: UpdateReferences' table)
  const bool has_parent = !(d.flags & JF_PARENT_ROOT);
  const T ax0 = (T)d.axis[0], ax1 = (T)d.axis[1], ax2 = (T)d.axis[2];
  const int slot = slots[has_inst ? idx : 0];
  const int sidx = slot;  // decade slots are indexed by the instance's slot in the set
  char* ip = lane_ptr<T>(Bf.tiles, L, slot);
  const char* rec = ip + (size_t)jl * JREC * pair_bytes<T>();
  T R[9], t[3];
  {
    const typename Vec2<T>::type cs = ldp<T>(rec, JP_CS);
    joint_xform<T>(d, rec, cs.x, cs.y, R, t);
  }
  T ata[21];
#pragma unroll
  for (int k = 0; k < 21; ++k) ata[k] = T(0);
  if (isj && d.cslot >= 0) {
    const char* crec = ip + (size_t)(L.off_c + d.cslot * L.crec) * pair_bytes<T>();
    for (int k = 0; k < 21; ++k)
      ata[k] = (P.mode & MODE_A_SHARED) ? Bf.uni[L.nc * 36 + d.cslot * 21 + k]
                                        : *reinterpret_cast<const T*>(crec + (size_t)(CP_ATA + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T));
  }
  typename Vec2<T>::type* hp = reinterpret_cast<typename Vec2<T>::type*>(hslots);
  T mu = P.mu0;  // mu of this lane's NEXT decade: mu0 * 10^kexp_lo, then x10 per decade
  for (int k = 0; k < kexp_lo; ++k) mu *= T(10);
  for (int k = 0; k > kexp_lo; --k) mu *= T(0.1);
  // The decades travel up the tree as a pipeline: at step s the joints of depth l work on decade s - (maxdepth - l), so
  // every level is busy at once (on different decades) and the whole table costs maxdepth + ndec - 1 steps instead of
  // maxdepth * ndec.  A parent reads at step s what its children wrote at step s - 1: the same decade.
  const int lag = maxdepth - depth;
  for (int st = 0; st < maxdepth + ndec - 1; ++st) {
    const int dsl = st - lag;
    const bool on = isj && depth > 0 && dsl >= 0 && dsl < ndec;
    T part[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) part[k] = T(0);
    if (on) {
      const T mu_eq = P.mu_scale * mu, mu_in = mu;
      T hh[21];
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b2 = a; b2 < 6; ++b2)
          hh[sym(a, b2)] = mass * ((a == b2 ? P.rho : T(0)) + ((HDIAG && a != b2) ? T(0) : (PERLINK ? hrow[6 * a + b2] : P.Href[6 * a + b2])));
#pragma unroll
      for (int k = 0; k < 21; ++k) hh[k] += mu_eq * ata[k];
      for (int c = 0; c < tp.nchild; ++c) {
        const T* x = xch + (gbase + child_list[tp.child_start + c]) * HX;
#pragma unroll
        for (int k = 0; k < 21; ++k) hh[k] += x[k];
      }
      T U[6], UD[6];
      T dinv;
      if (rev) {
#pragma unroll
        for (int k = 0; k < 6; ++k) U[k] = hh[sym(k, 3)] * ax0 + hh[sym(k, 4)] * ax1 + hh[sym(k, 5)] * ax2;
        dinv = T(1) / ((ax0 * U[3] + ax1 * U[4] + ax2 * U[5]) + mu_in);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) U[k] = hh[sym(k, 0)] * ax0 + hh[sym(k, 1)] * ax1 + hh[sym(k, 2)] * ax2;
        dinv = T(1) / ((ax0 * U[0] + ax1 * U[1] + ax2 * U[2]) + mu_in);
      }
      if (d.flags & JF_HELICAL) {  // S = [pitch a; a]
        const T ph = (T)d.pitch;
#pragma unroll
        for (int k = 0; k < 6; ++k) U[k] += ph * (hh[sym(k, 0)] * ax0 + hh[sym(k, 1)] * ax1 + hh[sym(k, 2)] * ax2);
        dinv = T(1) / (((ax0 * U[3] + ax1 * U[4] + ax2 * U[5]) + ph * (ax0 * U[0] + ax1 * U[1] + ax2 * U[2])) + mu_in);
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) UD[k] = U[k] * dinv;
      // the slot: H (pre-projection), Dinv, UDinv
#pragma unroll
      for (int k = 0; k < 10; ++k) hp[hslot_pair(sidx, ndec, dsl, G, k, jlane)] = typename Vec2<T>::type{hh[2 * k], hh[2 * k + 1]};
      hp[hslot_pair(sidx, ndec, dsl, G, 10, jlane)] = typename Vec2<T>::type{hh[20], dinv};
      if (has_parent) {
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b2 = a; b2 < 6; ++b2) hh[sym(a, b2)] -= UD[a] * U[b2];
        congr_sym(R, t, hh, part);
      }
      mu *= T(10);
    }
    tail_sync();  // every lane has read its children's rows of the previous step
    if (on && has_parent) {
      T* x = xch + lane * HX;
#pragma unroll
      for (int k = 0; k < 21; ++k) x[k] = part[k];
    }
    tail_sync();
  }
}

// the lean kernel's work queue at launch: ring[0 .. n) = the listed instances, the rest empty; pops start at 0, pushes at n
#ifndef LOIKB_FLAT_KERNELS_TU   // (not a template: defined in the host translation unit only)
__global__ void k_ring_fill(int* __restrict__ ring, int cap, const int* __restrict__ list, int n, unsigned int* __restrict__ counters)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) ring[i] = i < n ? list[i] : -1;
  if (i == 0) {
    counters[LEAN_Q_HEAD] = 0u; counters[LEAN_Q_TAIL] = (unsigned int)n; counters[LEAN_Q_REQUEUES] = 0u;
    counters[LEAN_Q_RETIRED] = 0u;
    counters[14] = (unsigned int)wall_clock64();  // (FLAT_COUNTERS_T0: the launch that follows starts now)
  }
}
#endif

// small batches (every instance gets a wavefront at once): the identity list, the ring and the launch's counters in ONE kernel -- a lone
// problem's Solve() is a handful of launches, and each costs the stream ~5 us (tests/loik-loid.cpp:987-1032 times exactly that call)
#ifndef LOIKB_FLAT_KERNELS_TU
__global__ void k_queue_init_iota(int* __restrict__ ring, int cap, int* __restrict__ list, int n, unsigned int* __restrict__ counters, int ncounters)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) ring[i] = i < n ? i : -1;
  if (i < n) list[i] = i;
  if (i < ncounters) {
    unsigned int v = 0u;
    if (i == LEAN_Q_TAIL) v = (unsigned int)n;
    if (i == 14) v = (unsigned int)wall_clock64();   // (FLAT_COUNTERS_T0)
    counters[i] = v;
  }
}
// ... and, for the plain Solve() of a small handle (loikb_solve: ResetRecursion + ResetSolver, then the loop -- hpp:370-374), the reset of the
// home set with it: workgroups [0, ntiles) are k_reset's, the rest k_queue_init_iota's -- one launch less in front of a lone problem's solve
template <typename T>
__global__ void __launch_bounds__(WAVE) k_reset_and_queue(char* tiles, Layout L, int what, T mu0, int ntiles, int* __restrict__ ring, int cap,
                                                          int* __restrict__ list, int n, unsigned int* __restrict__ counters, int ncounters)
{
  if ((int)blockIdx.x < ntiles) { reset_tile<T>(tiles, L, what, mu0, (int)blockIdx.x, (int)threadIdx.x); return; }
  const int i = ((int)blockIdx.x - ntiles) * WAVE + (int)threadIdx.x;
  if (i < cap) ring[i] = i < n ? i : -1;
  if (i < n) list[i] = i;
  if (i < ncounters) {
    unsigned int v = 0u;
    if (i == LEAN_Q_TAIL) v = (unsigned int)n;
    if (i == 14) v = (unsigned int)wall_clock64();   // (FLAT_COUNTERS_T0)
    counters[i] = v;
  }
}
#endif

// the listed instances that are still iterating after a lean launch (the ones that escaped), in arbitrary order
template <typename T>
// counter[1] (= Bufs::counters[4] in the on-chip launches): the listed instances that are neither converged nor flagged infeasible -- when the
// list is a whole batch and nothing is still iterating, loikb_stats::n_unfinished without a kernel, a copy and a synchronisation of its own
__global__ void k_list_unfinished(char* tiles, Layout L, const int* __restrict__ list_in, int n_in, int* __restrict__ list_out,
                                  unsigned int* __restrict__ counter)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_in) return;
  const int b = list_in[i];
  char* sp = lane_ptr<T>(tiles, L, b);
  const int status = (int)ldp<T>(sp + (size_t)L.off_s * pair_bytes<T>(), SP_ST).x;
  if (!(status & ST_DONE)) list_out[atomicAdd(counter, 1u)] = b;
  if (!(status & (ST_CONVERGED | ST_PRIMAL_INF))) atomicAdd(counter + 1, 1u);
}

// The small batches' last kernel (run_tail, small_flat: at most a few thousand listed instances, ONE workgroup): k_list_unfinished's list and
// counts, then the launch's counters go straight into the chunk's pinned host copy -- the memory is host-coherent and mapped, the stores
// are out when the kernel is: a Solve() of one problem (the reference's own call, tests/loik-loid.cpp:987-1032) ends with one
// synchronisation and no copy of its own behind the last kernel.
template <typename T>
__global__ void __launch_bounds__(256) k_small_finish(char* tiles, Layout L, const int* __restrict__ list_in, int n_in, int* __restrict__ list_out,
                                                      unsigned int* __restrict__ counters, int ncounters, unsigned int* __restrict__ host_counters)
{
  for (int i = threadIdx.x; i < n_in; i += blockDim.x) {
    const int b = list_in[i];
    char* sp = lane_ptr<T>(tiles, L, b);
    const int status = (int)ldp<T>(sp + (size_t)L.off_s * pair_bytes<T>(), SP_ST).x;
    if (!(status & ST_DONE)) list_out[atomicAdd(counters + 3, 1u)] = b;
    if (!(status & (ST_CONVERGED | ST_PRIMAL_INF))) atomicAdd(counters + 4, 1u);
  }
  __threadfence();   // (the block's atomics are performed at the L2 before anybody looks; the loads below go there too)
  __syncthreads();
  for (int k = threadIdx.x; k < ncounters; k += blockDim.x)
    host_counters[k] = __hip_atomic_load(counters + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- longest first.  The iteration counts of a batch are heavy-tailed (headline workload: median 26, mean 80, 1.2 % run into
// max_iter = 1000) and unknown before the solve; fetched in arrival order, the long runners that are fetched late keep the launch
// alive for milliseconds after the work queue ran dry.  A handle's consecutive solves resemble each other (a planner's loop: the
// same robots a step later; the reference's timing test: the same batch again): after a solve these three kernels sort the
// instances by the iteration count they just had, in ORDER_BINS classes, longest first, and the next launch of the flat engine
// takes its list in that order.  The order changes when an instance runs, not what it computes (results are bit-identical).
constexpr int ORDER_BINS = 256;
template <typename T>
__device__ __forceinline__ int order_bin(char* tiles, const Layout& L, int b, int max_iter)
{
  char* sp = lane_ptr<T>(tiles, L, b);
  const int it = (int)ldp<T>(sp + (size_t)L.off_s * pair_bytes<T>(), SP_BI).y;
  const int k = (int)(((long long)it * ORDER_BINS) / (max_iter > 0 ? max_iter : 1));
  return ORDER_BINS - 1 - (k < 0 ? 0 : k > ORDER_BINS - 1 ? ORDER_BINS - 1 : k);   // (bin 0 = the longest)
}
// bins[0 .. ORDER_BINS) counts (zeroed by the caller)
// dec_hist (may be null): [32] how many instances ended the solve in decade kexp = index - 16 of mu (clamped) -- what the handle's next
// sliced launch sizes k_fslots' window with (the lazily populated table, loik_flat2.hpp)
template <typename T>
__global__ void __launch_bounds__(256) k_order_count(char* tiles, Layout L, int n, int max_iter, unsigned int* __restrict__ bins,
                                                     unsigned int* __restrict__ dec_hist = nullptr)
{
  __shared__ unsigned int h[ORDER_BINS], hd[32];
  h[threadIdx.x] = 0u;
  if (threadIdx.x < 32) hd[threadIdx.x] = 0u;
  __syncthreads();
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n) {
    atomicAdd(&h[order_bin<T>(tiles, L, b, max_iter)], 1u);
    if (dec_hist != nullptr) {
      const int kx = (int)ldp<T>(lane_ptr<T>(tiles, L, b) + (size_t)L.off_s * pair_bytes<T>(), SP_MU).y;
      atomicAdd(&hd[kx < -16 ? 0 : kx > 15 ? 31 : kx + 16], 1u);
    }
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&bins[threadIdx.x], h[threadIdx.x]);
  if (dec_hist != nullptr && threadIdx.x < 32 && hd[threadIdx.x]) atomicAdd(&dec_hist[threadIdx.x], hd[threadIdx.x]);
}
// bins[ORDER_BINS .. 2 ORDER_BINS) <- exclusive prefix sums of the counts (one workgroup of ORDER_BINS threads)
#ifndef LOIKB_FLAT_KERNELS_TU   // (not a template: defined in the host translation unit only)
__global__ void __launch_bounds__(ORDER_BINS) k_order_scan(unsigned int* __restrict__ bins)
{
  __shared__ unsigned int h[ORDER_BINS];
  const int t = threadIdx.x;
  h[t] = bins[t];
  __syncthreads();
  for (int d = 1; d < ORDER_BINS; d <<= 1) {
    const unsigned int v = t >= d ? h[t - d] : 0u;
    __syncthreads();
    h[t] += v;
    __syncthreads();
  }
  bins[ORDER_BINS + t] = h[t] - bins[t];
}
#endif
template <typename T>
__global__ void __launch_bounds__(256) k_order_scatter(char* tiles, Layout L, int n, int max_iter, unsigned int* __restrict__ bins,
                                                       int* __restrict__ order)
{
  __shared__ unsigned int h[ORDER_BINS], base[ORDER_BINS];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  int bin = 0;
  unsigned int rank = 0u;
  if (b < n) { bin = order_bin<T>(tiles, L, b, max_iter); rank = atomicAdd(&h[bin], 1u); }
  __syncthreads();
  if (h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&bins[ORDER_BINS + threadIdx.x], h[threadIdx.x]);
  __syncthreads();
  if (b < n) order[base[bin] + rank] = b;
}

}  // namespace loikb
