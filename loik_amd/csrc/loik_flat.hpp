// loik_flat.hpp -- the ADMM iteration WITHOUT loops over the tree levels ("flat" engine, round 3).
//
// k_tail / k_lean (loik_tail.hpp, loik_lean.hpp) keep one joint per lane and walk the two recursions of an iteration --
// p, r leaf -> root (FwdPass1 + BwdPass, loik-loid-optimized.hxx:290-338, :31-81) and nu, v root -> leaf (FwdPass2,
// hxx:102-163) -- as level-synchronous loops: maxdepth steps each, one LDS round trip and ~60 dependent fp64 instructions
// per step, every lane recomputing at every level.  Half of an iteration was those two loops, and a 1000-iteration instance
// was a serial chain of 10 us per iteration that decided when a launch ends (profiles/r02_e_k_lean_phase_timeline.txt).
//
// The recursions are the sparse LDL^T solve of  (J^T H J + mu I) nu = -(J^T p^base + w - mu z)  in tree order.  With every
// spatial quantity expressed at the WORLD origin the joint-to-joint transports disappear and what is left are sums over
// subtrees / root paths of the kinematic tree, which do not need one step per level:
//
//   tau_a = (w_a - mu z_a) + S^w_a . sum_{d in subtree(a)} p^base,w_d           subtree sum of 6-vectors   (log2 G steps)
//   r'    = W tau ,   W = (I + L)^-1 ,  L_{a,d} = S^w_a . UDinv^w_d  (d below a)  scalars per (ancestor, joint) pair
//   nu    = -W^T (Dinv r')
//   v^w_i = sum_{a in root path of i} S^w_a nu_a                                path sum of 6-vectors      (log2 depth steps)
//   f^w_i = sum_{d in subtree(i)} phi^w_d ,  phi_d = H^base_d v_d + p^base_d    force balance
//   g_i   = A^T y_i - phi_i                                                      BwdPass2 (hxx:185-241) in closed form
//
// (r' is upstream's `r` after `+= S^T p`, Dinv its JointData::Dinv; W is the explicit inverse of the unit-triangular factor the
// recursion applies by substitution -- W_{a,d} != 0 only for d in the subtree of a: <= depth-1 scalars per joint.)
// W and Dinv depend on (q, mu) only; mu moves by decades (UpdateMu, hxx:613-641): k_fslots precomputes them for the decades
// mu0 * 10^(kexp_lo ..) like k_hslots did for H_i -- 10 scalars per joint and decade instead of 22 -- and an instance keeps the
// two most recent decades in LDS (the typical 1000-iteration instance flips between two).
//
// Subtree sums: the joints are numbered depth-first, so a subtree is a contiguous range of lanes [i, i + size_i): window sums
// B_k[i] = x_i + ... + x_{i+2^k-1} are built by doubling and each lane assembles its range from the windows that tile it
// (only members of the subtree are ever added: no cancellation).  Path sums: pointer jumping over the ancestors at distance
// 1, 2, 4, 8.  The sum r'_a = tau_a + sum_d W_{a,d} tau_d runs over the descendants of a (torso: 20, most joints: 0..7): the
// host deals the long rows out to lanes that have none (build_flat_schedule), <= FLAT_RED terms per lane.
//
// This first version serves H_ref = h I (the reference fixture's H_ref = I, tests/loik-loid.cpp:118-120): then
// H^base_d v_d = (rho + h) v_d and ONE subtree sum per iteration (of the links' velocities, as forces, in the world frame)
// yields both p^base,w of the next iteration and f^w of this one.  Other reference costs run in k_lean / k_tail.
//
// scripts/r03/flat_proto.py is the same arithmetic in numpy: identical iteration counts to the CPU oracle on 8192 headline
// instances (incl. the 70 that run all 999 iterations), max |dz| 4e-11.
#pragma once

#include "loik_lean.hpp"

namespace loikb {

constexpr int FLAT_RED = 8;    // terms of the W tau products one lane sums
constexpr int FLAT_PART = 8;   // partial sums a joint with a long row collects from helper lanes
constexpr int FLAT_JMP = 5;    // pointer-jumping rounds (tree depth <= 32)
constexpr int FLAT_MAXA = 16;  // strict ancestors per joint (tree depth <= 17)
constexpr int FOLDW = 10;      // scalars per lane row of the norm fold (80 B: an odd number of 16-byte slots)
constexpr int FLAT_COUNTERS_SLOT_HITS = 12;  // Bufs::counters[12]: decade changes served from the second LDS slot
constexpr int FLAT_NA_SMALL = 10;  // k_flat is compiled for <= 10 and <= FLAT_MAXA ancestors per joint
constexpr int FSLOT_ROWS = 8;      // decade slot of a joint between the passes of k_fslots: UDinv (6, world origin), Dinv, pad -- one 64-byte line

// per lane of a group: the lane's joint (lane j <-> device joint j + 1, depth-first numbering) in the static tree
struct FlatLane {
  int depth;               // 1 = child of the universe; 0 = no joint on this lane
  int size;                // joints in the subtree (incl. this one)
  int jmp[FLAT_JMP];       // lane of the ancestor at distance 2^r, -1 = beyond the root
  int anc[FLAT_MAXA];      // lane of the ancestor at depth k + 1 (k < depth - 1), else -1
  int red[FLAT_RED];       // terms this lane sums: entry k * G + lane' of the product buffer (W_{anc_k(lane'), lane'} tau_lane'), -1 = none
  int helper;              // bit 0: the sum is a partial of another joint's row (published in the partial buffer); bits 8..: the
                           // joint's column offset in a packed decade slot (fslotW_at)
  int part[FLAT_PART];     // lanes whose partials belong to this joint's row, -1 = none
};

// constraint block of an instance in LDS (T each)
enum : int { FC_LANE = 0, FC_B = 1, FC_Y = 7, FC_ATY = 13, FC_ATYW = 19, FC_ATBW = 25, FC_DY = 31, FC_DLT = 37, FC_ATYF = 43, FC_AW = 49, FC_A = 85,
              FC_VC = FC_DLT /* v of the constrained joint: written before the update's first half, which is done with it before the
                                second half writes FC_DLT */, FCD = 121 };  // (the block carries its own copy of A, shared or not: one constant stride, no selects in the loop)
// per-instance scalars kept in LDS for the getters (written when an instance stops)
enum : int { FI_BNORM = 0, FI_TGIN, FI_STY, FI_TOLP, FI_TOLD, FI_DYQP, FI_ATDY, FI_UBP, FI_LBM, FI_C1, FI_C2, FI_PRIMAL, FI_DUAL, FI_DX,
             FI_DZ, FI_MULAST, FI_RED /* 16 folded values */, FISC = FI_RED + 16 };

// LDS of one wavefront of k_flat<T, NA> (T units unless said otherwise); the regions up to `shv` have compile-time offsets
template <int NA>
__host__ __device__ constexpr int flat_xregion()
{
  int n = XROWS * 9;                                   // placement rows of the oMi chain (R, then t), scan / path-sum rows [6]
  if (NA * WAVE + 2 > n) n = NA * WAVE + 2;            // W tau products (+ a zero slot)
  if (WAVE * FOLDW > n) n = WAVE * FOLDW;              // norm fold rows
  return (n + 1) & ~1;
}
template <int NA> __host__ __device__ constexpr int flat_off_wl() { return flat_xregion<NA>(); }             // [2][NA + 1][WAVE]: W rows, Dinv row
template <int NA> __host__ __device__ constexpr int flat_off_nbuf() { return flat_off_wl<NA>() + 2 * (NA + 1) * WAVE; }  // [WAVE + 2]
template <int NA> __host__ __device__ constexpr int flat_off_pbuf() { return flat_off_nbuf<NA>() + WAVE + 2; }         // [WAVE + 2]
template <int NA> __host__ __device__ constexpr int flat_off_rbuf() { return flat_off_pbuf<NA>() + WAVE + 2; }         // [WAVE]
template <int NA> __host__ __device__ constexpr int flat_off_tail() { return flat_off_rbuf<NA>() + WAVE; }

template <typename T, int NA>
__host__ __device__ __forceinline__ size_t flat_lds_bytes(int nc, int G, bool a_shared, bool has_hv)
{
  (void)a_shared;
  const size_t per_inst = (size_t)nc * FCD + FISC;
  const size_t n = (size_t)flat_off_tail<NA>() + (has_hv ? (size_t)WAVE * 6 : 0) + (size_t)(WAVE / G) * per_inst;
  return (n * sizeof(T) + 15) & ~(size_t)15;
}

// Decade slot of an instance: one block of `fblk` scalars per decade.  Between the two passes of k_fslots it holds UDinv / Dinv of
// the joints as [lane][8] (7 used: fslotA_at); pass B overwrites it with the joints' columns PACKED one after the other
// (fslotW_at): joint j's column starts at col_j = sum_{i < j} depth_i and holds its depth_j - 1 entries W_{anc_k(j), j}, nearest
// the root first, then Dinv_j -- 168 scalars for Talos-32 where a [10 rows][32 lanes] rectangle took 320 (the rows of a rectangle
// beyond a joint's depth are zeros).  fblk = max(8 G, sum_j depth_j).  col_j travels in the upper bits of FlatLane::helper.
// (round 4: [lane][8] instead of [k][lane] -- the seven values of a joint and decade are ONE 64-byte line, written by one lane in one
//  go.  As rows [k][lane] every line collected its eight 8-byte pieces from joints of different tree levels, i.e. at different steps
//  of the pipeline, milliseconds of launch time apart for the L2: 3.9 GB written for 1.9 GB of rows and columns.)
__device__ __forceinline__ size_t fslotA_at(int idx, int ndec, int dsl, int fblk, int G, int k, int jlane)
{
  (void)G;
  return ((size_t)idx * ndec + dsl) * fblk + (size_t)jlane * 8 + k;
}
__device__ __forceinline__ size_t fslotW_at(int idx, int ndec, int dsl, int fblk, int col)
{
  return ((size_t)idx * ndec + dsl) * fblk + col;
}

// (R, t) <- (Ra, ta) o (R, t): SE3 composition, the left factor being the transform of an ancestor frame
template <typename T>
__device__ __forceinline__ void se3_compose_left(const T* Ra, const T* ta, T* R, T* t)
{
  T Rn[9], tn[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Rn[3 * i + j] = Ra[3 * i] * R[j] + Ra[3 * i + 1] * R[3 + j] + Ra[3 * i + 2] * R[6 + j];
    tn[i] = ta[i] + Ra[3 * i] * t[0] + Ra[3 * i + 1] * t[1] + Ra[3 * i + 2] * t[2];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = Rn[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = tn[k];
}

// oMi of every lane's joint by pointer jumping over liMi (4 rounds for depth <= 16): after round r, (R, t) is the placement of
// the joint in the frame of its ancestor at distance 2^(r+1) (or in the world when the root path is shorter).  xb: scratch rows
// of 9 scalars (the rotations, then the translations in the same rows); row WAVE = the identity.
template <typename T>
__device__ __forceinline__ void flat_world_placement(T* xb, int lane, int jlane, const unsigned int* jrow4, int njmp, T* R, T* t)
{
  // (jlane, not lane: one lane group of a wavefront may run this alone, inside a divergent branch)
#pragma unroll 1
  for (int r = 0; r < njmp; ++r) {
    {
      const int jr = (int)(((r < 4 ? jrow4[0] : jrow4[1]) >> (8 * (r & 3))) & 0xFFu);
      T Ra[9], ta[3];
      tail_sync();
      if (jlane < 9) xb[WAVE * 9 + jlane] = (jlane == 0 || jlane == 4 || jlane == 8) ? T(1) : T(0);
#pragma unroll
      for (int k = 0; k < 9; ++k) xb[lane * 9 + k] = R[k];
      tail_sync();
#pragma unroll
      for (int k = 0; k < 9; ++k) Ra[k] = xb[jr * 9 + k];
      tail_sync();
      if (jlane < 3) xb[WAVE * 9 + jlane] = T(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) xb[lane * 9 + k] = t[k];
      tail_sync();
#pragma unroll
      for (int k = 0; k < 3; ++k) ta[k] = xb[jr * 9 + k];
      se3_compose_left(Ra, ta, R, t);
    }
  }
  tail_sync();
}

// S_i = sum over the subtree of lane i (lanes [i, i + size)) of the 6-vectors x: window sums by doubling, each lane adds the
// windows that tile its range (low bits of `size` first).  rows: [WAVE + 1][6], row WAVE = 0.  `lim` = first lane behind the group.
template <typename T>
__device__ __forceinline__ void flat_subtree_sum(T* rows, int lane, int jlane, int lim, int size, int nscan, const T* x, T* S)
{
  // (the region is shared with other phases: the zero row is re-established; by jlane -- a lane group may run this alone)
  tail_sync();
  if (jlane < 6) rows[WAVE * 6 + jlane] = T(0);
  T B[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { B[k] = x[k]; S[k] = T(0); }
  int pos = lane;
  for (int k = 0; k < nscan; ++k) {
    tail_sync();
#pragma unroll
    for (int c = 0; c < 6; ++c) rows[lane * 6 + c] = B[c];
    tail_sync();
    const bool take = (size >> k) & 1;
    const int ra = take ? pos : WAVE;
    const int nx = lane + (1 << k);
    const int rb = nx < lim ? nx : WAVE;
    T a[6], b[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) a[c] = rows[ra * 6 + c];
#pragma unroll
    for (int c = 0; c < 6; ++c) b[c] = rows[rb * 6 + c];
#pragma unroll
    for (int c = 0; c < 6; ++c) { S[c] += a[c]; B[c] += b[c]; }
    pos += take ? (1 << k) : 0;
  }
}

// y_i = sum over the root path of lane i (the joint and its ancestors) of the 6-vectors x: pointer jumping
template <typename T>
__device__ __forceinline__ void flat_path_sum(T* rows, int lane, int jlane, const unsigned int* jrow4, int njmp, T* y)
{
  tail_sync();
  if (jlane < 6) rows[WAVE * 6 + jlane] = T(0);
#pragma unroll 1
  for (int r = 0; r < njmp; ++r) {
    {
      const int jr = (int)(((r < 4 ? jrow4[0] : jrow4[1]) >> (8 * (r & 3))) & 0xFFu);
      tail_sync();
#pragma unroll
      for (int c = 0; c < 6; ++c) rows[lane * 6 + c] = y[c];
      tail_sync();
      T a[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) a[c] = rows[jr * 6 + c];
#pragma unroll
      for (int c = 0; c < 6; ++c) y[c] += a[c];
    }
  }
}

// force-like 6-vector of a link frame -> world origin: (R0 f_l, R0 f_a + t0 x R0 f_l)   [SE3::act(Force)]
// motion world -> link: actinv_motion(R0, t0, ...); force world -> link: (R0^T F_l, R0^T (F_a - t0 x F_l))
template <typename T>
__device__ __forceinline__ void actinv_force(const T* R, const T* t, const T* F, T* o)
{
  T c[3], d[3];
  cross3(t, F, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = F[3 + k] - c[k];
  mat3t_vec(R, F, o);
  mat3t_vec(R, d, o + 3);
}

// fold 8 columns of the group's rows: columns < NMAX by max, the others by sum (lane order); every lane returns with all
// eight results.  rows: [WAVE][FOLDW]; columns 8 / 9 of a row are scratch.
template <typename T, int NMAX>
__device__ __forceinline__ void flat_fold8(T* rows, int lane, int gbase, int jlane, int lgG, const T* in, T* out)
{
  tail_sync();
  T* row = rows + lane * FOLDW;
#pragma unroll
  for (int q = 0; q < 8; ++q) row[q] = in[q];
  tail_sync();
  const int q = jlane & 7;
  const bool ismax = q < NMAX;
  const int nparts = 1 << (lgG - 3);
  {
    // lane (q, part) folds column q over the rows part, part + nparts, part + 2 nparts, ... of its group (eight rows; rows
    // interleaved over the parts: with a contiguous block of eight rows per part -- 640 B apart -- all parts of the wavefront
    // hit the same banks, an eight-way conflict on each of the eight reads)
    const T* col = rows + (gbase + (jlane >> 3)) * FOLDW + q;
    T a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = col[u * nparts * FOLDW];
    const T m = tmax(tmax(tmax(a[0], a[1]), tmax(a[2], a[3])), tmax(tmax(a[4], a[5]), tmax(a[6], a[7])));
    T sm = a[0];
#pragma unroll
    for (int u = 1; u < 8; ++u) sm += a[u];
    row[8] = (NMAX >= 8 || ismax) ? m : sm;
  }
  tail_sync();
  {
    const T* col = rows + (gbase + q) * FOLDW + 8;  // (lane (q, part) left its partial in its own row: gbase + q + 8 part)
    T a[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) a[p] = (p < 4 || lgG > 5) ? col[p * 8 * FOLDW] : T(0);  // (G = 32: four parts, G = 64: eight)
    const T m = tmax(tmax(tmax(a[0], a[1]), tmax(a[2], a[3])), tmax(tmax(a[4], a[5]), tmax(a[6], a[7])));
    T sm = a[0];
#pragma unroll
    for (int p = 1; p < 8; ++p) sm += a[p];
    row[9] = (NMAX >= 8 || ismax) ? m : sm;
  }
  tail_sync();
#pragma unroll
  for (int u = 0; u < 8; ++u) out[u] = rows[(gbase + u) * FOLDW + 9];
}

// packed per-lane address tables: four lane indices (bytes) / two product indices (16 bit) per register.  The tables are
// loop-invariant; left alone, the compiler unpacks them once before the iteration loop and keeps (spills) ~30 addresses --
// `opaque` makes a register's content unknown to it at the point of use, so the unpacking stays where the address is needed.
__device__ __forceinline__ unsigned int opaque(unsigned int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int unpack8(const unsigned int* w, int k) { return (int)((opaque(w[k >> 2]) >> (8 * (k & 3))) & 0xFFu); }
__device__ __forceinline__ int unpack16(const unsigned int* w, int k) { return (int)((opaque(w[k >> 1]) >> (16 * (k & 1))) & 0xFFFFu); }

// (A second build of this kernel -- two wavefronts per SIMD at 256 registers, ~100 values of the iteration in scratch, drained
// and relaunched in this build once the queue had run dry -- was measured through round 3 and removed: slower in bulk (21 against
// 17 ms on the headline) and no faster in the tail.  k_flat2 is what two wavefronts per SIMD take: loik_flat2.hpp.)
// mu of decade k as k_fslots and the builders form it (ONE definition: repeated products from mu0 upwards or downwards)
__device__ __forceinline__ double flat_decade_mu(double mu0, int k)
{
  double mu = mu0;
  for (int i = 0; i < k; ++i) mu *= 10.0;
  for (int i = 0; i > k; --i) mu *= 0.1;
  return mu;
}
constexpr int FLAT_COUNTERS_DRY = 13;  // Bufs::counters[13]: set by the first lane group that finds the work queue empty
constexpr int FLAT_COUNTERS_ERR = 16;  // Bufs::counters[16]: a wavefront gave up waiting on the work queue (never: reported as an error)
constexpr int FLAT_COUNTERS_T0 = 14, FLAT_COUNTERS_TDRY = 15;  // the 100 MHz clock (low word) when the ring was filled / when the queue ran dry
// LOG: the lists of LoikSolverInfo (loik-loid-optimized.hpp:47-127, filled at hpp:406-420 -- after ComputeResiduals, before the
// stopping tests, so mu_list_ holds the mu the iteration RAN with) are written from here: nine scalars per main-loop iteration
// (one more fold for the four residuals the stopping logic only needs combined).  An instantiation of its own.
template <typename T, int NA, bool LOG = false>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_flat(const Params<T> P, const Bufs<T> Bf, const JointDesc* __restrict__ jd, const FlatLane* __restrict__ fl, int nanc, int nscan,
       int njmp, const int* __restrict__ ring, int nslots, int lgG, const T* __restrict__ fslots, int frows, int kexp_lo, int ndec,
       T href_s, int has_hv)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Layout& L = P.L;
  const bool a_shared = P.mode & MODE_A_SHARED;
  constexpr int cs = FCD;
  const int lane = threadIdx.x;
  const int G = 1 << lgG;
  const int sub = lane >> lgG, jlane = lane & (G - 1), gbase = sub << lgG, glim = gbase + G;
  // ---- LDS of the wavefront (one wavefront per workgroup)
  T* const xb = reinterpret_cast<T*>(smem_raw);          // placement rows | scan rows | W tau products | fold rows (one after the other)
  T* const wl = xb + flat_off_wl<NA>();                  // [2][NA + 1][WAVE]  W rows and the Dinv row of two decades of mu
  T* const nbuf = xb + flat_off_nbuf<NA>();              // [WAVE + 2]         Dinv r' of every joint (+ a zero)
  T* const pbuf = xb + flat_off_pbuf<NA>();              // [WAVE + 2]         partial sums of long rows
  T* const rbuf = xb + flat_off_rbuf<NA>();              // [WAVE]             r' of the last iteration (stored with the instance)
  T* const shv = xb + flat_off_tail<NA>();               // [WAVE][6]          subtree sums of the links' H_ref v_ref (if != 0)
  T* const cd = shv + (has_hv ? WAVE * 6 : 0);           // [64/G][nc][FCD]    constraint blocks of the instances
  T* const iscb = cd + (size_t)(WAVE >> lgG) * L.nc * cs;  // [64/G][FISC]
  T* const cdi = cd + (size_t)sub * L.nc * cs;
  T* const isc = iscb + (size_t)sub * FISC;

  const bool isj_lane = jlane < L.nb;
  const int jl = isj_lane ? jlane : 0;
  // (the joint's placement is only needed when an instance is loaded: load_instance reads the description again instead of
  //  keeping 15 scalars of it alive through the iteration loop)
  const int jflags = jd[jl + 1].flags, jcslot = isj_lane ? jd[jl + 1].cslot : -1;
  const T mass = (!isj_lane || (jflags & JF_MASSLESS)) ? T(0) : T(1);
  int size;
  unsigned int jrow4[(FLAT_JMP + 3) / 4], ra2[FLAT_RED / 2], prow4[(FLAT_PART + 3) / 4], anc4[(NA + 3) / 4];
  bool helper;
  int fcol, fdm1;  // this joint's column in a packed decade slot, its number of ancestors
  {
    // static rows / entries of this group's lanes; helper lanes may be lanes without a joint
    const FlatLane F = fl[jlane];
    size = isj_lane ? F.size : 0;
    helper = (F.helper & 1) != 0;
    fcol = F.helper >> 8; fdm1 = F.depth > 0 ? F.depth - 1 : 0;
#pragma unroll
    for (int k = 0; k < (FLAT_JMP + 3) / 4; ++k) jrow4[k] = 0u;
#pragma unroll
    for (int k = 0; k < FLAT_RED / 2; ++k) ra2[k] = 0u;
#pragma unroll
    for (int k = 0; k < (FLAT_PART + 3) / 4; ++k) prow4[k] = 0u;
#pragma unroll
    for (int k = 0; k < (NA + 3) / 4; ++k) anc4[k] = 0u;
#pragma unroll
    for (int r = 0; r < FLAT_JMP; ++r) jrow4[r >> 2] |= (unsigned int)(F.jmp[r] >= 0 ? gbase + F.jmp[r] : WAVE) << (8 * (r & 3));
#pragma unroll
    for (int t = 0; t < FLAT_RED; ++t)
      ra2[t >> 1] |= (unsigned int)(F.red[t] >= 0 ? (F.red[t] >> lgG) * WAVE + gbase + (F.red[t] & (G - 1)) : NA * WAVE) << (16 * (t & 1));
#pragma unroll
    for (int j = 0; j < FLAT_PART; ++j) prow4[j >> 2] |= (unsigned int)(F.part[j] >= 0 ? gbase + F.part[j] : WAVE) << (8 * (j & 3));
#pragma unroll
    for (int k = 0; k < NA; ++k) anc4[k >> 2] |= (unsigned int)((k < FLAT_MAXA && F.anc[k] >= 0) ? gbase + F.anc[k] : WAVE) << (8 * (k & 3));
  }
  if (jlane < 2) { nbuf[WAVE + jlane] = T(0); pbuf[WAVE + jlane] = T(0); }
#pragma unroll
  for (int k = 0; k <= NA; ++k) { wl[k * WAVE + lane] = T(0); wl[(NA + 1 + k) * WAVE + lane] = T(0); }

  // ---- the instance of this lane group
  bool has_inst = false, isj = false, done = true, any_iter = false, need_load = true;
  int lidx = 0;
  char *ip = Bf.tiles, *rec = Bf.tiles;
  T R0[9], t0[3], Sw[6], v[6], f[6], g[6], SE[6];
  T w = T(0), z = T(0), nu = T(0), s = T(0), lbi = T(0), ubi = T(0), mu = T(1);
  int kexp = 0, kslot = -(1 << 30), kslot_o = -(1 << 30), wsel = 0;
  int iter = 0, status = ST_DONE, tail_it = 0, nflip = 0;
  unsigned int my_iters = 0, n_wave_iters = 0, n_slot_loads = 0, n_slot_hits = 0;
  unsigned int* q_head = Bf.counters + LEAN_Q_HEAD;

  // constraint c: is its joint in this lane's subtree?  (1 / 0; the bits are set when an instance is loaded)
  unsigned int cbits = 0u;
  auto cmask = [&](int c) -> T { return ((cbits >> c) & 1u) ? T(1) : T(0); };
  // the DualUpdate's lanes: lane 6 c + k of the group owns row k of constraint c (all constraints side by side)
  const int ccl = jlane / 6, ckl = jlane - 6 * ccl;
  const bool iscl = jlane < 6 * L.nc;
  T* const ccb = cdi + (iscl ? ccl : 0) * cs;
  // links' velocities as forces at the world origin: E = mass * (R0 v_l, R0 v_a + t0 x R0 v_l), from the world-frame motion
  // (vw_l = R0 v_l + t0 x R0 v_a, vw_a = R0 v_a):  R0 v_l = vw_l - t0 x vw_a
  auto force_of_motion = [&](const T* vw, T* E) {
    T c1[3], c2[3];
    cross3(t0, vw + 3, c1);
#pragma unroll
    for (int k = 0; k < 3; ++k) E[k] = vw[k] - c1[k];
    cross3(t0, E, c2);
#pragma unroll
    for (int k = 0; k < 3; ++k) E[3 + k] = vw[3 + k] + c2[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) E[k] *= mass;
  };
  auto load_instance = [&]() {
    int slot_in;
    {
      int nx = 0;
      if (jlane == 0) nx = (int)atomicAdd(q_head, 1u);
      nx = __shfl(nx, gbase);
      slot_in = nx < nslots ? ring[nx] : -1;
      if (nx >= nslots && jlane == 0) {
        // the queue is empty: the first group to find it so notes the time (the launch's bulk phase ends here: from now on lane
        // groups idle and the launch waits for its long runners)
        if (atomicCAS(Bf.counters + FLAT_COUNTERS_DRY, 0u, 1u) == 0u)
          __hip_atomic_store(Bf.counters + FLAT_COUNTERS_TDRY, (unsigned int)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    has_inst = slot_in >= 0;
    isj = has_inst && isj_lane;
    const int slot = has_inst ? slot_in : 0;
    lidx = slot;
    ip = lane_ptr<T>(Bf.tiles, L, slot);
    rec = ip + (size_t)jl * JREC * pair_bytes<T>();
    const char* srec = ip + (size_t)L.off_s * pair_bytes<T>();
    T ax[3];
    const bool rev = jflags & JF_REVOLUTE;
    const int jflags_h = jflags;
    const T pitch_h = (jflags & JF_HELICAL) ? (T)jd[jl + 1].pitch : T(0);
    {
      const typename Vec2<T>::type csn = ldp<T>(rec, JP_CS), wz = ldp<T>(rec, JP_WZ), nus = ldp<T>(rec, JP_NUS);
      const JointDesc d = jd[jl + 1];
#pragma unroll
      for (int k = 0; k < 3; ++k) ax[k] = isj_lane ? (T)d.axis[k] : T(0);
      joint_xform<T>(d, rec, csn.x, csn.y, R0, t0);  // liMi ...
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
    }
    if (!isj_lane) {
#pragma unroll
      for (int k = 0; k < 9; ++k) R0[k] = (k % 4 == 0) ? T(1) : T(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) t0[k] = T(0);
#pragma unroll
      for (int k = 0; k < 6; ++k) { v[k] = T(0); f[k] = T(0); g[k] = T(0); }
      w = z = nu = s = T(0);
    }
    flat_world_placement<T>(xb, lane, jlane, jrow4, njmp, R0, t0);  // ... -> oMi (FwdPassInit's oMi chain, hxx:265)
    {
      // S^w: the joint's motion subspace at the world origin (R0 S_l + t0 x R0 S_a, R0 S_a)
      T ra3[3], c[3];
      mat3_vec(R0, ax, ra3);
      cross3(t0, ra3, c);
#pragma unroll
      for (int k = 0; k < 3; ++k) { Sw[k] = rev ? c[k] : ra3[k]; Sw[3 + k] = rev ? ra3[k] : T(0); }
      if (jflags_h & JF_HELICAL) {  // S = [pitch a; a] at the world origin: (t x R a + pitch R a, R a)
#pragma unroll
        for (int k = 0; k < 3; ++k) Sw[k] += pitch_h * ra3[k];
      }
    }
    // constraint blocks: lane of the joint, b, y, A^T y (+ A); then AW = X*_{0<-joint} A^T, AW b
    for (int c = 0; c < L.nc; ++c) {
      const char* crec = ip + (size_t)(L.off_c + c * L.crec) * pair_bytes<T>();
      T* c_ = cdi + c * cs;
      if (jlane < 18) {
        const int which = jlane / 6, k = jlane % 6;
        const int pair = which == 0 ? CP_B : which == 1 ? CP_Y : CP_ATY;
        const int dst = which == 0 ? FC_B : which == 1 ? FC_Y : FC_ATY;
        c_[dst + k] = *reinterpret_cast<const T*>(crec + (size_t)(pair + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T));
      }
      for (int e = jlane; e < LCA; e += G)
        c_[FC_A + e] = a_shared ? Bf.uni[c * LCA + e]
                                : *reinterpret_cast<const T*>(crec + (size_t)(CP_A + e / 2) * pair_bytes<T>() + (e & 1) * sizeof(T));
    }
    if (jcslot >= 0) cdi[jcslot * cs + FC_LANE] = (T)jlane;
    tail_sync();
    cbits = 0u;
    for (int c = 0; c < L.nc; ++c) {
      const int cl = (int)cdi[c * cs + FC_LANE];
      if (isj_lane && cl >= jlane && cl < jlane + size) cbits |= 1u << c;
    }
    if (jcslot >= 0) {  // the constrained joint's lane: column j of AW = row j of A carried to the world origin
      T* c_ = cdi + jcslot * cs;
      const T* A_ = c_ + FC_A;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        T aj[6], o[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) aj[k] = A_[6 * j + k];
        act_force(R0, t0, aj, o);
#pragma unroll
        for (int k = 0; k < 6; ++k) c_[FC_AW + 6 * k + j] = o[k];
      }
      // A^T y as the instance brings it (a warm-started tailored solve arrives with the A^T y of the matrix it had BEFORE
      // UpdateEqConstraint replaced it, and upstream's first FwdPass1 uses exactly that, hxx:329-331), at the world origin
      T ay[6], o[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) ay[k] = c_[FC_ATY + k];
      act_force(R0, t0, ay, o);
#pragma unroll
      for (int k = 0; k < 6; ++k) c_[FC_ATYW + k] = o[k];
    }
    tail_sync();
    for (int c = 0; c < L.nc; ++c) {
      T* c_ = cdi + c * cs;
      if (jlane < 6) {
        const int k = jlane;
        T ab = T(0);
#pragma unroll
        for (int j = 0; j < 6; ++j) ab += c_[FC_AW + 6 * k + j] * c_[FC_B + j];
        c_[FC_ATBW + k] = ab;
      }
    }
    // subtree sums of the state the instance arrives with (cold start: v = 0) and of the reference term
    {
      T vw[6], E[6];
      T a[3], l[3], c[3];
      mat3_vec(R0, v, l);
      mat3_vec(R0, v + 3, a);
      cross3(t0, a, c);
#pragma unroll
      for (int k = 0; k < 3; ++k) { vw[k] = l[k] + c[k]; vw[3 + k] = a[k]; }
      force_of_motion(vw, E);
      flat_subtree_sum<T>(xb, lane, jlane, glim, size, nscan, E, SE);
      if (has_hv) {
        T hv[6], hw[6], Sh[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) hv[k] = mass * P.Hv[k];
        act_force(R0, t0, hv, hw);
        flat_subtree_sum<T>(xb, lane, jlane, glim, size, nscan, hw, Sh);
#pragma unroll
        for (int k = 0; k < 6; ++k) shv[lane * 6 + k] = Sh[k];
      }
    }
    const typename Vec2<T>::type mu2 = ldp<T>(srec, SP_MU), bi2 = ldp<T>(srec, SP_BI), st2 = ldp<T>(srec, SP_ST);
    mu = mu2.x;
    kexp = (int)mu2.y;
    kslot = -(1 << 30); kslot_o = -(1 << 30);
    status = has_inst ? (int)st2.x : ST_DONE;
    iter = (int)bi2.y;
    tail_it = (int)ld_scal<T>(srec, SC_TAIL_ITER);
    nflip = (int)ldp<T>(srec, SP_FLIP).x;
    done = (status & ST_DONE) != 0;
    if (!done && !(status & ST_TAIL) && iter + 1 >= P.max_iter) { done = true; status |= ST_DONE; }
    if (jlane == 0) {
      isc[FI_BNORM] = bi2.x; isc[FI_TGIN] = ldp<T>(srec, SP_TAG).x; isc[FI_STY] = st2.y; isc[FI_MULAST] = T(-1);
      isc[FI_TOLP] = ld_scal<T>(srec, SC_TOL_PRIMAL); isc[FI_TOLD] = ld_scal<T>(srec, SC_TOL_DUAL);
      isc[FI_DYQP] = ld_scal<T>(srec, SC_DELTA_Y_QP); isc[FI_ATDY] = ld_scal<T>(srec, SC_AT_DELTA_Y_QP);
      isc[FI_UBP] = ld_scal<T>(srec, SC_UB_DY_PLUS); isc[FI_LBM] = ld_scal<T>(srec, SC_LB_DY_MINUS);
      isc[FI_C1] = ld_scal<T>(srec, SC_COND1); isc[FI_C2] = ld_scal<T>(srec, SC_COND2);
    }
    tail_sync();
    my_iters = 0;
    any_iter = false;
  };
  auto store_instance = [&]() {
    char* srec = ip + (size_t)L.off_s * pair_bytes<T>();
    if (isj) {
      st6<T>(rec, JP_V, v);
      st6<T>(rec, JP_F, f);
      st6<T>(rec, JP_G, g);
      stp<T>(rec, JP_WZ, w, z);
      stp<T>(rec, JP_NUS, nu, s);
      if (any_iter) {
        // inter-sweep temporaries of the last iteration: r_i and Dinv_i.  This engine forms neither UDinv_i nor the
        // accumulated p_i: the scalar record's tag says so (SP_TAG = -2) and the getters rebuild them (k_rebuild_ud).
        stp<T>(rec, JP_R, rbuf[lane], wl[(wsel * (NA + 1) + NA) * WAVE + lane]);
      }
    }
    if (has_inst) {
      for (int c = 0; c < L.nc; ++c) {
        char* crec = ip + (size_t)(L.off_c + c * L.crec) * pair_bytes<T>();
        if (jlane < 6) {
          const int k = jlane;
          *reinterpret_cast<T*>(crec + (size_t)(CP_Y + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T)) = cdi[c * cs + FC_Y + k];
          *reinterpret_cast<T*>(crec + (size_t)(CP_ATY + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T)) = cdi[c * cs + FC_ATY + k];
        }
      }
      if (LOG && jlane == 0 && any_iter) Bf.log_rows[lidx] = iter - ((status & ST_TAIL) ? tail_it : 0);  // (main-loop iterations)
      if (jlane == 0) {
        stp<T>(srec, SP_MU, mu, (T)kexp);
        stp<T>(srec, SP_TAG, any_iter ? T(-2) : isc[FI_TGIN], T(0));
        stp<T>(srec, SP_BI, isc[FI_BNORM], (T)iter);
        stp<T>(srec, SP_FLIP, (T)nflip, T(0));
        stp<T>(srec, SP_ST, (T)(any_iter ? (status & ~ST_PFULL) : status), any_iter ? isc[FI_MULAST] : isc[FI_STY]);
        if (any_iter) {
          const T* rr = isc + FI_RED;  // prt prs stf dvis dnu dfis dyis dw av nu hrefv g dualv (13), filled when the instance stopped
          const T mu_s = mu;
          stp<T>(srec, SP_SCAL + 0, isc[FI_PRIMAL], isc[FI_DUAL]);
          stp<T>(srec, SP_SCAL + 1, rr[0], rr[1]);
          stp<T>(srec, SP_SCAL + 2, rr[12], rr[2]);
          stp<T>(srec, SP_SCAL + 3, isc[FI_TOLP], isc[FI_TOLD]);
          stp<T>(srec, SP_SCAL + 4, mu_s, P.mu_scale * mu_s);
          stp<T>(srec, SP_SCAL + 5, mu_s, isc[FI_DX]);
          stp<T>(srec, SP_SCAL + 6, isc[FI_DZ], isc[FI_DYQP]);
          stp<T>(srec, SP_SCAL + 7, isc[FI_ATDY], isc[FI_UBP]);
          stp<T>(srec, SP_SCAL + 8, isc[FI_LBM], rr[5]);
          stp<T>(srec, SP_SCAL + 9, rr[6], rr[7]);
          stp<T>(srec, SP_SCAL + 10, rr[3], rr[4]);
          stp<T>(srec, SP_SCAL + 11, rr[8], rr[9]);
          stp<T>(srec, SP_SCAL + 12, rr[10], rr[11]);
          stp<T>(srec, SP_SCAL + 13, rr[2], isc[FI_C1]);
          stp<T>(srec, SP_SCAL + 14, isc[FI_C2], (T)tail_it);
        }
        if (my_iters) atomicAdd(&Bf.counters[1], my_iters);
      }
    }
    tail_sync();
  };

#ifdef LOIKB_TAIL_PROF
  unsigned long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = clock64();
  const unsigned long long wall0_ = wall_clock64(), clk0_ = tprev_;
#endif
  while (true) {
    if (need_load) { load_instance(); need_load = false; }
    if (!__any(has_inst)) break;
    // ---- does the instance leave before this iteration?  (fetched already finished; this launch's share of iterations used
    // up; the queue has run dry and the latency build takes over; mu left the precomputed decades.)  Then it goes back as it
    // is, and the group idles through this iteration: a group without an instance computes on garbage, inside its own lanes
    // and LDS rows, and nothing of it is kept -- so the iteration below updates its registers without predicates.
    bool exit_now = has_inst && (done || (int)my_iters >= P.max_launch_iters);
    // ---- decade of mu: W rows and Dinv.  Two decades stay in LDS: a flip back to the previous one costs nothing.
    if (has_inst && !exit_now && kexp != kslot) {
      if (kexp == kslot_o) {
        { const int tk = kslot; kslot = kslot_o; kslot_o = tk; }
        wsel ^= 1;
        ++n_slot_hits;
      } else {
        const int dsl = kexp - kexp_lo;
        if (dsl < 0 || dsl >= ndec) {
          exit_now = true;  // mu left the precomputed decades: written back unfinished, k_tail takes over
          if (jlane == 0) atomicAdd(&Bf.counters[2], 1u);
        } else {
          kslot_o = kslot;  // the slot that was not used last is overwritten
          wsel ^= 1;
          T* wdst = wl + (size_t)wsel * (NA + 1) * WAVE;
          // (k_fslots precomputes the joints' W columns for every decade.  Measured and rejected: slots that hold UDinv / Dinv
          //  only -- 7 scalars instead of 10 -- with the column built here when an instance enters a decade: the instances that
          //  cycle through three decades rebuild at every other move, +12 % launch time on the headline.)
          if (isj) {
            T in[NA + 1];
#pragma unroll
            for (int k = 0; k <= NA; ++k)
              in[k] = (k == NA || k < fdm1) ? fslots[fslotW_at(lidx, ndec, dsl, frows, fcol + (k == NA ? fdm1 : k))] : T(0);  // (entries beyond the
            // tree's depth repeat the Dinv row: such a W entry only ever multiplies the zero behind the root)
#pragma unroll
            for (int k = 0; k <= NA; ++k) wdst[k * WAVE + lane] = in[k];
          } else {
#pragma unroll
            for (int k = 0; k <= NA; ++k) wdst[k * WAVE + lane] = T(0);
          }
          kslot = kexp;
          n_slot_loads = (n_slot_loads + 0x10000u) | (1u << dsl);
        }
      }
    }
    if (exit_now) {
      store_instance();
      has_inst = false; isj = false; done = true;
      need_load = true;
    }
    const T* wcur = wl + (size_t)wsel * (NA + 1) * WAVE;  // (per lane group: wsel differs between the groups of a wavefront)
    TAIL_TP(8)
    const bool act = has_inst;  // (an instance that is here iterates: `done` is only set by the stopping logic below)
    const T mu_eq = P.mu_scale * mu, mu_in = mu;
    if (act) { ++my_iters; any_iter = true; }
    ++n_wave_iters;

    // ================= p^base at the world origin, summed over the subtrees; tau  (FwdPass1 + the p part of BwdPass) ===========
    T wc[NA];  // this joint's W entries (its ancestors' rows): used twice, up and down
#pragma unroll
    for (int k = 0; k < NA; ++k) wc[k] = wcur[k * WAVE + lane];
    const T dinv = wcur[NA * WAVE + lane];
    T tau;
    {
      T PB[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) PB[k] = -P.rho * SE[k];
      if (has_hv) {
#pragma unroll
        for (int k = 0; k < 6; ++k) PB[k] -= shv[lane * 6 + k];
      }
      for (int c = 0; c < L.nc; ++c) {
        const T* c_ = cdi + c * cs;
        const T m = cmask(c);
#pragma unroll
        for (int k = 0; k < 6; ++k) PB[k] += m * (c_[FC_ATYW + k] - mu_eq * c_[FC_ATBW + k]);
      }
      tau = (w - mu_in * z) + dot6_halves(Sw, PB);
    }
    TAIL_TP(0)
    // ================= r' = W tau: products to LDS, every lane sums its share, long rows collect their partials ================
    tail_sync();
#pragma unroll
    for (int k = 0; k < NA; ++k) xb[k * WAVE + lane] = wc[k] * tau;
    if (jlane == 0) { xb[NA * WAVE] = T(0); xb[NA * WAVE + 1] = T(0); }
    tail_sync();
    T rn;
    {
      T a[FLAT_RED];
#pragma unroll
      for (int t = 0; t < FLAT_RED; ++t) a[t] = xb[unpack16(ra2, t)];
      T acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      pbuf[lane] = helper ? acc : T(0);
      tail_sync();
      T pp[FLAT_PART];
#pragma unroll
      for (int j = 0; j < FLAT_PART; ++j) pp[j] = pbuf[unpack8(prow4, j)];
      if (helper) acc = T(0);
#pragma unroll
      for (int j = 0; j < FLAT_PART; ++j) acc += pp[j];
      rn = tau + acc;
      rbuf[lane] = rn;
      nbuf[lane] = dinv * rn;
    }
    tail_sync();
    TAIL_TP(1)
    // ================= nu = -W^T (Dinv r')  (FwdPass2's nu_i, hxx:127) ======================================================
    T nui;
    {
      T nb_[NA];
#pragma unroll
      for (int k = 0; k < NA; ++k) nb_[k] = nbuf[unpack8(anc4, k)];
      T acc = dinv * rn;
#pragma unroll
      for (int k = 0; k < NA; ++k) acc += wc[k] * nb_[k];
      nui = -acc;
    }
    // ================= v = J nu: path sum of S^w nu at the world origin, then into the link frame (hxx:125-134) ===============
    T vi[6], E[6];
    {
      // (measured and rejected: the ancestors' S^w in 120 registers and only their nu fetched -- one exchange instead of four,
      //  but 66 dependent-ish fp64 multiply-adds and the registers they displace: 1.88 k cycles against 1.54 k)
      T vw[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) vw[k] = Sw[k] * nui;
      flat_path_sum<T>(xb, lane, jlane, jrow4, njmp, vw);
      actinv_motion(R0, t0, vw, vi);
      force_of_motion(vw, E);
    }
    TAIL_TP(2)
    // ================= DualUpdate of the task constraints (hxx:410-451) inside the subtree sum of the links' velocities ==========
    // f by force balance at the world origin needs one subtree sum (BwdPass2's transport, hxx:210-212): window sums by doubling,
    // one LDS exchange per step.  The constraints' update -- (A v - b, dy, y) and then (A^T y, the same at the world origin, the
    // pieces of this iteration's force balance) -- needs two exchanges of its own: it rides on the first two steps.
    T l_dyis = T(0), l_av = T(0), l_prt = T(0), l_up = T(0), l_lm = T(0);
    T l_nu = T(0), l_dfis = T(0), l_hrefv = T(0), l_dvis = T(0), l_dnu = T(0), l_dz = T(0), l_dw = T(0), l_prs = T(0);
    T l_dg = T(0), l_g = T(0), l_stf = T(0), l_dstf = T(0), l_dualv = T(0);
    T fi[6], si;
    {
      T SEn[6], Fw[6], Bk[6];
      int pos = lane;
#pragma unroll
      for (int k = 0; k < 6; ++k) { Bk[k] = E[k]; SEn[k] = T(0); }
      auto put_rows = [&]() {
        tail_sync();
#pragma unroll
        for (int c = 0; c < 6; ++c) xb[lane * 6 + c] = Bk[c];
      };
      auto scan_step = [&](int k) {  // (the rows hold the window sums of width 2^k)
        const bool take = (size >> k) & 1;
        const int ra_ = take ? pos : WAVE;
        const int nx = lane + (1 << k);
        const int rb_ = nx < glim ? nx : WAVE;
        T a[6], b[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) a[c] = xb[ra_ * 6 + c];
#pragma unroll
        for (int c = 0; c < 6; ++c) b[c] = xb[rb_ * 6 + c];
#pragma unroll
        for (int c = 0; c < 6; ++c) { SEn[c] += a[c]; Bk[c] += b[c]; }
        pos += take ? (1 << k) : 0;
      };
      put_rows();
      if (jlane < 6) xb[WAVE * 6 + jlane] = T(0);
      if (jcslot >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cdi[jcslot * cs + FC_VC + k] = vi[k];
      }
      tail_sync();
      scan_step(0);
      if (iscl) {
        const T* A_ = ccb + FC_A;
        const T* vc = ccb + FC_VC;
        T avk = A_[6 * ckl] * vc[0];
#pragma unroll
        for (int j = 1; j < 6; ++j) avk += A_[6 * ckl + j] * vc[j];
        const T bk = ccb[FC_B + ckl];
        const T ek = avk - bk;
        const T dy = mu_eq * ek;
        const T yk = ccb[FC_Y + ckl] + dy;
        l_dyis = tabs(dy);
        l_up = bk * tmax(dy, T(0));
        l_lm = bk * tmin(dy, T(0));
        l_prt = tabs(ek);
        l_av = tabs(avk);
        ccb[FC_Y + ckl] = yk;
        ccb[FC_DY + ckl] = dy;
      }
      put_rows();
      tail_sync();
      scan_step(1);
      if (iscl) {
        // A^T y (hxx:422) and the same at the world origin; and the two pieces of THIS iteration's force balance that are not
        // A^T y of the new dual: the constraint's share of H^base v + p^base is A^T dy + (the A^T y FwdPass1 used), which is
        // A^T y_new only when the instance arrived with A^T y consistent with its A (not after UpdateEqConstraint replaced A
        // under a warm start: upstream's first iteration then runs on the old product, hxx:329-331)
        const T* A_ = ccb + FC_A;
        const int k = ckl;
        T at = A_[k] * ccb[FC_Y], aw = ccb[FC_AW + 6 * k] * ccb[FC_Y], atd = A_[k] * ccb[FC_DY], awd = ccb[FC_AW + 6 * k] * ccb[FC_DY];
#pragma unroll
        for (int j = 1; j < 6; ++j) {
          at += A_[6 * j + k] * ccb[FC_Y + j]; aw += ccb[FC_AW + 6 * k + j] * ccb[FC_Y + j];
          atd += A_[6 * j + k] * ccb[FC_DY + j]; awd += ccb[FC_AW + 6 * k + j] * ccb[FC_DY + j];
        }
        ccb[FC_DLT + k] = (at - atd) - ccb[FC_ATY + k];   // A^T y_old - (A^T y used): added to g of the constrained joint
        ccb[FC_ATYF + k] = ccb[FC_ATYW + k] + awd;       // the constraint's force in f, world origin
        ccb[FC_ATY + k] = at;
        ccb[FC_ATYW + k] = aw;
      }
      // ---- per-joint work that needs v and nu only -- BoxProj, the w update, their norms, g (hxx:129-158, :384-397, :454-458) --
      // placed here so that it runs while the last exchange of the subtree sum is in flight
      auto part_v = [&]() {
        {
          T dv6[6], gi[6], dg[6], dvr[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            dv6[k] = vi[k] - v[k];
            // g_i = A^T y_i + sum_children act(f_c) - f_i = A^T y_i - (H^base_i v_i + p^base_i)  (force balance)
            gi[k] = -mass * (P.rho * dv6[k] + href_s * vi[k]);
          }
          if (has_hv) {
#pragma unroll
            for (int k = 0; k < 6; ++k) gi[k] += mass * P.Hv[k];
          }
          if (jcslot >= 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) gi[k] += cdi[jcslot * cs + FC_DLT + k];
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            dg[k] = gi[k] - g[k];
            dvr[k] = mass * (href_s * vi[k]) + gi[k];  // dual residual, v block (hxx:228): H_ref v - Hv + g
          }
          if (has_hv) {
#pragma unroll
            for (int k = 0; k < 6; ++k) dvr[k] -= mass * P.Hv[k];
          }
          l_dualv = inf6(dvr);
          l_nu = tabs(nui);
          l_hrefv = mass * tabs(href_s) * inf6(vi);
          l_dvis = mass * inf6(dv6);
          l_dnu = tabs(nui - nu);
          const T x = nui + (T(1) / mu_in) * w;
          const T zi = tmin(ubi, tmax(lbi, x));
          l_dz = tabs(zi - z);
          l_prs = tabs(nui - zi);
          const T dwi = mu_in * (nui - zi);
          l_dw = tabs(dwi);
          l_up += ubi * tmax(dwi, T(0));
          l_lm += lbi * tmin(dwi, T(0));
          w = w + dwi; z = zi; nu = nui;
          l_dg = inf6(dg);
          l_g = inf6(gi);
#pragma unroll
          for (int k = 0; k < 6; ++k) { v[k] = vi[k]; g[k] = gi[k]; }
        }
      };
      int kk = 2;
      for (; kk + 2 < nscan; ++kk) {
        put_rows();
        tail_sync();
        scan_step(kk);
      }
      if (kk >= nscan) {  // (a tree without a subtree of four joints: no exchange left to hide behind)
        tail_sync();
        part_v();
      } else {
        // the last two bits of `size` in one exchange: the window of width 2^(kk+1) is two windows of width 2^kk
        put_rows();
        tail_sync();
        const int wv = 1 << kk;
        const bool take0 = (size >> kk) & 1, take1 = (size >> (kk + 1)) & 1;
        const int p1 = pos + (take0 ? wv : 0);
        const int r0 = take0 ? pos : WAVE, r1 = take1 ? p1 : WAVE, r2 = take1 ? p1 + wv : WAVE;
        T a[6], b[6], c2[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) a[c] = xb[r0 * 6 + c];
#pragma unroll
        for (int c = 0; c < 6; ++c) b[c] = xb[r1 * 6 + c];
#pragma unroll
        for (int c = 0; c < 6; ++c) c2[c] = xb[r2 * 6 + c];
        part_v();
#pragma unroll
        for (int c = 0; c < 6; ++c) SEn[c] += a[c] + (b[c] + c2[c]);
      }
      tail_sync();
      TAIL_TP(4)
#pragma unroll
      for (int k = 0; k < 6; ++k) Fw[k] = (P.rho + href_s) * SEn[k] - P.rho * SE[k];
      if (has_hv) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Fw[k] -= shv[lane * 6 + k];
      }
      for (int c = 0; c < L.nc; ++c) {
        const T* c_ = cdi + c * cs;
        const T m = cmask(c);
#pragma unroll
        for (int k = 0; k < 6; ++k) Fw[k] += m * c_[FC_ATYF + k];
      }
      actinv_force(R0, t0, Fw, fi);
      si = dot6_halves(Sw, Fw);  // S^T f (hxx:231-233): the pairing of a motion and a force does not depend on the frame
#pragma unroll
      for (int k = 0; k < 6; ++k) SE[k] = SEn[k];
    }
    TAIL_TP(5)
    // ================= what is left of the per-joint work: the norms that need f (hxx:137-146, :231-236) =========================
    {
      T df[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) df[k] = fi[k] - f[k];
      l_dfis = mass * inf6(df);
      si += w;
      l_stf = tabs(si);
      l_dstf = tabs(si - s);
      s = si;
#pragma unroll
      for (int k = 0; k < 6; ++k) f[k] = fi[k];
    }
    TAIL_TP(3)
    // ================= the scalars of the stopping logic, folded over the group ================================================
    T red[8];
    {
      T in[8] = {tmax(l_prt, l_prs), tmax(l_dualv, l_stf), tmax(l_dvis, l_dnu), l_dz,
                 tmax(l_dfis, tmax(l_dyis, l_dw)), tmax(l_dg, l_dstf), l_up, l_lm};
      flat_fold8<T, 6>(xb, lane, gbase, jlane, lgG, in, red);
    }
    T ntol_p = T(0), ntol_d = T(0);
    if (P.tol_rel != T(0)) {  // (uniform) relative tolerances need two more maxima
      T in2[8] = {tmax(l_av, l_nu), tmax(tmax(l_hrefv, l_g), l_stf), T(0), T(0), T(0), T(0), T(0), T(0)}, r2[8];
      flat_fold8<T, 8>(xb, lane, gbase, jlane, lgG, in2, r2);
      ntol_p = r2[0]; ntol_d = r2[1];
    }
    TAIL_TP(6)
    // ================= CheckConvergence, CheckFeasibility, UpdateMu, the tail solve's stopping rule (hpp:377-454, :271-319) ====
    // without branches: every lane of the group evaluates the same scalars
    const T primal = red[0], dual = red[1], dx = red[2], dz = red[3], dyqp = red[4], atdy = red[5], ubp = red[6], lbm = red[7];
    const T mu_used = mu;
    const bool fixed = P.mode & MODE_FIXED_ITERS;
    const bool in_tail = (status & ST_TAIL) != 0;
    const bool logic = act && !fixed && !in_tail;  // the main loop's stopping logic runs
    const T tol_p = P.tol_abs + P.tol_rel * tmax(ntol_p, isc[FI_BNORM]);
    const T tol_d = P.tol_abs + P.tol_rel * tmax(ntol_d, P.Hv_inf_norm);
    const int itn = iter + (act ? 1 : 0);
    const bool conv = logic && (primal < tol_p) && (dual < tol_d);
    const bool feas_chk = logic && itn > 1;
    const bool c1 = atdy <= P.tol_primal_inf * dyqp, c2 = (ubp + lbm) <= P.tol_primal_inf * dyqp;
    const bool infeas = feas_chk && c1 && c2;
    const bool enter_tail = infeas && !conv;
    const bool upd = logic && !conv && !infeas;
    const bool mu_up = upd && (primal > T(10) * dual), mu_dn = upd && !mu_up && (dual > T(10) * primal);
    const bool tail_stop = !(dx >= P.tol_tail_solve || dz >= P.tol_tail_solve) || itn >= P.max_iter;
    const bool stop = act && (conv || ((enter_tail || in_tail) && tail_stop) || ((upd || fixed) && itn + 1 >= P.max_iter));
    iter = itn;
    status |= (conv ? ST_CONVERGED : 0) | (infeas ? ST_PRIMAL_INF : 0) | (enter_tail ? ST_TAIL : 0) | (stop ? ST_DONE : 0);
    tail_it = enter_tail ? 0 : (act && in_tail ? tail_it + 1 : tail_it);
    mu = mu_up ? mu * T(10) : (mu_dn ? mu * T(0.1) : mu);
    kexp += (mu_up ? 1 : 0) - (mu_dn ? 1 : 0);
    nflip += (mu_up || mu_dn) ? 1 : 0;
    done = done || stop;
    const bool finishing = stop;
    const bool leaving = done && has_inst;
    if (LOG) {
      T inl[8] = {l_prt, l_prs, l_stf, l_dualv, T(0), T(0), T(0), T(0)}, rl[8];
      flat_fold8<T, 8>(xb, lane, gbase, jlane, lgG, inl, rl);
      const int row = itn - 1;
      if (act && !in_tail && jlane == 0 && row < Bf.log_cap) {
        // (the nine lists of k_pass_solve, loik_passes.hpp: LOG_PR_TASK, LOG_PR_SLACK, LOG_PRIMAL, LOG_DUAL_NU, LOG_DUAL_V, LOG_DUAL,
        //  LOG_MU, LOG_MU_EQ, LOG_MU_INEQ)
        const double vals[9] = {(double)rl[0], (double)rl[1], (double)primal, (double)rl[2], (double)rl[3], (double)dual,
                                (double)mu_used, (double)(P.mu_scale * mu_used), (double)mu_used};
#pragma unroll
        for (int l = 0; l < 9; ++l) Bf.log[((size_t)l * Bf.log_B + lidx) * Bf.log_cap + row] = vals[l];
      }
    }
    // ---- what the getters report: written when an instance stops (the certificate's scalars also when it enters the tail
    // solve: they keep the values of the last iteration that evaluated them) -------------------------------------------------
    if (__any(finishing || enter_tail)) {
      if ((finishing || enter_tail) && jlane == 0) {
        isc[FI_PRIMAL] = primal; isc[FI_DUAL] = dual; isc[FI_DX] = dx; isc[FI_DZ] = dz; isc[FI_MULAST] = mu_used;
        if (logic) { isc[FI_TOLP] = tol_p; isc[FI_TOLD] = tol_d; }
        if (feas_chk) {
          isc[FI_C1] = (T)(c1 ? 1 : 0); isc[FI_C2] = (T)(c2 ? 1 : 0); isc[FI_DYQP] = dyqp; isc[FI_ATDY] = atdy; isc[FI_UBP] = ubp; isc[FI_LBM] = lbm;
        }
      }
    }
    if (__any(finishing)) {
      T in1[8] = {l_prt, l_prs, l_stf, l_dvis, l_dnu, l_dfis, l_dyis, l_dw}, in2[8] = {l_av, l_nu, l_hrefv, l_g, l_dualv, T(0), T(0), T(0)};
      T r1[8], r2[8];
      flat_fold8<T, 8>(xb, lane, gbase, jlane, lgG, in1, r1);
      flat_fold8<T, 8>(xb, lane, gbase, jlane, lgG, in2, r2);
      if (finishing && jlane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) isc[FI_RED + k] = r1[k];
#pragma unroll
        for (int k = 0; k < 5; ++k) isc[FI_RED + 8 + k] = r2[k];
      }
      tail_sync();
    }
    if (leaving) {
      store_instance();
      need_load = true;
    }
    TAIL_TP(7)
  }
#ifdef LOIKB_TAIL_PROF
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int k = 0; k < 8; ++k) g_tail_prof[k] = prof_[k];
    for (int k = 8; k < 12; ++k) g_tail_prof[2 + k] = prof_[k];
    g_tail_prof[8] = n_wave_iters;
    g_tail_prof[9] = (clock64() - clk0_) * 100000ull / (wall_clock64() - wall0_ + 1);
  }
#endif
  if (lane == 0) atomicAdd(&Bf.counters[5], n_wave_iters);
  if (jlane == 0) {
    atomicAdd(&Bf.counters[6], n_slot_loads >> 16);
    atomicAdd(&Bf.counters[FLAT_COUNTERS_SLOT_HITS], n_slot_hits);
    atomicOr(&Bf.counters[LEAN_DECADES_SEEN], n_slot_loads & 0xFFFFu);
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Decade slots of the flat engine: the joints' columns of W and Dinv of every listed instance for mu = mu0 * 10^(kexp_lo + s).
// Pass A: the H recursion (FwdPass1 + BwdPass, hxx:290-338, :31-81, H part) with every H_i expressed at the WORLD origin:
//   H^w_i = X*_{0<-i} H^base_i X*^T_{0<-i} + sum_children (H^w_c - UDinv^w_c U^w_cT),  U^w = H^w S^w,  Dinv = 1 / (S^w . U^w + mu)
// -- the congruence SE3actOn(liMi, .) that carries a child's H to its parent (hxx:66, ~250 instructions per joint and
// decade in k_hslots) is the identity between quantities at one origin; the two base terms (rho I + H_ref and A^T A of a
// constrained joint, both independent of mu) are carried to the origin once per instance.  Dinv is a scalar of the
// elimination, the same number in any frame.  The decades run as a pipeline over the tree levels (as in k_hslots); a joint
// that has its UDinv^w for a decade leaves it and Dinv in rows 0..6 of the instance's slot.
// Pass B (same wavefront, the rows come back from the L2): per decade L_{a,d} = S^w_a . UDinv^w_d for the ancestors a of every
// joint d go to LDS, and the joint inverts its column of the unit-triangular factor:
//   W_{a,d} = -(L_{a,d} + sum_{e strictly between a and d} L_{a,e} W_{e,d}),  nearest ancestor first;
// the slot's rows are overwritten with W (rows 0 .. nanc-1) and Dinv (row nanc).  frows = max(nanc + 1, 7) rows per slot:
// 10 scalars per joint and decade on Talos-32 where k_hslots wrote 22.
// ------------------------------------------------------------------------------------------------------------------------
template <typename T, int NA>
__global__ void __launch_bounds__(WAVE)
k_fslots(const Params<T> P, const Bufs<T> Bf, const JointDesc* __restrict__ jd, const TailTopo* __restrict__ topo,
         const int* __restrict__ child_list, const FlatLane* __restrict__ fl, int maxdepth, int nanc, int frows, int njmp,
         const int* __restrict__ slots, int nslots, int G, T* __restrict__ fslots, int kexp_lo, int ndec, int dgrp, int dw0, int nw,
         unsigned int* __restrict__ fmask)
{
  // dw0, nw: the decades to build NOW, [dw0, dw0 + nw) of the table's [0, ndec) -- the table is addressed for its whole range and
  // populated lazily: k_flat2<.., MUR = 2> builds the other decades of an instance in-wave if the instance ever gets there
  // (flat_build_slot, loik_flat2.hpp) and notes it in fmask (bit d: decade d is there; bit 16 + d: written during the launch, by a
  // wavefront of whatever XCD: coherent loads).
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Layout& L = P.L;
  constexpr int HX = 22;
  T* xch = reinterpret_cast<T*>(smem_raw);        // [WAVE + 1][22] placement rows [9], then the H^w a joint passes to its parent
  T* swt = xch + (WAVE + 1) * HX;                 // [WAVE + 1][6]  S^w of every lane's joint (+ a zero row)
  T* ata_l = swt + (WAVE + 1) * 6;                // [64/G][nc][21] A^T A of the instances' constraints at the world origin
  T* mutab = ata_l + (size_t)(WAVE / G) * P.L.nc * 21;   // [16] mu of the decades (flat_decade_mu: the ONE definition the in-wave builders share)
  T* lb = xch;                                    // [NA][WAVE] + WAVE: L columns of one decade (+ zeros) -- pass B, in the rows pass A is
                                                  // done with: 14.9 instead of 20.5 KB per wavefront, ten instead of seven per CU
  const int lane = threadIdx.x;
  const int ipw = WAVE / G;
  const int sub = lane / G, jlane = lane % G, gbase = sub * G;
  // Blocks b, b + 8, b + 16 .. run on the same XCD (observed, MI355X_MICROARCH.md; nothing depends on it but speed), and the 64
  // instances of a tile share every 1 KB row of their records: handed out in launch order, the 32 blocks of a tile sat on all eight
  // XCDs and each of the eight L2s fetched the tile's rows from HBM for its four blocks -- 21 KB fetched per instance where the records
  // hold 2.6.  Each XCD takes a contiguous eighth of the list instead.
  const int nblk = gridDim.x, xcd = blockIdx.x & 7, rnd = blockIdx.x >> 3;
  const int blk = xcd * (nblk >> 3) + (xcd < (nblk & 7) ? xcd : (nblk & 7)) + rnd;
  const int idx = blk * ipw + sub;
  const bool has_inst = idx < nslots;
  const bool isj_lane = jlane < L.nb;
  const bool isj = has_inst && isj_lane;
  const int jl = isj_lane ? jlane : 0;
  const int sidx = slots[has_inst ? idx : 0];
  char* ip = lane_ptr<T>(Bf.tiles, L, sidx);
  int depth, fcol, arow[NA];
  T Sw[6];
  bool has_parent;
  int cslot;
  T base0[21];  // mass * (rho I + H_ref) of this joint's link at the world origin
  {
    const JointDesc d = jd[jl + 1];
    const FlatLane F = fl[jlane];
    depth = isj_lane ? F.depth : 0;
    fcol = F.helper >> 8;
    has_parent = !(d.flags & JF_PARENT_ROOT);
    cslot = isj_lane ? d.cslot : -1;
    unsigned int jrow4[(FLAT_JMP + 3) / 4] = {0u, 0u};
#pragma unroll
    for (int r = 0; r < FLAT_JMP; ++r) jrow4[r >> 2] |= (unsigned int)(F.jmp[r] >= 0 ? gbase + F.jmp[r] : WAVE) << (8 * (r & 3));
#pragma unroll
    for (int k = 0; k < NA; ++k) arow[k] = (k < FLAT_MAXA && F.anc[k] >= 0) ? gbase + F.anc[k] : WAVE;
    const char* rec = ip + (size_t)jl * JREC * pair_bytes<T>();
    const typename Vec2<T>::type cs = ldp<T>(rec, JP_CS);
    T R0[9], t0[3];
    joint_xform<T>(d, rec, cs.x, cs.y, R0, t0);
    if (!isj_lane) {
#pragma unroll
      for (int k = 0; k < 9; ++k) R0[k] = (k % 4 == 0) ? T(1) : T(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) t0[k] = T(0);
    }
    flat_world_placement<T>(xch, lane, jlane, jrow4, njmp, R0, t0);
    const bool rev = d.flags & JF_REVOLUTE;
    const int jflags_h = d.flags;
    const T pitch_h = (d.flags & JF_HELICAL) ? (T)d.pitch : T(0);
    T ax[3], ra3[3], c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ax[k] = isj_lane ? (T)d.axis[k] : T(0);
    mat3_vec(R0, ax, ra3);
    cross3(t0, ra3, c);
#pragma unroll
    for (int k = 0; k < 3; ++k) { Sw[k] = rev ? c[k] : ra3[k]; Sw[3 + k] = rev ? ra3[k] : T(0); }
    if (jflags_h & JF_HELICAL) {  // S = [pitch a; a] at the world origin: (t x R a + pitch R a, R a)
#pragma unroll
      for (int k = 0; k < 3; ++k) Sw[k] += pitch_h * ra3[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) swt[lane * 6 + k] = Sw[k];
    if (lane < 6) swt[WAVE * 6 + lane] = T(0);
    // the base terms at the world origin
    {
      const T mass = (!isj_lane || (d.flags & JF_MASSLESS)) ? T(0) : T(1);
      T hb[21];
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b2 = a; b2 < 6; ++b2)  // (per-link references, UpdateReferences: the joint's row of the table)
          hb[sym(a, b2)] = mass * ((a == b2 ? P.rho : T(0)) + (P.href_tab ? P.href_tab[(size_t)(jl + 1) * HREF_ROW + 6 * a + b2] : P.Href[6 * a + b2]));
      congr_sym(R0, t0, hb, base0);
    }
    if (cslot >= 0) {
      const char* crec = ip + (size_t)(L.off_c + cslot * L.crec) * pair_bytes<T>();
      T at[21], atw[21];
      for (int k = 0; k < 21; ++k)
        at[k] = (P.mode & MODE_A_SHARED) ? Bf.uni[L.nc * 36 + cslot * 21 + k]
                                         : *reinterpret_cast<const T*>(crec + (size_t)(CP_ATA + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T));
      congr_sym(R0, t0, at, atw);
#pragma unroll
      for (int k = 0; k < 21; ++k) ata_l[(sub * L.nc + cslot) * 21 + k] = atw[k];
    }
  }
  const TailTopo tp = topo[jl + 1];
  // the rows of the first children stay in registers: the step loop must not wait for global memory (the child list)
  constexpr int NCH_REG = 3;
  int chl[NCH_REG];
#pragma unroll
  for (int c = 0; c < NCH_REG; ++c) chl[c] = (isj_lane && c < tp.nchild) ? gbase + child_list[tp.child_start + c] : WAVE;
  if (lane < ndec && lane < 16) mutab[lane] = (T)flat_decade_mu((double)P.mu0, kexp_lo + lane);
  tail_sync();
  if (lane < HX) xch[WAVE * HX + lane] = T(0);
  // The decades go through the two passes in groups of `dgrp` (LOIKB_FSLOT_DGRP, default: all at once): a group's rows of pass A
  // are read back by pass B while they are still in the L2 of the XCD (ten wavefronts x two instances x eight decades x 2 KB per CU
  // is 10 MB per XCD, its L2 has 4), at the price of refilling the pipeline over the tree levels once per group.
  if (fmask != nullptr && has_inst && jlane == 0) fmask[sidx] = (nw >= 32 ? 0xFFFFu : ((1u << nw) - 1u)) << dw0;
  for (int d0 = dw0; d0 < dw0 + nw; d0 += dgrp) {
  const int nd = (dw0 + nw - d0 < dgrp) ? dw0 + nw - d0 : dgrp;
  if (d0 > dw0) {
    tail_sync();
    if (lane < HX) xch[WAVE * HX + lane] = T(0);
  }
  // ---- pass A
  {
    const int lag = maxdepth - depth;
    tail_sync();
    for (int st = 0; st < maxdepth + nd - 1; ++st) {
      const int dsl = d0 + st - lag;
      const bool on = isj && depth > 0 && dsl >= d0 && dsl < d0 + nd;
      T hh[21];
#pragma unroll
      for (int k = 0; k < 21; ++k) hh[k] = base0[k];
      // (most joints of a robot have one child: the second and third rows are read only on the steps where a lane at work has them
      //  -- 21 LDS reads each for every lane of the wavefront otherwise, of the zero row)
      const bool any2 = __any(on && tp.nchild > 1), any3 = __any(on && tp.nchild > 2);
      if (on) {  // the children's contributions of the previous step (the same decade); a missing child is the zero row
#pragma unroll
        for (int c = 0; c < NCH_REG; ++c) {
          if ((c == 1 && !any2) || (c == 2 && !any3)) continue;
          const T* x = xch + chl[c] * HX;
#pragma unroll
          for (int k = 0; k < 21; ++k) hh[k] += x[k];
        }
        for (int c = NCH_REG; c < tp.nchild; ++c) {
          const T* x = xch + (gbase + child_list[tp.child_start + c]) * HX;
#pragma unroll
          for (int k = 0; k < 21; ++k) hh[k] += x[k];
        }
      }
      tail_sync();  // every lane has read its children's rows: they may be overwritten
      if (on) {
        const T mu = mutab[dsl];
        const T mu_eq = P.mu_scale * mu, mu_in = mu;
        if (cslot >= 0) {
          const T* at = ata_l + (sub * L.nc + cslot) * 21;
#pragma unroll
          for (int k = 0; k < 21; ++k) hh[k] += mu_eq * at[k];
        }
        T U[6], UD[6];
        symv(hh, Sw, U);
        const T dinv = T(1) / (dot6_halves(Sw, U) + mu_in);  // (S^T H S + mu: calc_aba with the armature mu, hxx:60-63)
#pragma unroll
        for (int k = 0; k < 6; ++k) UD[k] = U[k] * dinv;
#pragma unroll
        for (int k = 0; k < 6; ++k) fslots[fslotA_at(sidx, ndec, dsl, frows, G, k, jlane)] = UD[k];
        fslots[fslotA_at(sidx, ndec, dsl, frows, G, 6, jlane)] = dinv;
        if (has_parent) {
          T* x = xch + lane * HX;
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b2 = a; b2 < 6; ++b2) x[sym(a, b2)] = hh[sym(a, b2)] - UD[a] * U[b2];
        }
      }
      tail_sync();
    }
  }
  // ---- pass B (every lane reads back the rows it wrote itself)
  lb[NA * WAVE + lane] = T(0);
  __builtin_amdgcn_s_waitcnt(0);
  tail_sync();
  for (int dsl = d0; dsl < d0 + nd; ++dsl) {
    T UDw[6], dinv = T(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) UDw[k] = isj ? fslots[fslotA_at(sidx, ndec, dsl, frows, G, k, jlane)] : T(0);
    if (isj) dinv = fslots[fslotA_at(sidx, ndec, dsl, frows, G, 6, jlane)];
    T Lc[NA], Wc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      Lc[k] = dot6_halves(swt + arow[k] * 6, UDw);
      Wc[k] = T(0);
    }
    tail_sync();
#pragma unroll
    for (int k = 0; k < NA; ++k) lb[k * WAVE + lane] = Lc[k];
    tail_sync();
#pragma unroll
    for (int k = NA - 1; k >= 0; --k) {
      T acc = Lc[k];
#pragma unroll
      for (int k2 = k + 1; k2 < NA; ++k2) acc += lb[k * WAVE + arow[k2]] * Wc[k2];
      Wc[k] = (k < depth - 1) ? -acc : T(0);
    }
    tail_sync();  // (every lane of the instance has read its rows of this decade: the packed columns overwrite them)
    if (isj) {
      const size_t base = fslotW_at(sidx, ndec, dsl, frows, fcol);
#pragma unroll
      for (int k = 0; k < NA; ++k)
        if (k < depth - 1) fslots[base + k] = Wc[k];
      fslots[base + depth - 1] = dinv;
    }
  }
  }
}

// JointData::UDinv / Dinv and pis of the last executed iteration for instances solved by the flat engine (scalar record's
// SP_TAG == -2: that engine forms neither): one instance per thread, the H recursion of sweep_bwd<.., true, ..> at that
// iteration's mu; results into the record's JP_UD / JP_R / JP_P slots (p flagged ST_PFULL), the tag set to that mu.
template <typename T>
__global__ void k_rebuild_ud(char* tiles, Layout L, const JointDesc* __restrict__ jd, const T* __restrict__ uni, T rho, T mu_scale,
                             const T* __restrict__ href_tab, int a_shared, int B, T* __restrict__ scratch)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  char* lp = lane_ptr<T>(tiles, L, b);
  char* srec = lp + (size_t)L.off_s * pair_bytes<T>();
  if (*elem_ptr<T>(srec, SP_TAG, 0) != T(-2)) return;
  const T mu = *elem_ptr<T>(srec, SP_ST, 1);
  const T mu_eq = mu_scale * mu, mu_in = mu;
  T* o = scratch + (size_t)b * L.nb * 21;
  for (int i = 1; i <= L.nb; ++i) {
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c)
        o[(i - 1) * 21 + sym(r, c)] = (jd[i].flags & JF_MASSLESS) ? T(0) : ((r == c ? rho : T(0)) + href_tab[(size_t)i * HREF_ROW + 6 * r + c]);
    if (jd[i].cslot >= 0) {
      const int cs = jd[i].cslot;
      for (int k = 0; k < 21; ++k) {
        const T a = a_shared ? uni[L.nc * 36 + cs * 21 + k]
                             : *elem_ptr<T>(lp + (size_t)(L.off_c + cs * L.crec) * pair_bytes<T>(), CP_ATA + k / 2, k & 1);
        o[(i - 1) * 21 + k] += mu_eq * a;
      }
    }
  }
  for (int i = L.nb; i >= 1; --i) {
    const JointDesc d = jd[i];
    T hh[21], U[6], UD[6], R[9], t[3], part[21];
    for (int k = 0; k < 21; ++k) hh[k] = o[(i - 1) * 21 + k];
    const int a0 = (d.flags & JF_REVOLUTE) ? 3 : 0;
    const T ax0 = (T)d.axis[0], ax1 = (T)d.axis[1], ax2 = (T)d.axis[2];
    for (int k = 0; k < 6; ++k) U[k] = hh[sym(k, a0)] * ax0 + hh[sym(k, a0 + 1)] * ax1 + hh[sym(k, a0 + 2)] * ax2;
    T sus = ax0 * U[a0] + ax1 * U[a0 + 1] + ax2 * U[a0 + 2];
    if (d.flags & JF_HELICAL) {  // S = [pitch a; a]
      const T ph = (T)d.pitch;
      for (int k = 0; k < 6; ++k) U[k] += ph * (hh[sym(k, 0)] * ax0 + hh[sym(k, 1)] * ax1 + hh[sym(k, 2)] * ax2);
      sus = (ax0 * U[3] + ax1 * U[4] + ax2 * U[5]) + ph * (ax0 * U[0] + ax1 * U[1] + ax2 * U[2]);
    }
    const T dd = T(1) / (sus + mu_in);
    for (int k = 0; k < 6; ++k) UD[k] = U[k] * dd;
    char* rec = lp + (size_t)(i - 1) * JREC * pair_bytes<T>();
    st6<T>(rec, JP_UD, UD);
    st_hi<T>(rec, JP_R, dd);
    {
      // pis[i] of the last backward pass from f_i = H_i v_i + p_i (hxx:139-140) with the final iterates
      T vv[6], ff[6], hv[6];
      ld6<T>(rec, JP_V, vv);
      ld6<T>(rec, JP_F, ff);
      symv(hh, vv, hv);
      for (int k = 0; k < 6; ++k) ff[k] -= hv[k];
      st6<T>(rec, JP_P, ff);
    }
    if (d.parent == 0) continue;
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c) hh[sym(r, c)] -= UD[r] * U[c];
    const typename Vec2<T>::type cs = ldp<T>(rec, JP_CS);
    joint_xform<T>(d, rec, cs.x, cs.y, R, t);
    congr_sym(R, t, hh, part);
    for (int k = 0; k < 21; ++k) o[(d.parent - 1) * 21 + k] += part[k];
  }
  *elem_ptr<T>(srec, SP_TAG, 0) = mu;
  *elem_ptr<T>(srec, SP_ST, 0) = (T)((int)*elem_ptr<T>(srec, SP_ST, 0) | ST_PFULL);
}

}  // namespace loikb
