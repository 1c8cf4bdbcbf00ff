// loik_flat2.hpp -- the flat engine (loik_flat.hpp) with every 6-vector of the iteration SPLIT over two lanes: lane 32 h + j
// carries the linear (h = 0) or the angular (h = 1) half of the spatial vectors of joint j + 1, one instance per wavefront
// (the linear halves of all joints in lanes 0..31, the angular halves in lanes 32..63).
//
// Why: k_flat keeps ~360 registers per lane alive (the joint's placement, S^w, v, f, g, the subtree sum, the scan's rows in
// flight), which allows ONE wavefront per SIMD; its fp64 issue slots are two-thirds empty (dependent chains, LDS round trips)
// and a second wavefront that could fill them does not fit (the 256-register build spills ~100 values and is slower).  With
// the halves on two lanes a lane's state is 3-vectors: ~2/3 of the registers, two or three wavefronts per SIMD without
// scratch; the sums over subtrees / root paths move 3 scalars per lane instead of 6; and an instance alone on its SIMD (the
// 999-iteration instances that decide when a batch ends) runs a shorter iteration.  What does NOT halve is what couples the
// halves -- the cross products of the frame changes (R0^T (v - t0 x w), F_a - t0 x F_l) and everything scalar per joint
// (BoxProj, the dot products S . x, which cost one exchange between the halves of the wavefront: v_permlane32_swap) -- so a
// wavefront-iteration costs ~0.7 of k_flat's instructions for half as many instances.
// With the joints of an instance in depth-first order along the lanes of a half, two of k_flat's LDS phases become register
// work: the subtree sums are differences of a prefix sum (DPP row shifts + row broadcast; one LDS exchange fetches the prefix at
// the subtree's last joint) instead of four window-doubling exchanges, and the eight scalars of the stopping logic are reduced by
// a transpose-reduce over lane pairs / quads / rows (DPP, v_permlane16/32_swap) and leave in SGPRs (v_readlane) instead of
// three LDS round trips.  What k_flat2 waits for is LDS latency (about thirty dependent round trips per iteration in k_flat),
// not bandwidth: every exchange removed shortens the iteration of a lone instance AND frees the LDS pipe, which at two
// wavefronts per SIMD is what the wavefronts of a CU queue for.
//
// The arithmetic is k_flat's (same formulation at the world origin, same decade slots from k_fslots, same record format when an
// instance is stored: SP_TAG = -2), summed in a slightly different order where a sum is split between the two lanes of a joint.
// Loading an instance runs k_flat's full-width code on BOTH lanes of a joint (identical values, identical LDS rows); each lane
// then keeps its half.  Applies to robots of 17..32 joints (G = 32: smaller ones run in k_solve + k_tail, larger ones need more
// than one wavefront's lanes in this layout and stay in k_flat), at most FLAT_NA_SMALL ancestors per joint, fp64.
// Reference: /root/reference/include/loik/loik-loid-optimized.hxx (passes as cited in loik_flat.hpp).
#pragma once

#include <type_traits>

#include "loik_flat.hpp"

namespace loikb {

constexpr int F2G = 32;  // joints per instance (lane pairs)
constexpr int FLAT_PARKED = 1 << 30;  // ring entry: the instance is parked (k_flat2<.., SLICED>)
constexpr int FLAT_BUILD_REQ = 1 << 29;  // ... and the decade it needs (bits 24..27: kexp + 8) is not in the table: whoever unparks it builds the slot first
constexpr int FLAT2_PARK_ROWS = 33;   // rows of 64 doubles of a parked instance's lane state
constexpr int FLAT2_PARK_BATCH = 8;   // elements of the LDS blocks a lane moves per batch (512 per wavefront: up to three task constraints in one)
__host__ __device__ __forceinline__ int flat2_park_stride(int nc, bool has_hv)
{
  return (FLAT2_PARK_ROWS * WAVE + (has_hv ? 32 * 6 : 0) + nc * 164 + 64 + 8 + 1) & ~1;   // (+ the LDS blocks, FISC <= 64, six scalars)
}
constexpr int F2W = 32;  // row stride of the [ancestor][joint] arrays (W rows, W tau products): lane (j, h) touches row 2 i + h

// Constraint block of an instance in the LDS of k_flat2 / k_flat1 (doubles; every 6-vector starts at an even offset).  What the loop
// needs of a task constraint (A, b) on joint c is its image at the world origin: AW = X*_{0<-c} A^T ([world component k][row q],
// and transposed), because  A v_c = AW^T v^w_c  (the constrained link's velocity as the path sum leaves it, no frame change) and
// X* (A^T y) = AW y.  A itself stays for the stored record's A^T y.
enum : int { C2_LANE = 0, C2_B = 2, C2_Y = 8, C2_ATYW = 14 /* A^T y at the world origin: the next FwdPass1's */, C2_ATBW = 20,
             C2_ATYF = 26 /* the constraint's force in this iteration's f */, C2_CW = 32, C2_DLT = 38 /* first-iteration corrections */,
             C2_VC = 44 /* v^w of the constrained joint */, C2_ATY = 50 /* A^T y as the record brought it */, C2_AW = 56, C2_AWT = 92,
             C2_A = 128, C2D = 164 };

// v_max_f64 / v_min_f64 as the hardware does them.  __builtin_fmax on a value the compiler cannot prove quiet (anything that came
// through a DPP / permlane move, an LDS read, a fabs) is preceded by a canonicalising v_max_f64 x, x, x in IEEE mode: 93 of the
// 159 v_max_f64 of the round-3 loop.  The instruction itself already returns the other operand for a quiet NaN, which is fmax.
__device__ __forceinline__ double hmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double hmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double hmaxa(double a, double b) { double r; asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }   // max(|a|, |b|)
__device__ __forceinline__ double hmax_a(double a, double b) { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }    // max(a, |b|)
__device__ __forceinline__ double hinf3(const double* x) { return hmax_a(hmaxa(x[0], x[1]), x[2]); }
__device__ __forceinline__ double hinf6(const double* x) { return hmax(hinf3(x), hinf3(x + 3)); }

// ---- cross-lane helpers (registers only) --------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
__device__ __forceinline__ double dpp_f64(double old, double x)
{
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, ROW_MASK, BANK_MASK, BOUND);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, ROW_MASK, BANK_MASK, BOUND);
  return __hiloint2double(hi, lo);
}
// both halves of the wavefront's values of x: lo = what lane (lane & 31) holds, hi = what lane 32 + (lane & 31) holds
// (v_permlane32_swap: the upper half of one register changes places with the lower half of the other)
__device__ __forceinline__ void both_halves(double x, double& lo, double& hi)
{
  const int xl = __double2loint(x), xh = __double2hiint(x);
  const auto rl = __builtin_amdgcn_permlane32_swap(xl, xl, false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap(xh, xh, false, false);
  lo = __hiloint2double(rh[0], rl[0]);
  hi = __hiloint2double(rh[1], rl[1]);
}
// x of the linear lane + x of the angular lane of the joint (the same value, bit for bit, on both)
__device__ __forceinline__ double pair_sum(double x)
{
  double lo, hi;
  both_halves(x, lo, hi);
  return lo + hi;
}
// inclusive prefix sum along the lanes of each half of the wavefront (lanes 0..31 and 32..63 separately)
__device__ __forceinline__ double prefix32(double x)
{
  x += dpp_f64<0x111, 0xF, 0xF, true>(0.0, x);   // row_shr:1  (lanes without a source add 0)
  x += dpp_f64<0x112, 0xF, 0xF, true>(0.0, x);   // row_shr:2
  x += dpp_f64<0x114, 0xF, 0xF, true>(0.0, x);   // row_shr:4
  x += dpp_f64<0x118, 0xF, 0xF, true>(0.0, x);   // row_shr:8
  x += dpp_f64<0x142, 0xA, 0xF, false>(0.0, x);  // row_bcast:15 into rows 1 and 3: the total of the row before
  return x;
}
__device__ __forceinline__ double lane_read(double x, int src_lane)  // x of lane src_lane (ds_bpermute: the LDS crossbar, no memory)
{
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double uniform_of(double x, int src_lane)  // x of lane src_lane in SGPRs
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src_lane), __builtin_amdgcn_readlane(__double2loint(x), src_lane));
}
// Eight scalars reduced over the 64 lanes: columns 0..5 by max, 6 and 7 by sum when SUMS (else max), all eight results uniform.
// Transpose-reduce.  The first two exchanges trade REGISTERS between lane groups, which is exactly what v_permlane32_swap /
// v_permlane16_swap do (the upper half / the odd rows of one register change places with the lower half / the even rows of
// another): after swap(in[q], in[q + 4]) and one max, the lower half of the wavefront holds column q folded over the lane pairs
// (l, l + 32) and the upper half column q + 4 -- no selects; the same between the rows of a half with columns q, q + 2.  The
// third exchange (lanes 8 apart in a row) needs selects; a lane then owns ONE column (4 [half] + 2 [odd row] + [lane & 8]),
// folded over 8 lanes, and three butterflies inside its 8-lane group finish it.
template <bool SUMS>
__device__ __forceinline__ void wave_fold8(int lane, const double* in, double* out)
{
  const bool up = lane >= 32, b3 = lane & 8;
  const bool sm = SUMS && lane >= 48;  // columns 6, 7: the lanes of row 3
  auto comb = [&](double a, double b, bool is_sum) { return is_sum ? a + b : hmax(a, b); };
  double k4[4], k2[2], k1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const auto rl = __builtin_amdgcn_permlane32_swap(__double2loint(in[q]), __double2loint(in[q + 4]), false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(__double2hiint(in[q]), __double2hiint(in[q + 4]), false, false);
    k4[q] = comb(__hiloint2double(rh[0], rl[0]), __hiloint2double(rh[1], rl[1]), SUMS && q >= 2 && up);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const auto rl = __builtin_amdgcn_permlane16_swap(__double2loint(k4[q]), __double2loint(k4[q + 2]), false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(__double2hiint(k4[q]), __double2hiint(k4[q + 2]), false, false);
    k2[q] = comb(__hiloint2double(rh[0], rl[0]), __hiloint2double(rh[1], rl[1]), sm);
  }
  {
    const double send = b3 ? k2[0] : k2[1], keep = b3 ? k2[1] : k2[0];
    double recv = dpp_f64<0x108, 0xF, 0x3, false>(0.0, send);  // row_shl:8 into lanes 0-7 of a row
    recv = dpp_f64<0x118, 0xF, 0xC, false>(recv, send);        // row_shr:8 into lanes 8-15
    k1 = comb(keep, recv, sm);
  }
  {
    double recv = dpp_f64<0x104, 0xF, 0x5, false>(0.0, k1);    // row_shl:4 into lanes 0-3, 8-11
    recv = dpp_f64<0x114, 0xF, 0xA, false>(recv, k1);          // row_shr:4 into lanes 4-7, 12-15
    k1 = comb(k1, recv, sm);
  }
  k1 = comb(k1, dpp_f64<0x4E, 0xF, 0xF, true>(0.0, k1), sm);   // quad_perm [2, 3, 0, 1]
  k1 = comb(k1, dpp_f64<0xB1, 0xF, 0xF, true>(0.0, k1), sm);   // quad_perm [1, 0, 3, 2]
#pragma unroll
  for (int q = 0; q < 8; ++q) out[q] = uniform_of(k1, 32 * ((q >> 2) & 1) + 16 * ((q >> 1) & 1) + 8 * (q & 1));
}

// Four scalars reduced over the 64 lanes (column c by sum when bit c of SUMMASK is set, else by max), all four results uniform.
// The same transpose-reduce with half the registers in flight: swap(in[q], in[q + 2]) between the halves of the wavefront, then
// between the rows of a half -- row r of the wavefront now owns column r, folded over the four rows -- and four rotations inside
// the row of 16 lanes.  The iteration's stopping logic needs four maxima in the main loop (primal, dual and the two sides of the
// infeasibility certificate's first test) and the other four scalars only on the rare iterations that pass that test or run the
// tail solve: 7 exchanges per iteration instead of 17.
template <unsigned SUMMASK>
__device__ __forceinline__ void wave_fold4(int lane, const double* in, double* out)
{
  const bool up = lane >= 32, odd = (lane & 16) != 0;
  auto comb = [&](double a, double b, bool is_sum) { return is_sum ? a + b : hmax(a, b); };
  double k2[2], k1;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const auto rl = __builtin_amdgcn_permlane32_swap(__double2loint(in[q]), __double2loint(in[q + 2]), false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(__double2hiint(in[q]), __double2hiint(in[q + 2]), false, false);
    const bool sm = ((SUMMASK >> q) & 1u) && ((SUMMASK >> (q + 2)) & 1u) ? true
                    : ((SUMMASK >> q) & 1u) ? !up : ((SUMMASK >> (q + 2)) & 1u) ? up : false;
    k2[q] = comb(__hiloint2double(rh[0], rl[0]), __hiloint2double(rh[1], rl[1]), sm);
  }
  const bool smr = (SUMMASK >> ((up ? 2 : 0) + (odd ? 1 : 0))) & 1u;  // is this row's column a sum?
  {
    const auto rl = __builtin_amdgcn_permlane16_swap(__double2loint(k2[0]), __double2loint(k2[1]), false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(__double2hiint(k2[0]), __double2hiint(k2[1]), false, false);
    k1 = comb(__hiloint2double(rh[0], rl[0]), __hiloint2double(rh[1], rl[1]), SUMMASK == 0u ? false : SUMMASK == 0xFu ? true : smr);
  }
  const bool s4 = SUMMASK == 0u ? false : SUMMASK == 0xFu ? true : smr;
  k1 = comb(k1, dpp_f64<0x128, 0xF, 0xF, true>(0.0, k1), s4);   // row_ror:8
  k1 = comb(k1, dpp_f64<0x124, 0xF, 0xF, true>(0.0, k1), s4);   // row_ror:4
  k1 = comb(k1, dpp_f64<0x4E, 0xF, 0xF, true>(0.0, k1), s4);    // quad_perm [2, 3, 0, 1]
  k1 = comb(k1, dpp_f64<0xB1, 0xF, 0xF, true>(0.0, k1), s4);    // quad_perm [1, 0, 3, 2]
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = uniform_of(k1, 16 * q);
}

// ---- the stopping test's quick look, in single precision ---------------------------------------------------------------------
// The common iteration decides nothing (not converged, the certificate's first test fails, mu stays, not the last iteration), but
// finding that out took the four fp64 maxima of wave_fold4 -- six dependent v_max_f64 with two-register moves between them -- and
// the compares behind them: ~900 cycles of a lone wavefront's 5700 per iteration, with nothing else to issue.  Rounding to fp32 is
// monotone, so max_i RN(x_i) = RN(max_i x_i) exactly, and v_max_f32 takes its DPP move in the same instruction: the same four maxima
// in one register each and a tenth of the latency.  The fp32 maxima then decide the iteration only when they do so with room to
// spare (every compare is made against a threshold moved 1e-6 to the safe side, the fp32 rounding being 6e-8, and only on values in
// [1e-30, 1e30]); otherwise -- an iteration that converges, flags, changes mu or is the last, or one within 1e-6 of doing so -- the
// fp64 fold and the full logic run exactly as before.  Decisions and results are those of the fp64 test, bit for bit.
#ifndef LOIKB_QUIET32
#define LOIKB_QUIET32 0
#endif
#ifndef LOIKB_QUIET_SKIPS_TOP
#define LOIKB_QUIET_SKIPS_TOP 1
#endif
struct QuietF32 { unsigned int tol_hi; float tpi_hi; bool ok; };   // (tol_hi: the bits of a non-negative float)
__device__ __forceinline__ QuietF32 quiet_f32_thresholds(double tol_abs, double tol_rel, double tol_primal_inf)
{
  QuietF32 q;
  q.tol_hi = __float_as_uint((float)(tol_abs * 1.000001));
  q.tpi_hi = (float)(tol_primal_inf * 1.000001);
  q.ok = tol_rel == 0.0 && tol_abs >= 0.0 && tol_abs < 1e30 && tol_primal_inf > 1e-30 && tol_primal_inf < 1e30;
  return q;
}
__device__ __forceinline__ float hmaxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// in = {primal, dual, dyqp, atdy} of this lane; true: the iteration is certainly a quiet one
// iter < quiet_limit: the iteration after this one is neither the last of max_iter (the logic's `iter + 2 < max_iter`) nor beyond
// this launch's share (the loop top's `my_iters >= max_launch_iters`)
__device__ __forceinline__ int quiet_limit(int max_iter, int max_launch_iters, int iter_at_load)
{
  const int a = max_iter - 2, b = max_launch_iters > (1 << 29) ? 0x7fffffff : max_launch_iters + iter_at_load - 1;
  return a < b ? a : b;
}
__device__ __forceinline__ bool quiet_f32(const double* in, const QuietF32& th, int iter, int q_lim)
{
  float c[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) c[q] = (float)in[q];   // v_cvt_f32_f64, round to nearest: monotone
  float k2[2], k1;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(c[q]), __float_as_int(c[q + 2]), false, false);
    k2[q] = hmaxf(__int_as_float(r[0]), __int_as_float(r[1]));    // lower half: column q over the lane pairs (l, l + 32); upper: q + 2
  }
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(k2[0]), __float_as_int(k2[1]), false, false);
    k1 = hmaxf(__int_as_float(r[0]), __int_as_float(r[1]));       // row r of the wavefront: column r over the four rows
  }
  // four rotations inside the row, each one instruction (2 wait states between a VALU write and a DPP read of the register)
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "+v"(k1));
  // The compares are made on the BITS of the (non-negative) floats, as unsigned integers in the scalar unit -- the order is the
  // same, the thresholds are literals there and no vector register holds one over the loop; a NaN's bits are above 1e30's, so the
  // range checks send it to the fp64 test.  The three products are taken per lane before the values leave the vector registers.
  float t10, ty;
  asm("v_mul_f32 %0, 0x411ffff6, %1" : "=v"(t10) : "v"(k1));   // 10 (1 - 1e-6) x
  asm("v_mul_f32 %0, %1, %2" : "=v"(ty) : "s"(th.tpi_hi), "v"(k1));
  auto bits = [](float x, int l) { return (unsigned int)__builtin_amdgcn_readlane(__float_as_int(x), l); };
  const unsigned int p = bits(k1, 0), d = bits(k1, 16), y = bits(k1, 32), a = bits(k1, 48), p10 = bits(t10, 0), d10 = bits(t10, 16), tya = bits(ty, 32);
  const unsigned int lo = 0x0DA24260u /* 1e-30f */, hi = 0x7149F2CAu /* 1e30f */;
  const bool not_conv = (p > th.tol_hi) | (d > th.tol_hi);
  const bool no_cert = (iter == 0) | ((a > tya) & (a >= lo) & (y >= lo) & (a <= hi) & (y <= hi));
  const bool mu_stays = (p <= d10) & (d <= p10) & (p >= lo) & (d >= lo) & (p <= hi) & (d <= hi);
  return th.ok & not_conv & no_cert & mu_stays & (iter < q_lim);
}

template <typename T>
__device__ __forceinline__ T inf3(const T* x) { return tmax(tmax(tabs(x[0]), tabs(x[1])), tabs(x[2])); }

// A lane's gather addresses (which LDS entries its shares of a sum are) are constants of the launch, packed several to a register
// because registers are what the kernel is short of, and unpacked where they are used.  Round 3 unpacked with C shifts behind an
// opaque copy of the register (so that the compiler would not hoist thirty addresses out of the loop): v_mov + v_bfe + v_lshl_add
// per address, ~60 of the loop's VALU instructions.  Here the fields hold BYTE offsets and one v_bfe_u32, pinned where it stands,
// is the address.
#ifndef LOIKB_FIELD_TOKEN
#define LOIKB_FIELD_TOKEN 1
#endif
// (the same for a value that must not be hoisted: a copy the compiler cannot see through, tied to the iteration by the token)
__device__ __forceinline__ unsigned int opaque_here(unsigned int x, unsigned int tok)
{
#if LOIKB_FIELD_TOKEN
  asm("" : "+v"(x) : "s"(tok));
#else
  asm volatile("" : "+v"(x));
#endif
  return x;
}
template <int OFF, int W>
__device__ __forceinline__ unsigned int field_here(unsigned int x, unsigned int tok)
{
  unsigned int r;
#if LOIKB_FIELD_TOKEN
  // (not volatile: the scheduler may place it; the unused scalar operand changes every iteration, which is what keeps the thirty
  //  addresses from being hoisted out of the loop into registers the kernel does not have)
  asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(OFF), "n"(W), "s"(tok));
#else
  asm volatile("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(OFF), "n"(W));
#endif
  return r;
}
__device__ __forceinline__ double lds_at(const double* base, unsigned int byte_off)
{
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
}
// The same read by ABSOLUTE LDS address: BASE_BYTES (a constant of the instantiation, it ends up in the instruction's offset field) +
// byte_off.  Why: the dynamic LDS of these kernels starts at LDS address 0 (they have no static LDS), but the compiler learns that only
// when the module's LDS is laid out, after instruction selection -- every `base + variable offset` formed inside the loop kept its
// `v_add_u32 v, 0, v` (eighteen per iteration of k_flat2, 4 % of its vector instructions: the gathers of W tau, of the partial sums, of
// Dinv r' and the path sum's rounds).  An address built from the integer has nothing to add.  The kernels check the assumption once, at
// their start (lds_starts_at_zero), and tests/test_capi_abi.py reads `.group_segment_fixed_size: 0` from the shipped code object.
#ifndef LOIKB_LDS_ABS
#define LOIKB_LDS_ABS 1
#endif
typedef __attribute__((address_space(3))) const double lds_cdouble_t;
template <int BASE_BYTES>
__device__ __forceinline__ double lds_abs(unsigned int byte_off)
{
#if LOIKB_LDS_ABS
  return *reinterpret_cast<lds_cdouble_t*>(byte_off + (unsigned int)BASE_BYTES);
#else
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // (the A/B build: base + offset as the compiler forms it)
  return *reinterpret_cast<const double*>(smem_raw + BASE_BYTES + byte_off);
#endif
}
__device__ __forceinline__ bool lds_starts_at_zero(const void* dyn_lds)
{
  return (unsigned int)(size_t)(__attribute__((address_space(3))) const char*)dyn_lds == 0u;
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// y_i <- sum over the root path of lane i (the joint and its ancestors) of the NC-vectors y, by pointer jumping in base 4: a
// round adds the rows of the ancestors at distance 1, 2, 3 (x 4^round), so two rounds cover a tree of depth 16 where jumping in
// base 2 (flat_path_sum) takes four -- the same number of LDS rows moved, half the dependent round trips.  rows: [NC][PATH_RS],
// component-major (one address per ancestor, the components at constant offsets), entry WAVE of a component = 0; pa / pb / pc:
// the byte offsets (lane x 8, ten bits each) of the ancestors at distance 1, 2, 3 / 4, 8, 12 / 16 (WAVE x 8 = none).
constexpr int PATH_RS = WAVE + 2;
// PB2 (a build for joints with at most 11 ancestors): the second round has the ancestors at distance 4 and 8 only and there is no third
// round; the two offsets are the fields <20, 10> of pb and <22, 10> of pc -- bits the caller's other packed words have to spare (one
// register less across the loop), and the row of the ancestor at distance 12, which does not exist, is not read.
// `rows` MUST be the start of the kernel's dynamic LDS (xb at every call site): the gathers address it absolutely (lds_abs).
template <typename T, int NC, int PB2 = 0>
__device__ __forceinline__ void flat_path_sum4(T* rows, int lane, unsigned int pa, unsigned int pb, unsigned int pc, int njmp, T* y, unsigned int tok)
{
  tail_sync();
  if (lane < NC) rows[lane * PATH_RS + WAVE] = T(0);
  auto publish = [&]() {
    tail_sync();
#pragma unroll
    for (int c = 0; c < NC; ++c) rows[c * PATH_RS + lane] = y[c];
    tail_sync();
  };
  auto round = [&](unsigned int p3) {
    publish();
    const unsigned int r0 = field_here<0, 10>(p3, tok), r1 = field_here<10, 10>(p3, tok), r2 = field_here<20, 10>(p3, tok);
    T a[NC], b[NC], d[NC];
    static_for<0, NC>([&](auto c) { a[c] = lds_abs<c * PATH_RS * 8>(r0); });
    static_for<0, NC>([&](auto c) { b[c] = lds_abs<c * PATH_RS * 8>(r1); });
    static_for<0, NC>([&](auto c) { d[c] = lds_abs<c * PATH_RS * 8>(r2); });
#pragma unroll
    for (int c = 0; c < NC; ++c) y[c] += (a[c] + b[c]) + d[c];
  };
  round(pa);
  if constexpr (PB2) {
    if (njmp > 2) {
      publish();
      // (PB2 = 2, k_flat1: both in pb's upper half as LANES, eight bits each -- the word holds lane numbers)
      const unsigned int r0 = PB2 == 2 ? field_here<16, 8>(pb, tok) * 8u : field_here<20, 10>(pb, tok),
                         r1 = PB2 == 2 ? field_here<24, 8>(pb, tok) * 8u : field_here<22, 10>(pc, tok);
      T a[NC], b[NC];
      static_for<0, NC>([&](auto c) { a[c] = lds_abs<c * PATH_RS * 8>(r0); });
      static_for<0, NC>([&](auto c) { b[c] = lds_abs<c * PATH_RS * 8>(r1); });
#pragma unroll
      for (int c = 0; c < NC; ++c) y[c] += a[c] + b[c];   // (= (a + b) + 0: the same bits as the general round)
    }
  } else {
    if (njmp > 2) round(pb);
    if (njmp > 4) round(pc);
  }
}
// the packed ancestor offsets of flat_path_sum4 for the joint of FlatLane F: `off` is added to an ancestor's lane
__device__ __forceinline__ void flat_path_rows4(const FlatLane* fl, int j, int off, unsigned int& pa, unsigned int& pb, unsigned int& pc)
{
  const int depth = fl[j].depth;
  auto row = [&](int d) -> unsigned int {
    const int k = depth - d - 1;
    const int a = (k >= 0 && k < FLAT_MAXA) ? fl[j].anc[k] : -1;
    return (unsigned int)(a >= 0 ? a + off : WAVE) * 8u;
  };
  pa = row(1) | (row(2) << 10) | (row(3) << 20);
  pb = row(4) | (row(8) << 10) | (row(12) << 20);
  pc = row(16) | ((unsigned int)(WAVE * 8) << 10) | ((unsigned int)(WAVE * 8) << 20);
}

// ------------------------------------------------------------------------------------------------------------------------
// A decade slot of ONE instance built by the instance's own wavefront: k_fslots' two passes (loik_flat.hpp) for one value of mu, one
// joint per lane (lanes jl < G of the wavefront; `act` says which lanes take part), the same operations in the same order -- the
// columns are those k_fslots writes for that mu, bit for bit.  What it is for: (1) an instance whose mu leaves the decades k_fslots
// built for the launch builds the missing slot itself and carries on (it used to go back unfinished, to be finished by k_tail at
// 10 us per iteration: narrow tables cost more than they saved, so every launch built eight to ten decades of which an instance uses
// two or three); (2) a rule that takes mu off the decade grid (OSQP's, declared upstream at task-solver-base.hpp:13-18 and thrown at
// loik-loid-optimized.hxx:632-637) runs on the flat engine: every change of mu is one build (2.5 per solve on the headline batch).
// scr: (G + 1) x FB_HX exchange rows (pass B's L columns live in them afterwards), then (G + 1) x 6 rows of S^w.
// Math: FwdPass1 + BwdPass at the world origin, hxx:290-338, :31-81 (see k_fslots).
#ifndef LOIKB_BUILD_CALL
#define LOIKB_BUILD_CALL 0
#endif
constexpr int FB_HX = 22;
template <int G> __host__ __device__ constexpr int flat_build_scratch() { return (G + 1) * FB_HX + (G + 1) * 6; }
// hb: mass x (rho I + H_ref) of the lane's link, link frame, packed symmetric; at: A^T A of the constraint on the lane's joint (zeros
// where there is none) -- the caller reads them where k_fslots does (Params / the links' table; the constraint record or the batch's
// shared block): the builder itself touches neither the kernel's argument block nor the instance's records.
template <int NA, int G>
__device__ __forceinline__ void flat_build_slot(double* scr, int lane, bool act, int jl, int nb, const JointDesc* __restrict__ jd,
                                                const TailTopo* __restrict__ topo, const int* __restrict__ child_list,
                                                const FlatLane* __restrict__ fl, int maxdepth, const double* R0, const double* t0,
                                                const double* Sw, const double* hb, const double* at, double mu, double mu_scale,
                                                double* Wc, double& dinv_out, double* bcache = nullptr, bool cached = false)
{
  // bcache: [G][21] the joints' base terms at the world origin, then [nc][21] the constraints' A^T A there -- neither depends on mu: an
  // instance that builds more than once (OSQP's rule: 2.5 times per solve) carries them to the origin once (cached: they are there)
  using T = double;
  T* xch = scr;                         // [G + 1][FB_HX]
  T* swt = scr + (G + 1) * FB_HX;       // [G + 1][6]
  T* lb = scr;                          // [NA][G] + G: pass B
  const bool isj = act && jl < nb;
  const int jq = isj ? jl : 0;
  const JointDesc d = jd[jq + 1];
  const FlatLane F = fl[jq];
  const TailTopo tp = topo[jq + 1];
  const int depth = isj ? F.depth : 0;
  const bool has_parent = !(d.flags & JF_PARENT_ROOT);
  const int cslot = isj ? d.cslot : -1;
  int arow[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) arow[k] = (isj && k < FLAT_MAXA && F.anc[k] >= 0) ? F.anc[k] : G;
  constexpr int NCH_REG = 3;
  int chl[NCH_REG];
#pragma unroll
  for (int c = 0; c < NCH_REG; ++c) chl[c] = (isj && c < tp.nchild) ? child_list[tp.child_start + c] : G;
  T base0[21], atw[21];
#pragma unroll
  for (int k = 0; k < 21; ++k) atw[k] = T(0);
  if (bcache != nullptr && cached) {
    if (act) {
#pragma unroll
      for (int k = 0; k < 21; ++k) base0[k] = bcache[jl * 21 + k];
    }
    if (cslot >= 0) {
#pragma unroll
      for (int k = 0; k < 21; ++k) atw[k] = bcache[(G + cslot) * 21 + k];
    }
  } else {
    congr_sym(R0, t0, hb, base0);
    if (cslot >= 0) congr_sym(R0, t0, at, atw);
    if (bcache != nullptr) {
      if (act) {
#pragma unroll
        for (int k = 0; k < 21; ++k) bcache[jl * 21 + k] = base0[k];
      }
      if (cslot >= 0) {
#pragma unroll
        for (int k = 0; k < 21; ++k) bcache[(G + cslot) * 21 + k] = atw[k];
      }
    }
  }
  tail_sync();
  if (act) {
#pragma unroll
    for (int k = 0; k < 6; ++k) swt[jl * 6 + k] = Sw[k];
  }
  if (lane < 6) swt[G * 6 + lane] = T(0);
  if (lane < FB_HX) xch[G * FB_HX + lane] = T(0);
  // ---- pass A: leaves first, one tree level per step
  T UD[6], dinv = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) UD[k] = T(0);
  const int lag = maxdepth - depth;
  tail_sync();
  for (int st = 0; st < maxdepth; ++st) {
    const bool on = isj && depth > 0 && st == lag;
    T hh[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) hh[k] = base0[k];
    const bool any2 = __any(on && tp.nchild > 1), any3 = __any(on && tp.nchild > 2);
    if (on) {
#pragma unroll
      for (int c = 0; c < NCH_REG; ++c) {
        if ((c == 1 && !any2) || (c == 2 && !any3)) continue;
        const T* x = xch + chl[c] * FB_HX;
#pragma unroll
        for (int k = 0; k < 21; ++k) hh[k] += x[k];
      }
      for (int c = NCH_REG; c < tp.nchild; ++c) {
        const T* x = xch + child_list[tp.child_start + c] * FB_HX;
#pragma unroll
        for (int k = 0; k < 21; ++k) hh[k] += x[k];
      }
    }
    tail_sync();
    if (on) {
      const T mu_eq = mu_scale * mu, mu_in = mu;
      if (cslot >= 0) {
#pragma unroll
        for (int k = 0; k < 21; ++k) hh[k] += mu_eq * atw[k];
      }
      T U[6];
      symv(hh, Sw, U);
      dinv = T(1) / (dot6_halves(Sw, U) + mu_in);
#pragma unroll
      for (int k = 0; k < 6; ++k) UD[k] = U[k] * dinv;
      if (has_parent) {
        T* x = xch + jl * FB_HX;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b2 = a; b2 < 6; ++b2) x[sym(a, b2)] = hh[sym(a, b2)] - UD[a] * U[b2];
      }
    }
    tail_sync();
  }
  // ---- pass B: the joint's column of the unit-triangular factor's inverse
  for (int e = lane; e < G; e += WAVE) lb[NA * G + e] = T(0);
  T Lc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    Lc[k] = dot6_halves(swt + arow[k] * 6, UD);
    Wc[k] = T(0);
  }
  tail_sync();
  if (act) {
#pragma unroll
    for (int k = 0; k < NA; ++k) lb[k * G + jl] = Lc[k];
  }
  tail_sync();
#pragma unroll
  for (int k = NA - 1; k >= 0; --k) {
    T acc = Lc[k];
#pragma unroll
    for (int k2 = k + 1; k2 < NA; ++k2) acc += lb[k * G + arow[k2]] * Wc[k2];
    Wc[k] = (k < depth - 1) ? -acc : T(0);
  }
  dinv_out = dinv;
  tail_sync();
}

// The call the iteration kernels make: NOT inlined -- the builder's two hundred live values must not take part in the register
// allocation of the iteration loop (inlined, the loop of k_flat2 picked up three scratch reloads and two scalar loads per iteration) --
// and with the lane's inputs and outputs BY VALUE (an array handed over by pointer would live in scratch for the whole kernel).
struct FlatBuildIn { double R0[9], t0[3], Sw[6], hb[21], at[21], mu, mu_scale; };
template <int NA> struct FlatBuildOut { double Wc[NA], dinv; };
template <int NA, int G>
__device__ __noinline__ FlatBuildOut<NA> flat_build_slot_call(double* scr, int lane, bool act, int jl, int nb, const JointDesc* __restrict__ jd,
                                                              const TailTopo* __restrict__ topo, const int* __restrict__ child_list,
                                                              const FlatLane* __restrict__ fl, int maxdepth, const FlatBuildIn in)
{
  FlatBuildOut<NA> o;
  flat_build_slot<NA, G>(scr, lane, act, jl, nb, jd, topo, child_list, fl, maxdepth, in.R0, in.t0, in.Sw, in.hb, in.at, in.mu, in.mu_scale, o.Wc, o.dinv);
  return o;
}

// LDS of one wavefront of k_flat2<NA> (doubles): one instance
template <int NA>
__host__ __device__ constexpr int flat2_xregion()
{
  int n = XROWS * 9;                              // placement rows / full-width scan rows while an instance is loaded
  if (NA * F2W + 2 > n) n = NA * F2W + 2;         // W tau products (+ a zero slot)
  return (n + 1) & ~1;
}
// What the ITERATION keeps in the exchange region: the W tau products [NA][32] (+ a zero slot), the path rows [3][66], the prefix rows
// [64][3] (two sets with a non-scalar reference weight: [65][3] + [64][3]).  The rows an instance's LOAD needs (XROWS * 9: the oMi
// chain, the full-width scans) reach beyond that -- into the decade slots behind, which hold nothing while an instance is loaded
// (kslot / kslot_o are invalid from the moment the load starts; the first decade's columns are written at its end): 1.6 KB of LDS
// per wavefront that the third wavefront of a SIMD needs (12 wavefronts per CU: <= 13 653 bytes each).
template <int NA>
__host__ __device__ constexpr int flat2_xloop()
{
  int n = NA * F2W + 2;
  if ((WAVE + 1) * 3 + WAVE * 3 > n) n = (WAVE + 1) * 3 + WAVE * 3;
  if (3 * (WAVE + 2) > n) n = 3 * (WAVE + 2);
  return (n + 1) & ~1;
}
template <int NA> __host__ __device__ constexpr int flat2_off_wl() { return flat2_xloop<NA>(); }                            // [2][NA + 1][32]
template <int NA> __host__ __device__ constexpr int flat2_off_nbuf() { return flat2_off_wl<NA>() + 2 * (NA + 1) * F2W; }  // [34]
template <int NA> __host__ __device__ constexpr int flat2_off_pbuf() { return flat2_off_nbuf<NA>() + F2G + 2; }           // [34]
template <int NA> __host__ __device__ constexpr int flat2_off_rbuf() { return flat2_off_pbuf<NA>() + F2G + 2; }           // [32]
template <int NA> __host__ __device__ constexpr int flat2_off_jc() { return flat2_off_rbuf<NA>() + F2G; }                 // [32][3]: lb, ub, S^T f + w
template <int NA> __host__ __device__ constexpr int flat2_off_tail() { return flat2_off_jc<NA>() + F2G * 3; }
static_assert(flat2_xloop<10>() + 2 * 11 * F2W >= flat2_xregion<10>(), "the load-time rows must end inside the decade slots");

template <int NA>
__host__ __device__ __forceinline__ size_t flat2_lds_bytes(int nc, bool has_hv, bool build_cache = false)
{
  // (+ the null block; build_cache: the in-wave builder's mu-independent terms, [32 + nc][21] -- the OSQP build)
  const size_t n = (size_t)flat2_off_tail<NA>() + (has_hv ? (size_t)F2G * 6 : 0) + (size_t)nc * C2D + FISC + 36 + C2D + (build_cache ? (size_t)(F2G + nc) * 21 : 0);
  return (n * sizeof(double) + 15) & ~(size_t)15;
}

// A kernel argument the iteration uses, as a value of its own: the compiler re-fetches kernel arguments it has no scalar register for with
// s_load + s_waitcnt INSIDE the loop (the SLICED build, which has a few more live scalars, did that with the tolerances of the stopping
// test: 6 % of its iteration); a value that went through an asm is kept, or parked in a lane of a vector register (one v_readlane).
template <bool ON, typename X>
__device__ __forceinline__ X held(X x)
{
  if constexpr (ON) asm volatile("" : "+s"(x));
  return x;
}
// SLICED: round-robin time slicing inside the launch (the work queue of k_lean, loik_lean.hpp: a ring of instance slots with
// tickets on both sides).  An instance whose `quantum` iterations are used up while other instances wait for a wavefront is
// written back and goes to the BACK of the queue; the mutable part of its record then travels between wavefronts (possibly on
// different XCDs, whose L2s are not coherent with each other) and is accessed with agent-scope loads / stores.  Iteration counts
// are heavy-tailed and unknown in advance: run to completion in arrival order, the 999-iteration instances fetched late keep
// the launch alive for 3 ms after the queue ran dry (27 % of it); time-sliced, every long runner advances from the start.
// HM: the reference weight shared by the links.  0: H_ref = h I.  1: a DIAGONAL weight diag(d_1 .. d_6) (e.g. other weights on the
// angular than on the linear velocity).  2: a general symmetric 6x6 (read from LDS: 18 multiply-adds per lane).  3: one weight and
// one target PER LINK (UpdateReferences' table in HBM, Params::href_tab: the joint's 18 entries come through L1 / L2).  For 1..3
// H_ref v is no multiple of the link's velocity as a force at the world origin: the link velocities are weighted in the link
// frame, carried to the world origin and summed over the subtrees beside E -- three more prefix sums and one more frame change
// per iteration.
// LOG: the lists of LoikSolverInfo (loik-loid-optimized.hpp:47-127, filled at hpp:406-420), as k_flat<.., LOG> writes them: every
// iteration of the main loop folds the four scalars the lists need and stores its row -- no quiet iterations in this build.
// MUR: the rule that moves mu.  0: decade steps (DEFAULT / MAXEIGENVALUE, hxx:613-641), the decades' slots from k_fslots' table; an instance
// whose mu leaves the table goes back unfinished (k_tail finishes it).  2: the same rule, and a decade the table does not hold is built
// by the wavefront itself (flat_build_slot) -- a build of its own because the builder's presence costs the iteration loop a few scratch
// reloads (LOIKB_FLAT_BUILD=1 selects it).  1: OSQP's rule (update_mu, loik_device.hpp): mu is any number, there is no table, every change
// of mu is one in-wave build; every iteration folds the two norms the rule needs and walks the whole stopping logic (no quiet shortcut).
#ifndef LOIKB_TAU_AHEAD
#define LOIKB_TAU_AHEAD 1
#endif
#ifndef LOIKB_SPIN_SLEEP
#define LOIKB_SPIN_SLEEP 32   // (x 64 cycles between two looks of a wavefront that waits for a ring entry)
#endif
#ifndef LOIKB_HELD_ALWAYS
#define LOIKB_HELD_ALWAYS 0
#endif
constexpr int FLAT_COUNTERS_BUILT = 17;  // Bufs::counters[17]: decade slots built in-wave during the launch
// Two-launch schedule of a batch whose iteration counts nobody knows yet (round 6; k_flat2<.., SLICED>, flags in `quantum`):
//  FLAT_Q_PROBE  -- the PROBE launch: every listed instance runs `first slice + later slice` iterations at most; max(primal, dual) is noted
//    at three marks (the end of the first slice, FLAT_PROBE_LAST iterations before the probe's end, the probe's end); whoever is not done
//    by then is parked and its ring entry appended behind the fresh ones (ring[nslots ..): nobody takes it up in this launch).
//    k_probe_sort then orders the survivors by the iterations they are predicted to need still, longest first.
//  FLAT_Q_FINISH -- the launch that finishes them takes that list front to back, every instance to completion: the long runners start at
//    t = 0 of the launch and are never interrupted (the round robin of the single time-sliced launch serves the survivors of its first
//    slice in turns for as long as there are more of them than wavefronts).  The entries are parked records: an instance continues
//    exactly where the probe left it, so the schedule changes no bit of any result.
constexpr int FLAT_Q_PROBE = 1 << 30, FLAT_Q_FINISH = 1 << 29;
constexpr int FLAT_PROBE_LAST = 32;   // the probe's middle mark: this many iterations before its end
constexpr int FLAT_COUNTERS_FIN_N = 18;   // Bufs::counters[18]: entries of the list a FLAT_Q_FINISH launch takes (k_probe_sort)
constexpr int FI_MARK0 = FI_RED + 13, FI_MARK1 = FI_RED + 14, FI_MARKM = FI_RED + 15;   // (spare entries of the scalar block: max(primal, dual) at the marks; MARK1: the latest)
constexpr int FLAT_COUNTERS_DEC = 32;    // Bufs::counters[32 .. 63]: != 0: somebody took up (loaded or built) a slot of decade kexp + 16
template <int NA, int WPE, bool SLICED = false, int HM = 0, bool LOG = false, int MUR = 0>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
k_flat2(const Params<double> P, const Bufs<double> Bf, const JointDesc* __restrict__ jd, const FlatLane* __restrict__ fl, int nanc,
        int nscan, int njmp, int* ring, int nslots, const double* __restrict__ fslots, int frows, int kexp_lo,
        int ndec, double href_s, int has_hv_pk, int ring_mask, int quantum, double* __restrict__ park, int park_stride,
        const void* const* __restrict__ aux)
{
  // (what only the in-wave builder and the lazily populated table need comes through ONE pointer and the spare bits of has_hv: every
  //  kernel argument is a scalar register the iteration loop then lacks -- five more arguments cost the loop twenty v_readlane per iteration)
  // has_hv_pk: bit 0 = the reference target is not zero, bits 8..15 = the tree's depth, bits 16..31 = the decades k_fslots built for this
  // launch (win_bits); aux[0] = TailTopo*, aux[1] = the children's list, aux[2] = fmask
  const int has_hv = has_hv_pk & 1;
#define topo (reinterpret_cast<const TailTopo*>(aux[0]))
#define child_list (reinterpret_cast<const int*>(aux[1]))
#define fmask (reinterpret_cast<unsigned int*>(const_cast<void*>(aux[2])))
#define maxdepth ((has_hv_pk >> 8) & 0xFF)
#define win_bits ((has_hv_pk >> 16) & 0xFFFF)
  // fmask (MUR = 2): per instance, bit d: decade d of the table is there (k_fslots' window, win_bits, or built during the launch);
  // bit 16 + d: built during the launch, possibly by a wavefront of another XCD: fetched with agent-scope loads
  constexpr bool BUILD = MUR >= 1;
  // (the next iteration's tau formed beside this iteration's f: see `after_tau`.  Round 6, with the LDS accesses unmerged: it pays in the
  //  time-sliced builds only -- headline in arrival order 9.39 -> 8.50 ms with it -- and costs the plain build its lone iteration, 2.18 ->
  //  2.27 us, for nothing in the bulk (4 x batch 29.53 / 29.65 ms, ordered headline 7.48 / 7.51): profiles/r06_b_flat_switches_ab.txt.
  //  LOIKB_TAU_AHEAD=2: every build, as in round 5)
  constexpr bool TAU_AHEAD = (LOIKB_TAU_AHEAD != 0) && !LOG && (SLICED || MUR != 0 || LOIKB_TAU_AHEAD == 2);
  static_assert(flat_build_scratch<F2G>() <= flat2_off_nbuf<NA>(), "the in-wave builder's rows must end before the buffers the iteration keeps");
  using T = double;
  static_assert(NA % 2 == 0, "the W entries of a joint are dealt out to its two lanes");
  constexpr int G = F2G, GW = F2W, cs = C2D, NH = NA / 2;
  constexpr bool PB2 = NA <= 11 && NH % 3 != 0;   // (the path sum's second round from spare bits of anc3 / cb_pk: flat_path_sum4)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (!lds_starts_at_zero(smem_raw)) {   // (never: see lds_abs -- reported as an error, not computed wrongly)
    if (threadIdx.x == 0) atomicOr(Bf.counters + FLAT_COUNTERS_ERR, 4u);
    return;
  }
  const Layout& L = P.L;
  const bool a_shared = P.mode & MODE_A_SHARED;
  const int lane = threadIdx.x;
  const QuietF32 qth = quiet_f32_thresholds(P.tol_abs, P.tol_rel, P.tol_primal_inf);
  // (iterations per time slice; none: never ends.  quantum = first | later << 16: an instance's FIRST slice and its later ones -- the later
  //  ones shorter, so that the instances still running when the queue is empty have done the same number of iterations to within that)
  const int slice_len = (SLICED && (quantum & 0xffff) > 0) ? (quantum & 0xffff) : 0x3fffffff;
  const int slice_len2 = (SLICED && ((quantum >> 16) & 0x1fff) > 0) ? ((quantum >> 16) & 0x1fff) : slice_len;
  // (LOIKB_HELD_ALWAYS=1 -- the plain build too, whose quiet test re-fetches them since round 5 -- measured: lone 2.278 -> 2.325 us, bulk +0.5 %: not adopted)
  const double tol_abs_h = held<LOIKB_HELD_ALWAYS || SLICED || WPE >= 3 || MUR >= 1>(P.tol_abs), tpi_h = held<LOIKB_HELD_ALWAYS || SLICED || WPE >= 3 || MUR >= 1>(P.tol_primal_inf);
  const int max_iter_h = held<LOIKB_HELD_ALWAYS || SLICED || WPE >= 3 || MUR >= 1>(P.max_iter);
  const int j = lane & 31;       // lanes j and 32 + j <-> device joint j + 1
  const bool h = lane >= 32;     // 0: linear halves, 1: angular halves
  const int h3 = h ? 3 : 0;
  // Three wavefronts per SIMD leave a lane 168 registers: what an iteration touches once -- the joint's box, last iteration's
  // S^T f + w -- then lives in LDS (one row of three per joint; both lanes of a joint read the same address).
  constexpr bool JCL = WPE >= 3 || MUR == 1 || (MUR == 2 && !SLICED);   // (the builds with the in-wave builder: the loop has no register to spare for it)
  // ---- LDS of the wavefront
  T* const xb = reinterpret_cast<T*>(smem_raw);   // load-time rows | path rows [65][3] | W tau products [NA][32]
  T* const wl = xb + flat2_off_wl<NA>();          // [2][NA + 1][32]  W rows and the Dinv row of two decades of mu
  T* const nbuf = xb + flat2_off_nbuf<NA>();      // [34]             Dinv r' of every joint (+ zeros)
  T* const pbuf = xb + flat2_off_pbuf<NA>();      // [34]             partial sums of long rows
  T* const rbuf = xb + flat2_off_rbuf<NA>();      // [32]             r' of the last iteration (stored with the instance)
  T* const jc = xb + flat2_off_jc<NA>();          // [32][3]          lb, ub, S^T f + w of the joints (the builds for three wavefronts per SIMD)
  T* const shv = xb + flat2_off_tail<NA>();       // [32][6]          subtree sums of the links' H_ref v_ref (if != 0)
  T* const cdi = shv + (has_hv ? G * 6 : 0);      // [nc][C2D]        constraint blocks of the instance
  T* const isc = cdi + (size_t)L.nc * cs;         // [FISC]

  const bool isj_lane = j < L.nb;
  const int jl = isj_lane ? j : 0;
  const int jflags = jd[jl + 1].flags, jcslot = isj_lane ? jd[jl + 1].cslot : -1;
  const T mass = (!isj_lane || (jflags & JF_MASSLESS)) ? T(0) : T(1);
  const T hz = h ? T(0) : T(1);  // scalar-per-joint contributions to sums come from the linear lane only
  const T hh = T(1) - hz, mha = mass * hh;  // (1 on the angular lanes; the halves' formulas differ by terms multiplied by hz / hh:
                                            //  a select between two doubles is two v_cndmask, a 0 / 1 factor rides in an FMA)
  constexpr bool HD = HM > 0;  // (H_ref v needs its own subtree sums)
  T hd[3];  // this half's diagonal entries of H_ref (HM = 1; HM = 0 multiplies by the uniform href_s)
#pragma unroll
  for (int k = 0; k < 3; ++k) hd[k] = HM == 1 ? (h ? P.Href[7 * (3 + k)] : P.Href[7 * k]) : T(0);
  T* const hmat = isc + FISC;  // [36] H_ref (HM = 2)
  if (HM == 2 && lane < 36) hmat[lane] = P.Href[lane];
  for (int e = lane; e < cs; e += WAVE) hmat[36 + e] = T(0);   // the null constraint block
  T* const bcache = MUR == 1 ? hmat + 36 + cs : nullptr;   // [32 + nc][21]: flat_build_slot's terms that do not depend on mu (one instance)
  bool bc_valid = false;
  const T* const hrow = HM == 3 ? P.href_tab + (size_t)(jl + 1) * HREF_ROW : nullptr;  // (H_ref_i, H_ref_i v_ref_i) of this link
  auto hvl3_of = [&](int k) -> T { return HM == 3 ? hrow[36 + h3 + k] : (h ? P.Hv[3 + k] : P.Hv[k]); };  // this half of H_ref v_ref of the link
  int size, fcol, fdm1;  // (fcol, fdm1: this joint's column in a packed decade slot, its number of ancestors)
  bool helper;
  unsigned int pb2_d4 = 0u, pb2_d8 = 0u;
  unsigned int pathA, pathB, pathC;        // iteration: the ancestors at distance 1, 2, 3 / 4, 8, 12 / 16: byte offsets of their lanes of this half
  unsigned int ra2[2], part2[2], anc3[(NH + 2) / 3];  // byte offsets into the product / partial / Dinv r' buffers (16 / 16 / 10 bits each)
  {
    const FlatLane F = fl[j];
    size = isj_lane ? F.size : 0;
    helper = (F.helper & 1) != 0;
    fcol = F.helper >> 8; fdm1 = F.depth > 0 ? F.depth - 1 : 0;
    flat_path_rows4(fl, j, h ? 32 : 0, pathA, pathB, pathC);
    if constexpr (PB2) {   // (distance 4 into the last word of anc3, distance 8 into cb_pk: see flat_path_sum4)
      pb2_d4 = pathB & 0x3FFu;
      pb2_d8 = (pathB >> 10) & 0x3FFu;
    }
#pragma unroll
    for (int k = 0; k < (NH + 2) / 3; ++k) anc3[k] = 0u;
    // this lane's half of the joint's share of the W tau products, of its partials and of its W entries (k = 2 i + h)
    ra2[0] = ra2[1] = 0u;
    part2[0] = part2[1] = 0u;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = h ? F.red[4 + t] : F.red[t];  // (entry k * 32 + lane' of the product buffer: the schedule was built for G = 32)
      ra2[t >> 1] |= (unsigned int)((e >= 0 ? (e >> 5) * GW + (e & 31) : flat2_off_nbuf<NA>() + G) * 8) << (16 * (t & 1));  // (none: nbuf's zero pad)
      const int p = h ? F.part[4 + t] : F.part[t];
      part2[t >> 1] |= (unsigned int)((p >= 0 ? p : G) * 8) << (16 * (t & 1));
    }
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int a0 = F.anc[2 * i], a1 = (2 * i + 1 < FLAT_MAXA) ? F.anc[2 * i + 1] : -1;
      const int a = h ? a1 : a0;
      anc3[i / 3] |= (unsigned int)((a >= 0 ? a : G) * 8) << (10 * (i % 3));
    }
    if constexpr (PB2) anc3[(NH + 2) / 3 - 1] |= pb2_d4 << 20;
  }
  if (lane < 2) { nbuf[G + lane] = T(0); pbuf[G + lane] = T(0); }
  for (int e = lane; e < 2 * (NA + 1) * GW; e += WAVE) wl[e] = T(0);

  // ---- the instance of this wavefront
  bool has_inst = false, isj = false, done = true, any_iter = false;
  int lidx = 0;
  char *ip = Bf.tiles, *rec = Bf.tiles;
  T R0[9], t0[3], Sw3[3], v3[3], f3[3], g3[3], SE3[3];
  T w = T(0), z = T(0), nu = T(0), s = T(0), lbi = T(0), ubi = T(0), mu = T(1);
  int kexp = 0, kslot = -(1 << 30), kslot_o = -(1 << 30), wsel = 0;
  int iter = 0, status = ST_DONE, tail_it = 0, nflip = 0;
  unsigned int my_iters = 0, n_wave_iters = 0, n_slot_loads = 0, n_slot_hits = 0, n_inst_iters = 0, n_requeues = 0, dec_seen = 0u;
  unsigned int* q_head = Bf.counters + LEAN_Q_HEAD;
  unsigned int cbits = 0u;  // constraint c: is its joint in this joint's subtree?
  auto cmask = [&](int c) -> T { return ((cbits >> c) & 1u) ? T(1) : T(0); };
  // the DualUpdate's lanes: lane 6 c + k owns row k of constraint c.  The loop keeps TWO numbers of that (the block's byte offset and
  // 8 k) and forms every address from them where it is used (immediate offsets of the LDS instructions): the five row / vector
  // pointers the compiler would otherwise carry across the loop are what it spilled -- three scratch reloads per iteration in the
  // time-sliced build, each an s_waitcnt vmcnt(0) that is cheap while the line sits in the L1 and a trip to the L2 once park
  // records stream through it (iterations 40 % slower under time slices of 64).
  // The lanes that own no row work on a NULL block (zeros throughout, behind the instance's blocks): the two steps of the update
  // then run without a branch around them, i.e. in the basic block of the subtree prefix sums, and the scheduler fills the update's
  // LDS round trips and dependent multiply-adds with the prefix sums' DPP chains.
  const bool iscl = lane < 6 * L.nc;
  // (one register: the block's byte offset in the low 16 bits, 8 k above them)
  const unsigned int cb_pk = (unsigned int)((iscl ? (lane / 6) * cs : L.nc * cs + FISC + 36) * 8) | ((unsigned int)((lane % 6) * 8) << 16) | (PB2 ? pb2_d8 << 22 : 0u);
  // (AW y)_k and (A^T y)_k of the constraint of lane 6 c + k: ONE order of operations wherever they are formed (load, loop, store),
  // so that an instance resumed from the queue continues with the bits it would have had
  auto awy_of = [&](const char* blk, unsigned int k8) -> T {
    const T* r = reinterpret_cast<const T*>(blk + C2_AW * 8 + 6 * k8);
    const T* y = reinterpret_cast<const T*>(blk + C2_Y * 8);
    // (two chains of three: a dependent fp64 multiply-add costs a lone wavefront 32 cycles, the six in a row 190)
    return ((r[0] * y[0] + r[1] * y[1]) + r[2] * y[2]) + ((r[3] * y[3] + r[4] * y[4]) + r[5] * y[5]);
  };
  auto aty_of = [&](const char* blk, unsigned int k8) -> T {
    const T* A_ = reinterpret_cast<const T*>(blk + C2_A * 8 + k8);
    const T* y = reinterpret_cast<const T*>(blk + C2_Y * 8);
    return ((A_[0] * y[0] + A_[6] * y[1]) + A_[12] * y[2]) + ((A_[18] * y[3] + A_[24] * y[4]) + A_[30] * y[5]);
  };
  bool resumed = false;  // (SLICED) the instance came back from the queue: no first-iteration corrections
  unsigned int fmask_r = 0xFFFFu;   // (MUR = 2) this instance's word of fmask, uniform
  T bnorm_r = T(0);      // (MUR = 1) max |b| of the instance's constraints (bis_inf_norm_): OSQP's rule normalises the primal residual with it
  auto half = [&](const T* x6, T* x3) {
#pragma unroll
    for (int k = 0; k < 3; ++k) x3[k] = h ? x6[3 + k] : x6[k];
  };
  auto whole = [&](const T* x3, T* x6) {  // both halves of a split vector (one exchange between the halves of the wavefront)
#pragma unroll
    for (int k = 0; k < 3; ++k) both_halves(x3[k], x6[k], x6[3 + k]);
  };

  auto note_dry = [&]() {  // the first wavefront to find the queue empty notes the time (the launch's bulk phase ends here)
    if (lane == 0 && atomicCAS(Bf.counters + FLAT_COUNTERS_DRY, 0u, 1u) == 0u)
      __hip_atomic_store(Bf.counters + FLAT_COUNTERS_TDRY, (unsigned int)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // ---- work queue.  Plain: the listed instances in order, one atomic per fetch.  SLICED: the ring of k_lean (ticket t is served by
  // entry t; entries are consumed by resetting them to -1; LEAN_Q_RETIRED counts the instances that will not come back)
  unsigned int* q_tail = Bf.counters + LEAN_Q_TAIL;
  unsigned int* q_retired = Bf.counters + LEAN_Q_RETIRED;
  auto fetch = [&]() -> int {
    if constexpr (!SLICED) {
      int nx = 0;
      if (lane == 0) nx = (int)atomicAdd(q_head, 1u);
      nx = __builtin_amdgcn_readfirstlane(nx);
      if (nx >= nslots) note_dry();
      return nx < nslots ? ring[nx] : -1;
    } else {
      int got = -1;
      if (lane == 0 && (quantum & FLAT_Q_FINISH)) {
        // the sorted list front to back, one atomic per fetch
        const unsigned int n2 = __hip_atomic_load(Bf.counters + FLAT_COUNTERS_FIN_N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int at = atomicAdd(q_head, 1u);
        if (at < n2) got = __hip_atomic_load(ring + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (atomicCAS(Bf.counters + FLAT_COUNTERS_DRY, 0u, 1u) == 0u)
          __hip_atomic_store(Bf.counters + FLAT_COUNTERS_TDRY, (unsigned int)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (lane == 0) {
        const unsigned int ticket = atomicAdd(q_head, 1u);
        int* e = ring + (ticket & (unsigned int)ring_mask);
        bool noted = false;
        for (unsigned int spins = 0;; ++spins) {
          if ((quantum & FLAT_Q_PROBE) && ticket >= (unsigned int)nslots) {   // (the probe launch hands out the fresh entries only)
            if (atomicCAS(Bf.counters + FLAT_COUNTERS_DRY, 0u, 1u) == 0u)
              __hip_atomic_store(Bf.counters + FLAT_COUNTERS_TDRY, (unsigned int)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          const int v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v >= 0) { __hip_atomic_store(e, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); got = v; break; }
          if (__hip_atomic_load(q_retired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned int)nslots) break;
          if (!noted) {  // (this ticket is beyond the tail: from now on wavefronts wait for entries)
            noted = true;
            if (atomicCAS(Bf.counters + FLAT_COUNTERS_DRY, 0u, 1u) == 0u)
              __hip_atomic_store(Bf.counters + FLAT_COUNTERS_TDRY, (unsigned int)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (spins > (1u << 23)) { atomicOr(Bf.counters + FLAT_COUNTERS_ERR, 1u); break; }  // (never: a lost entry must not hang the GPU)
          __builtin_amdgcn_s_sleep(LOIKB_SPIN_SLEEP);
        }
      }
      return __builtin_amdgcn_readfirstlane(got);
    }
  };
#ifdef LOIKB_TAIL_PROF
  unsigned long long prof_[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = clock64();
  const unsigned long long wall0_ = wall_clock64(), clk0_ = tprev_;
#endif
  // ---- SLICED: an instance whose time slice is used up is PARKED -- the lane state and the instance's LDS blocks, as they are, in a
  // lane-contiguous record of its own (33 rows of 64 doubles + the LDS blocks: every access a full 512-byte row) -- and taken up
  // again by whichever wavefront pops its entry, possibly on another XCD (agent-scope accesses; the pusher waits for its stores
  // before the entry appears).  Round 3 sent it through the tile records (410 scattered stores, 90 loads, the whole set-up again:
  // ~130 us of a wavefront per switch, more than the slices brought); a park / unpark pair is ~40 row accesses each way.
  // (MUR = 2) a slot built in-wave goes into the table, where k_fslots would have put it: the instance's later visits of the decade
  // load it like any other (a long runner flips between two decades hundreds of times: rebuilt at every visit, the 999-iteration
  // instances spent 7 ms of their launch building -- measured)
  auto publish_slot = [&](int kx, const T* Wc, T dv) {
    const int dsl = kx - kexp_lo;
    if (dsl < 0 || dsl >= ndec || dsl >= 16) return;
    if (!h && isj_lane) {
      T* base = const_cast<T*>(fslots) + fslotW_at(lidx, ndec, dsl, frows, fcol);
#pragma unroll
      for (int k = 0; k < NA; ++k)
        if (k < fdm1) cst<T>(reinterpret_cast<char*>(base + k), Wc[k]);
      cst<T>(reinterpret_cast<char*>(base + fdm1), dv);
    }
    const unsigned int bits = (1u << dsl) | (1u << (16 + dsl));
    fmask_r |= bits;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the column is out before the bit says so
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_fetch_or(fmask + lidx, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto park_instance = [&]() {
    T* pk = park + (size_t)lidx * park_stride;
    int r = 0;
    auto put = [&](T x) { cst<T>(reinterpret_cast<char*>(pk + (r++) * WAVE + lane), x); };
#pragma unroll
    for (int k = 0; k < 9; ++k) put(R0[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) put(t0[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) put(Sw3[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) put(v3[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) put(f3[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) put(g3[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) put(SE3[k]);
    if constexpr (JCL) { put(w); put(z); put(nu); put(jc[j * 3 + 2]); put(jc[j * 3]); put(jc[j * 3 + 1]); }
    else { put(w); put(z); put(nu); put(s); put(lbi); put(ubi); }
    T* pl = pk + FLAT2_PARK_ROWS * WAVE;
    const int nl = (has_hv ? G * 6 : 0) + L.nc * cs + FISC;   // shv | constraint blocks | the getters' scalars: contiguous in LDS
    tail_sync();
    // (every read of a batch before its stores: one LDS latency per batch, not per element)
    for (int e0 = 0; e0 < nl; e0 += FLAT2_PARK_BATCH * WAVE) {
      T lb[FLAT2_PARK_BATCH];
#pragma unroll
      for (int k = 0; k < FLAT2_PARK_BATCH; ++k) { const int e = e0 + k * WAVE + lane; lb[k] = e < nl ? shv[e] : T(0); }
#pragma unroll
      for (int k = 0; k < FLAT2_PARK_BATCH; ++k) { const int e = e0 + k * WAVE + lane; if (e < nl) cst<T>(reinterpret_cast<char*>(pl + e), lb[k]); }
    }
    if (lane < 6) {
      const T x = lane == 0 ? mu : lane == 1 ? (T)kexp : lane == 2 ? (T)iter : lane == 3 ? (T)status : lane == 4 ? (T)tail_it : (T)nflip;
      cst<T>(reinterpret_cast<char*>(pl + nl + lane), x);
    }
    n_inst_iters += my_iters;
  };
  // (ALL loads of the record in flight together -- the lane rows, the LDS blocks' elements, the six scalars behind them: one round trip
  //  to HBM.  The first version read the LDS blocks and the scalars element by element, ten dependent round trips: 60 us per switch)
  auto unpark = [&](int entry) {
    const int slot = entry & 0xFFFFF, dsl_raw = (entry >> 24) & 15;  // (the decade of mu the instance was parked in travels in the entry:
    lidx = slot;                                                     //  its W columns are fetched WITH the record, not a round trip later)
    isj = isj_lane;
    ip = lane_ptr<T>(Bf.tiles, L, slot);
    rec = ip + (size_t)jl * JREC * pair_bytes<T>();
    const T* pk = park + (size_t)slot * park_stride;
    // (SLICED, MUR = 2) the instance left the iteration loop because its decade is not in the table: the slot is built HERE, from the
    // three lane rows the builder needs and before anything else of the instance is in registers -- nothing of the iteration's state is
    // live across the builder, so its two hundred values do not touch the register allocation of the loop (built in place, beside the
    // loop, they cost it four scratch reloads per iteration: 3-6 % of the headline)
    bool built = false;
    int kexp_b = 0;
    if constexpr (MUR == 2) fmask_r = (unsigned int)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(fmask + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if constexpr (SLICED && MUR == 2) {
      if (entry & FLAT_BUILD_REQ) {
        kexp_b = dsl_raw - 8;
        T Rb[9], tb[3], Sb3[3], Sw6[6], hb[21], at[21], Wc[NA], dv;
#pragma unroll
        for (int k = 0; k < 9; ++k) Rb[k] = cld<T>(reinterpret_cast<const char*>(pk + k * WAVE + lane));
#pragma unroll
        for (int k = 0; k < 3; ++k) tb[k] = cld<T>(reinterpret_cast<const char*>(pk + (9 + k) * WAVE + lane));
#pragma unroll
        for (int k = 0; k < 3; ++k) Sb3[k] = cld<T>(reinterpret_cast<const char*>(pk + (12 + k) * WAVE + lane));
#pragma unroll
        for (int k = 0; k < 3; ++k) both_halves(Sb3[k], Sw6[k], Sw6[3 + k]);
#pragma unroll
        for (int a_ = 0; a_ < 6; ++a_)   // (k_fslots' base term, read where it reads it)
#pragma unroll
          for (int b2 = a_; b2 < 6; ++b2)
            hb[sym(a_, b2)] = mass * ((a_ == b2 ? P.rho : T(0)) + (P.href_tab ? P.href_tab[(size_t)(jl + 1) * HREF_ROW + 6 * a_ + b2] : P.Href[6 * a_ + b2]));
#pragma unroll
        for (int k = 0; k < 21; ++k) at[k] = T(0);
        if (jcslot >= 0) {
          const char* crec = ip + (size_t)(L.off_c + jcslot * L.crec) * pair_bytes<T>();
          for (int k = 0; k < 21; ++k)
            at[k] = a_shared ? Bf.uni[L.nc * 36 + jcslot * 21 + k]
                             : *reinterpret_cast<const T*>(crec + (size_t)(CP_ATA + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T));
        }
#if LOIKB_BUILD_CALL
        {
          FlatBuildIn bi;
#pragma unroll
          for (int k = 0; k < 9; ++k) bi.R0[k] = Rb[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) bi.t0[k] = tb[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) bi.Sw[k] = Sw6[k];
#pragma unroll
          for (int k = 0; k < 21; ++k) { bi.hb[k] = hb[k]; bi.at[k] = at[k]; }
          bi.mu = flat_decade_mu(P.mu0, kexp_b);
          bi.mu_scale = P.mu_scale;
          const FlatBuildOut<NA> bo = flat_build_slot_call<NA, G>(xb, lane, !h, j, L.nb, jd, topo, child_list, fl, maxdepth, bi);
#pragma unroll
          for (int k = 0; k < NA; ++k) Wc[k] = bo.Wc[k];
          dv = bo.dinv;
        }
#else
        flat_build_slot<NA, G>(xb, lane, !h, j, L.nb, jd, topo, child_list, fl, maxdepth, Rb, tb, Sw6, hb, at, flat_decade_mu(P.mu0, kexp_b),
                               P.mu_scale, Wc, dv);
#endif
        if (!h) {   // (slot 0 of the two)
#pragma unroll
          for (int k = 0; k < NA; ++k) wl[k * GW + j] = Wc[k];
          wl[NA * GW + j] = dv;
        }
        publish_slot(kexp_b, Wc, dv);
        if (lane == 0) {
          atomicAdd(&Bf.counters[FLAT_COUNTERS_BUILT], 1u);
          atomicAdd(&Bf.counters[FLAT_COUNTERS_DEC + (kexp_b < -16 ? 0 : kexp_b > 15 ? 31 : kexp_b + 16)], 1u);
        }
        built = true;
        tail_sync();
      }
    }
    const int dsl_in = built ? 15 : dsl_raw;
    const T* pl = pk + FLAT2_PARK_ROWS * WAVE;
    const int nl = (has_hv ? G * 6 : 0) + L.nc * cs + FISC, nl6 = nl + 6;
    T lb[FLAT2_PARK_BATCH];
#pragma unroll
    for (int k = 0; k < FLAT2_PARK_BATCH; ++k) { const int e = k * WAVE + lane; lb[k] = e < nl6 ? cld<T>(reinterpret_cast<const char*>(pl + e)) : T(0); }
    const bool with_slot = dsl_in < ndec && (MUR != 2 || ((fmask_r >> dsl_in) & 1u));
    const bool coh_slot = MUR == 2 && ((fmask_r >> (16 + (dsl_in & 15))) & 1u);   // (written during this launch: agent-scope loads)
    T win[NH + 1];
#pragma unroll
    for (int i = 0; i <= NH; ++i) {
      const int k = 2 * i + (h ? 1 : 0);
      const T* src = fslots + fslotW_at(slot, ndec, dsl_in < ndec ? dsl_in : 0, frows, fcol + (k == NA ? fdm1 : k));
      win[i] = (with_slot && isj_lane && (k == NA || k < fdm1)) ? (coh_slot ? cld<T>(reinterpret_cast<const char*>(src)) : *src) : T(0);
    }
    int r = 0;
    auto get = [&]() -> T { return cld<T>(reinterpret_cast<const char*>(pk + (r++) * WAVE + lane)); };
#pragma unroll
    for (int k = 0; k < 9; ++k) R0[k] = get();
#pragma unroll
    for (int k = 0; k < 3; ++k) t0[k] = get();
#pragma unroll
    for (int k = 0; k < 3; ++k) Sw3[k] = get();
#pragma unroll
    for (int k = 0; k < 3; ++k) v3[k] = get();
#pragma unroll
    for (int k = 0; k < 3; ++k) f3[k] = get();
#pragma unroll
    for (int k = 0; k < 3; ++k) g3[k] = get();
#pragma unroll
    for (int k = 0; k < 3; ++k) SE3[k] = get();
    w = get(); z = get(); nu = get(); s = get(); lbi = get(); ubi = get();
    tail_sync();
    if constexpr (JCL) { jc[j * 3] = lbi; jc[j * 3 + 1] = ubi; jc[j * 3 + 2] = s; }
    auto put_lds = [&](int e, T x) { if (e < nl) shv[e] = x; else if (e < nl6) xb[e - nl] = x; };   // (the scalars: load-time rows, free now)
#pragma unroll
    for (int k = 0; k < FLAT2_PARK_BATCH; ++k) put_lds(k * WAVE + lane, lb[k]);
    for (int e0 = FLAT2_PARK_BATCH * WAVE; e0 < nl6; e0 += FLAT2_PARK_BATCH * WAVE) {   // (more than three task constraints)
#pragma unroll
      for (int k = 0; k < FLAT2_PARK_BATCH; ++k) { const int e = e0 + k * WAVE + lane; lb[k] = e < nl6 ? cld<T>(reinterpret_cast<const char*>(pl + e)) : T(0); }
#pragma unroll
      for (int k = 0; k < FLAT2_PARK_BATCH; ++k) put_lds(e0 + k * WAVE + lane, lb[k]);
    }
    tail_sync();
    mu = xb[0];
    if constexpr (MUR == 1) bnorm_r = isc[FI_BNORM];
    kexp = __builtin_amdgcn_readfirstlane((int)xb[1]);
    iter = __builtin_amdgcn_readfirstlane((int)xb[2]);
    status = __builtin_amdgcn_readfirstlane((int)xb[3]);
    tail_it = __builtin_amdgcn_readfirstlane((int)xb[4]);
    nflip = __builtin_amdgcn_readfirstlane((int)xb[5]);
    cbits = 0u;
    for (int c = 0; c < L.nc; ++c) {
      const int cl = (int)cdi[c * cs + C2_LANE];
      if (isj_lane && cl >= j && cl < j + size) cbits |= 1u << c;
    }
    kslot = -(1 << 30); kslot_o = -(1 << 30);
    if (with_slot) {
      wsel = 0;
#pragma unroll
      for (int i = 0; i <= NH; ++i) {
        const int k = 2 * i + (h ? 1 : 0);
        if (k <= NA) wl[k * GW + j] = win[i];
      }
      kslot = kexp_lo + dsl_in;
      n_slot_loads = (n_slot_loads + 0x10000u) | (1u << dsl_in);
    }
    if (built) { wsel = 0; kslot = kexp_b; }   // (the slot built above)
    tail_sync();
    done = false;
    bc_valid = false;
    resumed = true;   // (no first-iteration corrections: it has iterated)
    my_iters = 0;
    any_iter = true;
    TAIL_TP(16)
  };
  int next_slot = -2;  // (plain queue) the entry store_instance fetched while its stores were in flight; -2: none
  auto load_instance = [&]() {
    const int slot_in = next_slot != -2 ? next_slot : fetch();
    next_slot = -2;
    has_inst = slot_in >= 0;
    if (!has_inst) return;
    TAIL_TP(12)
    if (SLICED && (slot_in & FLAT_PARKED)) { unpark(slot_in); return; }
    resumed = false;
    bc_valid = false;
    if constexpr (MUR == 2) fmask_r = (unsigned int)win_bits;   // (what k_fslots wrote for every listed instance)
    isj = isj_lane;
    const int slot = slot_in;
    lidx = slot;
    ip = lane_ptr<T>(Bf.tiles, L, slot);
    rec = ip + (size_t)jl * JREC * pair_bytes<T>();
    const char* srec = ip + (size_t)L.off_s * pair_bytes<T>();
    // the scalar record travels with the joints' records (one round trip to HBM, not two: the loads are independent)
    const typename Vec2<T>::type mu2 = rldp<T, false>(srec, SP_MU), bi2 = rldp<T, false>(srec, SP_BI), st2 = rldp<T, false>(srec, SP_ST);
    const typename Vec2<T>::type tag2 = rldp<T, false>(srec, SP_TAG), flip2 = rldp<T, false>(srec, SP_FLIP);
    const T tail_iter0 = rld_scal<T, false>(srec, SC_TAIL_ITER);
    T sc0[8] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
    if (lane == 0) {
      sc0[0] = rld_scal<T, false>(srec, SC_TOL_PRIMAL); sc0[1] = rld_scal<T, false>(srec, SC_TOL_DUAL);
      sc0[2] = rld_scal<T, false>(srec, SC_DELTA_Y_QP); sc0[3] = rld_scal<T, false>(srec, SC_AT_DELTA_Y_QP);
      sc0[4] = rld_scal<T, false>(srec, SC_UB_DY_PLUS); sc0[5] = rld_scal<T, false>(srec, SC_LB_DY_MINUS);
      sc0[6] = rld_scal<T, false>(srec, SC_COND1); sc0[7] = rld_scal<T, false>(srec, SC_COND2);
    }
    // ---- full width on both lanes of a joint (identical values): k_flat's load, rows indexed by the joint
    T ax[3], v[6], f[6], g[6], Sw[6], SE[6];
    const bool zero_state = (P.mode & MODE_ZERO_STATE) != 0;
    // (straight from a cold reset mu = mu0, decade 0: its W columns are requested WITH the record -- one trip to HBM, not a second
    //  one behind it.  Should the record say otherwise, the loop's slot logic loads what it needs as always.)
    const int dsl0 = -kexp_lo;
    const bool slot0 = zero_state && dsl0 >= 0 && dsl0 < ndec && (MUR != 2 || ((win_bits >> dsl0) & 1));
    T win0[NH + 1];
#pragma unroll
    for (int i = 0; i <= NH; ++i) {
      const int k = 2 * i + (h ? 1 : 0);
      win0[i] = (slot0 && isj_lane && (k == NA || k < fdm1)) ? fslots[fslotW_at(slot, ndec, dsl0, frows, fcol + (k == NA ? fdm1 : k))] : T(0);
    }
    const bool rev = jflags & JF_REVOLUTE;
    const int jflags_h = jflags;
    const T pitch_h = (jflags & JF_HELICAL) ? (T)jd[jl + 1].pitch : T(0);
    {
      // (straight from a cold reset -- the plain queue hands every instance out once -- vis, fis, g, w, z are zeros in every
      //  record: ten of the twelve pairs of a joint are not fetched.  A record's 16-byte pairs lie 1 KiB apart in the tiles of
      //  the streaming engine: every pair costs a 64-byte line of its own)
      typename Vec2<T>::type wz;
      wz.x = T(0); wz.y = T(0);
      const typename Vec2<T>::type csn = ldp<T>(rec, JP_CS), nus = rldp<T, false>(rec, JP_NUS);
      if (!zero_state) wz = rldp<T, false>(rec, JP_WZ);
      const JointDesc d = jd[jl + 1];
#pragma unroll
      for (int k = 0; k < 3; ++k) ax[k] = isj_lane ? (T)d.axis[k] : T(0);
      joint_xform<T>(d, rec, csn.x, csn.y, R0, t0);  // liMi ...
      if (!zero_state) {
        rld6<T, false>(rec, JP_V, v);
        rld6<T, false>(rec, JP_F, f);
        rld6<T, false>(rec, JP_G, g);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) { v[k] = T(0); f[k] = T(0); g[k] = T(0); }
      }
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
      lbi = ubi = T(0);
    }
    TAIL_TP(13)
    {
      unsigned int jrow4[(FLAT_JMP + 3) / 4];  // rows of the ancestors at distance 2^r in joint-indexed rows (WAVE = identity)
#pragma unroll
      for (int k = 0; k < (FLAT_JMP + 3) / 4; ++k) jrow4[k] = 0u;
#pragma unroll
      for (int r = 0; r < FLAT_JMP; ++r) jrow4[r >> 2] |= (unsigned int)(fl[j].jmp[r] >= 0 ? fl[j].jmp[r] : WAVE) << (8 * (r & 3));
      flat_world_placement<T>(xb, j, j, jrow4, njmp, R0, t0);  // ... -> oMi (FwdPassInit's oMi chain, hxx:265)
    }
    TAIL_TP(14)
    {
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
    // constraint blocks: lane of the joint, b, y, A^T y, A; then AW = X*_{0<-joint} A^T (and transposed), AW b
    for (int c = 0; c < L.nc; ++c) {
      const char* crec = ip + (size_t)(L.off_c + c * L.crec) * pair_bytes<T>();
      T* c_ = cdi + c * cs;
      if (lane < 18) {
        const int which = lane / 6, k = lane % 6;
        const int pair = which == 0 ? CP_B : which == 1 ? CP_Y : CP_ATY;
        const int dst = which == 0 ? C2_B : which == 1 ? C2_Y : C2_ATY;
        c_[dst + k] = rld<T, false>(crec + (size_t)(pair + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T));  // (y changes in the kernel)
      }
      for (int e = lane; e < LCA; e += WAVE)
        c_[C2_A + e] = a_shared ? Bf.uni[c * LCA + e]
                                : *reinterpret_cast<const T*>(crec + (size_t)(CP_A + e / 2) * pair_bytes<T>() + (e & 1) * sizeof(T));
    }
    if (jcslot >= 0) cdi[jcslot * cs + C2_LANE] = (T)j;
    tail_sync();
    cbits = 0u;
    for (int c = 0; c < L.nc; ++c) {
      const int cl = (int)cdi[c * cs + C2_LANE];
      if (isj_lane && cl >= j && cl < j + size) cbits |= 1u << c;
    }
    if (jcslot >= 0) {  // the constrained joint's lanes: column q of AW = row q of A carried to the world origin
      T* c_ = cdi + jcslot * cs;
      const T* A_ = c_ + C2_A;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        T aj[6], o[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) aj[k] = A_[6 * q + k];
        act_force(R0, t0, aj, o);
#pragma unroll
        for (int k = 0; k < 6; ++k) { c_[C2_AW + 6 * k + q] = o[k]; c_[C2_AWT + 6 * q + k] = o[k]; }
      }
      // A^T y as the instance brings it (a warm-started tailored solve arrives with the A^T y of the matrix it had BEFORE
      // UpdateEqConstraint replaced it, and upstream's first FwdPass1 uses exactly that, hxx:329-331), at the world origin
      T ay[6], o[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) ay[k] = c_[C2_ATY + k];
      act_force(R0, t0, ay, o);
#pragma unroll
      for (int k = 0; k < 6; ++k) c_[C2_ATYW + k] = o[k];
    }
    tail_sync();
    if (iscl) {
      const int ccl = lane / 6, ckl = lane - 6 * ccl;
      T* const ccb = cdi + ccl * cs;
      // A^T b at the world origin; and what the first iteration owes to an A^T y that is not A^T (the y the record holds) -- the
      // loop forms the constraint's force as AW y and g without an A^T y term, which is exact from the second iteration on
      // (DualUpdate leaves A^T y = A^T y_new, hxx:422): C2_CW = X*(A^T y used) - AW y_old goes into the first f, C2_DLT =
      // A^T y_old - (A^T y used) into the first g.  Zero for a cold start; rounding noise for a warm start with an unchanged A.
      const int k = ckl;
      T ab = T(0);
#pragma unroll
      for (int q = 0; q < 6; ++q) ab += ccb[C2_AW + 6 * k + q] * ccb[C2_B + q];
      ccb[C2_ATBW + k] = ab;
      const T aw = awy_of(reinterpret_cast<const char*>(ccb), (unsigned int)(8 * ckl)), at = aty_of(reinterpret_cast<const char*>(ccb), (unsigned int)(8 * ckl));
      ccb[C2_CW + k] = resumed ? T(0) : ccb[C2_ATYW + k] - aw;
      ccb[C2_DLT + k] = resumed ? T(0) : at - ccb[C2_ATY + k];
      if (resumed) ccb[C2_ATYW + k] = aw;  // (what the loop had left there)
    }
    TAIL_TP(15)
    // subtree sums of the state the instance arrives with (cold start: v = 0) and of the reference term
    {
      T vw[6], E[6];
      T a[3], l[3], c[3], c1[3], c2[3];
      mat3_vec(R0, v, l);
      mat3_vec(R0, v + 3, a);
      cross3(t0, a, c);
#pragma unroll
      for (int k = 0; k < 3; ++k) { vw[k] = l[k] + c[k]; vw[3 + k] = a[k]; }
      // E = mass * (R0 v_l, R0 v_a + t0 x R0 v_l), from the world-frame motion: R0 v_l = vw_l - t0 x vw_a
      cross3(t0, vw + 3, c1);
#pragma unroll
      for (int k = 0; k < 3; ++k) E[k] = vw[k] - c1[k];
      cross3(t0, E, c2);
#pragma unroll
      for (int k = 0; k < 3; ++k) E[3 + k] = vw[3 + k] + c2[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) E[k] *= mass;
      if (zero_state) {   // (v = 0 in every record: the subtree sums are zeros -- five window-doubling exchanges not made)
#pragma unroll
        for (int k = 0; k < 6; ++k) SE[k] = T(0);
      } else {
        flat_subtree_sum<T>(xb, j, j, G, size, nscan, E, SE);
      }
      if (has_hv) {
        T hv[6], hw[6], Sh[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) hv[k] = mass * (HM == 3 ? hrow[36 + k] : P.Hv[k]);
        act_force(R0, t0, hv, hw);
        flat_subtree_sum<T>(xb, j, j, G, size, nscan, hw, Sh);
#pragma unroll
        for (int k = 0; k < 6; ++k) shv[j * 6 + k] = Sh[k];
      }
    }
    half(Sw, Sw3); half(v, v3); half(f, f3); half(g, g3); half(SE, SE3);
    if constexpr (JCL) { jc[j * 3] = lbi; jc[j * 3 + 1] = ubi; jc[j * 3 + 2] = s; }
    mu = mu2.x;
    kexp = (int)mu2.y;
    kslot = -(1 << 30); kslot_o = -(1 << 30);
    if (slot0) {
      wsel = 0;
#pragma unroll
      for (int i = 0; i <= NH; ++i) {
        const int k = 2 * i + (h ? 1 : 0);
        if (k <= NA) wl[k * GW + j] = win0[i];
      }
      kslot = 0;
      n_slot_loads = (n_slot_loads + 0x10000u) | (1u << dsl0);
    }
    status = (int)st2.x;
    if constexpr (MUR == 1) bnorm_r = bi2.x;
    iter = (int)bi2.y;
    tail_it = (int)tail_iter0;
    nflip = (int)flip2.x;
    done = (status & ST_DONE) != 0;
    if (!done && !(status & ST_TAIL) && iter + 1 >= P.max_iter) { done = true; status |= ST_DONE; }
    if (lane == 0) {
      isc[FI_BNORM] = bi2.x; isc[FI_TGIN] = tag2.x; isc[FI_STY] = st2.y; isc[FI_MULAST] = T(-1);
      isc[FI_TOLP] = sc0[0]; isc[FI_TOLD] = sc0[1];
      isc[FI_DYQP] = sc0[2]; isc[FI_ATDY] = sc0[3];
      isc[FI_UBP] = sc0[4]; isc[FI_LBM] = sc0[5];
      isc[FI_C1] = sc0[6]; isc[FI_C2] = sc0[7];
    }
    tail_sync();
    my_iters = 0;
    any_iter = false;
    TAIL_TP(16)
  };
  bool requeue = false;  // (SLICED) the instance goes back to the queue, to be continued by whichever wavefront takes it
  auto store_instance = [&]() {
    char* srec = ip + (size_t)L.off_s * pair_bytes<T>();
    // (plain queue) the next ticket is drawn now: the atomic's round trip runs under the stores below
    unsigned int nx_raw = 0u;
    if constexpr (!SLICED) { if (lane == 0) nx_raw = atomicAdd(q_head, 1u); }
    {
      T v[6], f[6], g[6];
      whole(v3, v); whole(f3, f); whole(g3, g);
      if (isj && !h) {
        rst6<T, false>(rec, JP_V, v);
        rst6<T, false>(rec, JP_F, f);
        rst6<T, false>(rec, JP_G, g);
        rstp<T, false>(rec, JP_WZ, w, z);
        rstp<T, false>(rec, JP_NUS, nu, JCL ? jc[j * 3 + 2] : s);
        if (any_iter) {
          // inter-sweep temporaries of the last iteration: r_i and Dinv_i.  This engine forms neither UDinv_i nor the
          // accumulated p_i: the scalar record's tag says so (SP_TAG = -2) and the getters rebuild them (k_rebuild_ud).
          rstp<T, false>(rec, JP_R, rbuf[j], wl[(wsel * (NA + 1) + NA) * GW + j]);
        }
      }
    }
    if (iscl) {  // y and A^T y (hxx:422; an instance that did not iterate goes back with the A^T y it came with)
      const int ccl = lane / 6, ckl = lane - 6 * ccl;
      const T* const ccb = cdi + ccl * cs;
      char* crec = ip + (size_t)(L.off_c + ccl * L.crec) * pair_bytes<T>();
      const int k = ckl;
      rst<T, false>(crec + (size_t)(CP_Y + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T), ccb[C2_Y + k]);
      rst<T, false>(crec + (size_t)(CP_ATY + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T), any_iter ? aty_of(reinterpret_cast<const char*>(ccb), (unsigned int)(8 * ckl)) : ccb[C2_ATY + k]);
    }
    if (LOG && lane == 0 && any_iter) Bf.log_rows[lidx] = iter - ((status & ST_TAIL) ? tail_it : 0);  // (main-loop iterations)
    if (lane == 0) {
      rstp<T, false>(srec, SP_MU, mu, (T)kexp);
      rstp<T, false>(srec, SP_TAG, any_iter ? T(-2) : isc[FI_TGIN], T(0));
      rstp<T, false>(srec, SP_BI, isc[FI_BNORM], (T)iter);
      rstp<T, false>(srec, SP_FLIP, (T)nflip, T(0));
      rstp<T, false>(srec, SP_ST, (T)(any_iter ? (status & ~ST_PFULL) : status), any_iter ? isc[FI_MULAST] : isc[FI_STY]);
      if (any_iter) {
        const T* rr = isc + FI_RED;  // prt prs stf dvis dnu dfis dyis dw av nu hrefv g dualv (13), filled when the instance stopped
        const T mu_s = mu;
        rstp<T, false>(srec, SP_SCAL + 0, isc[FI_PRIMAL], isc[FI_DUAL]);
        rstp<T, false>(srec, SP_SCAL + 1, rr[0], rr[1]);
        rstp<T, false>(srec, SP_SCAL + 2, rr[12], rr[2]);
        rstp<T, false>(srec, SP_SCAL + 3, isc[FI_TOLP], isc[FI_TOLD]);
        rstp<T, false>(srec, SP_SCAL + 4, mu_s, P.mu_scale * mu_s);
        rstp<T, false>(srec, SP_SCAL + 5, mu_s, isc[FI_DX]);
        rstp<T, false>(srec, SP_SCAL + 6, isc[FI_DZ], isc[FI_DYQP]);
        rstp<T, false>(srec, SP_SCAL + 7, isc[FI_ATDY], isc[FI_UBP]);
        rstp<T, false>(srec, SP_SCAL + 8, isc[FI_LBM], rr[5]);
        rstp<T, false>(srec, SP_SCAL + 9, rr[6], rr[7]);
        rstp<T, false>(srec, SP_SCAL + 10, rr[3], rr[4]);
        rstp<T, false>(srec, SP_SCAL + 11, rr[8], rr[9]);
        rstp<T, false>(srec, SP_SCAL + 12, rr[10], rr[11]);
        rstp<T, false>(srec, SP_SCAL + 13, rr[2], isc[FI_C1]);
        rstp<T, false>(srec, SP_SCAL + 14, isc[FI_C2], (T)tail_it);
      }
    }
    n_inst_iters += my_iters;
    if constexpr (!SLICED) {
      const int nx = __builtin_amdgcn_readfirstlane((int)nx_raw);
      if (nx >= nslots) note_dry();
      next_slot = nx < nslots ? ring[nx] : -1;
    }
    tail_sync();
    TAIL_TP(19)
  };

  // (two loops: everything only the load / store of an instance needs lives across the inner loop without being touched in it,
  //  so the register allocator can park it around the loop instead of in it)
  // (The in-wave build of a decade slot -- two hundred live values, LDS rows over both decade slots -- must not take part in the register
  //  allocation of the iteration loop: inlined there, the loop picked up three scratch reloads and two scalar loads per iteration; as a
  //  call it made the allocator spill around the call site through the whole loop.  The iteration loop therefore only ASKS for the
  //  build and leaves; the build runs out here, beside the load and the store of an instance, and the loop is entered again with the
  //  instance it had.)
  bool need_build = false;
  T inv_mu = T(1);
  int q_lim_run = 0, slice_end = 0x7fffffff, q_lim = 0;
  while (true) {
    if (!need_build) {
    load_instance();
    if (!has_inst) break;  // the queue is empty (SLICED: and every instance has retired): this wavefront is done
    inv_mu = T(1) / mu;  // (a division per change of mu, not per iteration: BoxProj's 1 / mu_ineq, hxx:384-397)
    requeue = false;
    // the last value of iter from which a quiet iteration may go straight into the next one (see quiet_f32): not the last but one of
    // max_iter, and this launch's share of iterations not used up (my_iters = iter - iter at load + 1 inside an iteration)
    q_lim_run = quiet_limit(P.max_iter, P.max_launch_iters, iter);
    // SLICED: a time slice ends at iteration slice_end, and it ends THROUGH THE SAME COMPARE -- the iteration itself has no code for
    // the slices (a counter and its compare against the kernel argument in the loop cost the SLICED build 6 % of every iteration:
    // the compiler fetched the argument with s_load + s_waitcnt each time)
    slice_end = SLICED ? iter + (iter >= slice_len ? slice_len2 : slice_len) : 0x7fffffff;
    q_lim = (SLICED && slice_end - 1 < q_lim_run) ? slice_end - 1 : q_lim_run;
    }
    need_build = false;
   while (true) {
    if (SLICED && !done && iter >= slice_end) {
      if (quantum & FLAT_Q_PROBE) {
        // the probe launch: the slices end at the marks (the residual is noted, the instance carries on) and at the probe's length:
        // parked, to be finished by the next launch
        const int k0 = slice_len, k1 = slice_len + slice_len2, km = slice_len2 > FLAT_PROBE_LAST ? k1 - FLAT_PROBE_LAST : k1;
        if (iter >= k1) { requeue = true; break; }
        if (iter <= k0) { if (lane == 0) isc[FI_MARK0] = isc[FI_MARK1]; }
        else if (lane == 0) isc[FI_MARKM] = isc[FI_MARK1];
        slice_end = iter < km ? km : k1;
      } else {
        // time slice used up: if others wait, to the back of the queue (one round trip to the queue's counters per slice: ~2 us in
        // several hundred; slightly stale is fine, the decision is a heuristic)
        unsigned int qt = 0u, qh = 0u;
        if (lane == 0) {
          qt = __hip_atomic_load(q_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          qh = __hip_atomic_load(q_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (__builtin_amdgcn_readfirstlane((int)(qt - qh) > 0 ? 1 : 0)) { requeue = true; break; }
        slice_end = iter + (iter >= slice_len ? slice_len2 : slice_len);   // nobody waits for this wavefront: carry on
      }
      q_lim = (slice_end - 1 < q_lim_run) ? slice_end - 1 : q_lim_run;
    }
    // ---- does the instance leave before this iteration?  (fetched already finished; this launch's share of iterations used up;
    // mu left the precomputed decades.)  Then it goes back as it is.
    bool exit_now = done || (int)my_iters >= P.max_launch_iters;
    // ---- decade of mu: W rows and Dinv.  Two decades stay in LDS: a flip back to the previous one costs nothing.
    if (!exit_now && kexp != kslot) {
      // (MUR = 1: the one slot the table holds is mu0's -- every instance starts there, and a solve that starts from a cold reset does its
      //  first iterations on it: a quarter of the rule's builds saved)
      const int dsl = MUR == 1 ? 0 : kexp - kexp_lo;
      const bool in_table = MUR == 1 ? (ndec > 0 && __builtin_amdgcn_readfirstlane((int)(mu == P.mu0)) != 0)
                                     : (dsl >= 0 && dsl < ndec && (MUR != 2 || ((fmask_r >> (dsl & 15)) & 1u)));
      if (MUR != 1 && kexp == kslot_o) {
        { const int tk = kslot; kslot = kslot_o; kslot_o = tk; }
        wsel ^= 1;
        ++n_slot_hits;
      } else if (!in_table && !BUILD) {
        exit_now = true;  // mu left the precomputed decades: written back unfinished, k_tail takes over
        if (lane == 0) atomicAdd(&Bf.counters[2], 1u);
      } else if (__builtin_expect(!in_table, 0)) {
        if (SLICED && MUR == 2 && (kexp < -8 || kexp > 7)) {   // (the ring entry has four bits for the decade: beyond them, k_tail as ever)
          exit_now = true;
          if (lane == 0) atomicAdd(&Bf.counters[2], 1u);
        } else {
          need_build = true;   // (the wavefront builds the slot itself -- OUTSIDE this loop: see the loop around it)
          break;
        }
      } else {
        {
          if (MUR == 2) dec_seen |= 1u << (kexp < -16 ? 0 : kexp > 15 ? 31 : kexp + 16);   // (flushed when the wavefront ends: the host asks which decades, not how often)
          kslot_o = kslot;  // the slot that was not used last is overwritten
          wsel ^= 1;
          T* wdst = wl + (size_t)wsel * (NA + 1) * GW;
          // rows k = 2 i + h of the joint's column (k <= NA: the linear lane also fetches the Dinv row)
          T in[NH + 1];
          const bool coh = MUR == 2 && ((fmask_r >> (16 + (dsl & 15))) & 1u);   // (written during this launch: agent-scope loads)
#pragma unroll
          for (int i = 0; i <= NH; ++i) {
            const int k = 2 * i + (h ? 1 : 0);
            const T* src = fslots + fslotW_at(lidx, ndec, dsl, frows, fcol + (k == NA ? fdm1 : k));
            in[i] = (isj && (k == NA || k < fdm1)) ? (coh ? cld<T>(reinterpret_cast<const char*>(src)) : *src) : T(0);
          }
          tail_sync();
#pragma unroll
          for (int i = 0; i <= NH; ++i) {
            const int k = 2 * i + (h ? 1 : 0);
            if (k <= NA) wdst[k * GW + j] = in[i];
          }
          kslot = kexp;
          n_slot_loads = (n_slot_loads + 0x10000u) | (1u << dsl);
          tail_sync();
          TAIL_TP(17)
        }
      }
    }
    if (exit_now) break;
    const T* wcur = wl + (size_t)wsel * (NA + 1) * GW;
    TAIL_TP(8)
   next_iteration:   // (a quiet iteration of a build without LOIKB_TAU_AHEAD comes straight back here: nothing above can have changed)
    unsigned int itok = my_iters + 1u;   // (changes every iteration: see field_here)
    unsigned int h3b = (opaque_here((unsigned int)lane, itok) >> 5) * 24u;   // byte offset of this half in a 6-vector of a constraint block
    // ================= p^base at the world origin, summed over the subtrees; tau  (FwdPass1 + the p part of BwdPass) ===========
    // (only the iterations that come through the loop's top form tau here.  A quiet iteration has formed the NEXT iteration's tau
    //  already, beside the chain that ends in f -- everything tau needs is final by then: the new subtree sums, w, z, the constraints'
    //  A^T y at the origin -- and goes straight to `after_tau`: the ~500 cycles of dependent steps a lone wavefront spent here at the
    //  start of every iteration, with nothing else to issue, now run under the f chain's own latencies)
    T tau, tau2 = T(0);
    T wc[NH], dinv;  // this lane's share of the joint's W entries (ancestors k = 2 i + h): used twice, up and down
#pragma unroll
    for (int i = 0; i < NH; ++i) wc[i] = wcur[(2 * i + (h ? 1 : 0)) * GW + j];
    dinv = wcur[NA * GW + j];
    {
      const T mu_eq = P.mu_scale * mu, mu_in = mu;
      T PB[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) PB[k] = -P.rho * SE3[k];
      if (has_hv) {
#pragma unroll
        for (int k = 0; k < 3; ++k) PB[k] -= shv[j * 6 + h3 + k];
      }
      for (int c = 0; c < L.nc; ++c) {
        const T* c_ = reinterpret_cast<const T*>(reinterpret_cast<const char*>(cdi + c * cs) + h3b);
        const T m = cmask(c);
#pragma unroll
        for (int k = 0; k < 3; ++k) PB[k] += m * (c_[C2_ATYW + k] - mu_eq * c_[C2_ATBW + k]);
      }
      const T d3 = Sw3[0] * PB[0] + Sw3[1] * PB[1] + Sw3[2] * PB[2];
      tau = (w - mu_in * z) + pair_sum(d3);
    }
   after_tau:
    const T mu_eq = P.mu_scale * mu, mu_in = mu;
    ++my_iters; any_iter = true;
    ++n_wave_iters;
    TAIL_TP(0)
    // ================= r' = W tau: products to LDS, every lane sums its share, long rows collect their partials ================
    tail_sync();
#pragma unroll
    for (int i = 0; i < NH; ++i) xb[(2 * i + (h ? 1 : 0)) * GW + j] = wc[i] * tau;
    tail_sync();
    T rn;
    {
      T a[4];
      static_for<0, 4>([&](auto t) { a[t] = lds_abs<0>(field_here<16 * (t & 1), 16>(ra2[t >> 1], itok)); });   // (xb)
      T acc = (a[0] + a[1]) + (a[2] + a[3]);
      acc = pair_sum(acc);
      if (!h) pbuf[j] = helper ? acc : T(0);
      tail_sync();
      T pp[4];
      static_for<0, 4>([&](auto q) { pp[q] = lds_abs<flat2_off_pbuf<NA>() * 8>(field_here<16 * (q & 1), 16>(part2[q >> 1], itok)); });   // (pbuf)
      T ps = (pp[0] + pp[1]) + (pp[2] + pp[3]);
      ps = pair_sum(ps);
      if (helper) acc = T(0);
      rn = tau + (acc + ps);
      if (!h) { rbuf[j] = rn; nbuf[j] = dinv * rn; }
    }
    tail_sync();
    TAIL_TP(1)
    // ================= nu = -W^T (Dinv r')  (FwdPass2's nu_i, hxx:127) ======================================================
    T nui;
    {
      // (measured and rejected: these gathers -- and the partial sums above -- from lane to lane through the LDS crossbar,
      //  ds_bpermute, instead of a write, a fence and the reads: one dependent trip less each, and 13.0 -> 13.9 ms)
      T nb_[NH];
      static_for<0, NH>([&](auto i) { nb_[i] = lds_abs<flat2_off_nbuf<NA>() * 8>(field_here<10 * (i % 3), 10>(anc3[i / 3], itok)); });   // (nbuf)
      T acc = T(0), acc2 = T(0);   // (two chains: see awy_of)
#pragma unroll
      for (int i = 0; i < NH; ++i) { if (i & 1) acc2 += wc[i] * nb_[i]; else acc += wc[i] * nb_[i]; }
      nui = -(dinv * rn + pair_sum(acc + acc2));
    }
    // ================= v = J nu: path sum of S^w nu at the world origin, then into the link frame (hxx:125-134) ===============
    T vi3[3], E3[3];
    {
      T y[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) y[k] = Sw3[k] * nui;
      if constexpr (PB2) flat_path_sum4<T, 3, 1>(xb, lane, pathA, anc3[(NH + 2) / 3 - 1], cb_pk, njmp, y, itok);
      else flat_path_sum4<T, 3>(xb, lane, pathA, pathB, pathC, njmp, y, itok);
      if (jcslot >= 0) {  // the constrained links' velocities at the world origin: the task constraints' update starts from them
#pragma unroll
        for (int k = 0; k < 3; ++k) cdi[jcslot * cs + C2_VC + h3 + k] = y[k];
      }
      // the two halves meet: R0 v_l = vw_l - t0 x vw_a is needed by both lanes (the linear lane rotates it into the link frame,
      // the angular lane builds the angular part of the link's momentum-like vector E from it)
      T lin[3], ang[3], c1[3], El[3], c2[3], X[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) both_halves(y[k], lin[k], ang[k]);
      cross3(t0, ang, c1);
#pragma unroll
      for (int k = 0; k < 3; ++k) El[k] = lin[k] - c1[k];
      cross3(t0, El, c2);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        X[k] = y[k] - hz * c1[k];               // linear lane: El; angular lane: its own half, ang
        E3[k] = mass * X[k] + mha * c2[k];      // mass * (El | ang + t0 x El)
      }
      mat3t_vec(R0, X, vi3);  // SE3::actInv(Motion): (R^T (v_l - t x v_a), R^T v_a)
    }
    auto hrefv_of = [&](const T* vv, T* out) {   // this half of H_ref v (link frame)
      if constexpr (HM >= 2) {
        T vl[3], va[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) both_halves(vv[k], vl[k], va[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const T* hr = (HM == 3 ? hrow : hmat) + (h3 + k) * 6;
          out[k] = ((hr[0] * vl[0] + hr[1] * vl[1]) + hr[2] * vl[2]) + ((hr[3] * va[0] + hr[4] * va[1]) + hr[5] * va[2]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] = (HM == 1 ? hd[k] : href_s) * vv[k];
      }
    };
    T hv3[3];
    hrefv_of(vi3, hv3);
    TAIL_TP(2)
    // ================= DualUpdate of the task constraints (hxx:410-451) inside the subtree sum of the links' velocities ==========
    // What the stopping logic looks at.  The four maxima of every iteration take these values SIGNED (v_max_f64's |x| modifiers do
    // the abs); what only the rare iterations look at -- the certificate's second test, the tail solve's rule, the getters' norms when
    // an instance stops, the relative tolerances -- is formed there, from these and from the state the iteration leaves (round 4:
    // ~34 instructions of every iteration, 14 of them fp64).
    T s_dy = T(0), s_av = T(0), s_ek = T(0);
    T l_dfis = T(0), l_dvis = T(0), s_dnu = T(0), s_dz = T(0), s_dw = T(0), s_prs = T(0);
    T l_dg = T(0), s_stf = T(0), s_dstf = T(0), l_dualv = T(0);
    T fi3[3], si, d32 = T(0);
    {
      T SEn[3], SHn[3], Fw[3];  // (SHn: HD only -- the subtree sums of the links' H_ref v at the world origin)
      // ---- the task constraints' update: (A v - b, dy, y), then the constraint's force AW y at the world origin: two dependent
      // exchanges through the constraint block in LDS (lane 6 c + k owns row k).  A v_c = AW^T v^w_c: no frame change first.
      const bool first = my_iters == 1u && !resumed;  // (the first iteration of a fresh record: see load_instance)
      tail_sync();
      const char* const ccb0 = reinterpret_cast<const char*>(cdi) + field_here<0, 16>(cb_pk, itok);   // this lane's constraint block
      const unsigned int ck8 = field_here<16, 6>(cb_pk, itok);                                         // 8 k
      {
        const T* col = reinterpret_cast<const T*>(ccb0 + C2_AWT * 8 + 6 * ck8);
        const T* vc = reinterpret_cast<const T*>(ccb0 + C2_VC * 8);
        const T avk = ((col[0] * vc[0] + col[1] * vc[1]) + col[2] * vc[2]) + ((col[3] * vc[3] + col[4] * vc[4]) + col[5] * vc[5]);
        const T bk = *reinterpret_cast<const T*>(ccb0 + C2_B * 8 + ck8);
        const T ek = avk - bk;
        const T dy = mu_eq * ek;
        s_dy = dy; s_ek = ek; s_av = avk;
        *reinterpret_cast<T*>(const_cast<char*>(ccb0) + C2_Y * 8 + ck8) += dy;
      }
      TAIL_TP(9)
      // (round 3 formed A v, A^T y, AW y, A^T dy and AW dy here, 30 multiply-adds on six lanes and 170 instructions of the
      //  wavefront: A^T y only matters for the stored record -- store_instance -- and the dy products are the y products'
      //  differences)
      // ---- subtree sums of E (BwdPass2's transport, hxx:210-212, as a force balance at the world origin): the joints of a subtree
      // are the lanes [j, j + size) of this half, so S_j = P[j + size - 1] - P[j] + E_j with P the inclusive prefix sum -- in
      // registers (DPP), placed here to run while the constraint block is on its way through LDS
      {
        T Pk[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) Pk[c] = prefix32(E3[c]);
        const int src = lane + (size > 0 ? size - 1 : 0);
        // (the prefix at the subtree's last joint comes through LDS rows, free at this point: ds_bpermute -- no write, no fence --
        //  measured 1.5 % slower on the headline and 2 % on a lone instance)
        T E2[3], P2[3];
        if constexpr (HD) {
          // the link's weighted velocity H_ref v (link frame) as a force at the world origin: (R0 f_l, R0 f_a + t0 x R0 f_l)
          T fl[3], y3[3], L3[3], A3[3], cc[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) fl[k] = mass * hv3[k];
          mat3_vec(R0, fl, y3);
#pragma unroll
          for (int k = 0; k < 3; ++k) both_halves(y3[k], L3[k], A3[k]);
          cross3(t0, L3, cc);
#pragma unroll
          for (int k = 0; k < 3; ++k) { E2[k] = y3[k] + hh * cc[k]; P2[k] = prefix32(E2[k]); }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) xb[lane * 3 + c] = Pk[c];
        if constexpr (HD) {
#pragma unroll
          for (int c = 0; c < 3; ++c) xb[(WAVE + 1) * 3 + lane * 3 + c] = P2[c];
        }
        tail_sync();
#pragma unroll
        for (int c = 0; c < 3; ++c) SEn[c] = (xb[src * 3 + c] - Pk[c]) + E3[c];
        if constexpr (HD) {
#pragma unroll
          for (int c = 0; c < 3; ++c) SHn[c] = (xb[(WAVE + 1) * 3 + src * 3 + c] - P2[c]) + E2[c];
        }
      }
      TAIL_TP(10)
      {
        // X* (A^T y_new) = AW y: what the next FwdPass1 adds to p of the constrained joint (hxx:329-331) and, this iteration,
        // the constraint's force in f: (A^T y used) + A^T dy
        const T aw = awy_of(ccb0, ck8);
        const T cw = first ? *reinterpret_cast<const T*>(ccb0 + C2_CW * 8 + ck8) : T(0);
        *reinterpret_cast<T*>(const_cast<char*>(ccb0) + C2_ATYW * 8 + ck8) = aw;
        *reinterpret_cast<T*>(const_cast<char*>(ccb0) + C2_ATYF * 8 + ck8) = aw + cw;
      }
      tail_sync();
      TAIL_TP(11)
      // ---- per-joint work that needs v and nu only -- BoxProj, the w update, their norms, g (hxx:129-158, :384-397, :454-458) --
      {
        T dv[3], gi[3], dg[3], dvr[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          dv[k] = vi3[k] - v3[k];
          // g_i = A^T y_i + sum_children act(f_c) - f_i = A^T y_i - (H^base_i v_i + p^base_i)  (force balance)
          gi[k] = -mass * (P.rho * dv[k] + hv3[k]);
        }
        if (has_hv) {
#pragma unroll
          for (int k = 0; k < 3; ++k) gi[k] += mass * hvl3_of(k);
        }
        if (first && jcslot >= 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) gi[k] += cdi[jcslot * cs + C2_DLT + h3 + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          dg[k] = gi[k] - g3[k];
          dvr[k] = mass * hv3[k] + gi[k];  // dual residual, v block (hxx:228): H_ref v - Hv + g
        }
        if (has_hv) {
#pragma unroll
          for (int k = 0; k < 3; ++k) dvr[k] -= mass * hvl3_of(k);
        }
        l_dualv = hinf3(dvr);
        l_dvis = mass * hinf3(dv);
        s_dnu = nui - nu;
        const T x = nui + inv_mu * w;
        const T zi = JCL ? hmin(jc[j * 3 + 1], hmax(jc[j * 3], x)) : hmin(ubi, hmax(lbi, x));
        s_dz = zi - z;
        s_prs = nui - zi;
        const T dwi = mu_in * (nui - zi);
        s_dw = dwi;
        w = w + dwi; z = zi; nu = nui;
        l_dg = hinf3(dg);
#pragma unroll
        for (int k = 0; k < 3; ++k) { v3[k] = vi3[k]; g3[k] = gi[k]; }
      }
      TAIL_TP(4)
#pragma unroll
      for (int k = 0; k < 3; ++k) Fw[k] = HD ? P.rho * (SEn[k] - SE3[k]) + SHn[k] : (P.rho + href_s) * SEn[k] - P.rho * SE3[k];
      // (TAU_AHEAD: the next iteration's p^base beside this iteration's F -- the operations of the loop's top, in its order, on the values
      //  the next iteration would find: SE3 <- SEn, the blocks' A^T y at the origin as the task update just left it)
      T PB2[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) PB2[k] = -P.rho * SEn[k];
      if (has_hv) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { Fw[k] -= shv[j * 6 + h3 + k]; if (TAU_AHEAD) PB2[k] -= shv[j * 6 + h3 + k]; }
      }
      for (int c = 0; c < L.nc; ++c) {
        const T* c_ = reinterpret_cast<const T*>(reinterpret_cast<const char*>(cdi + c * cs) + h3b);
        const T m = cmask(c);
#pragma unroll
        for (int k = 0; k < 3; ++k) Fw[k] += m * c_[C2_ATYF + k];
        if constexpr (TAU_AHEAD) {
#pragma unroll
          for (int k = 0; k < 3; ++k) PB2[k] += m * (c_[C2_ATYW + k] - mu_eq * c_[C2_ATBW + k]);
        }
      }
      {
        // SE3::actInv(Force): (R0^T F_l, R0^T (F_a - t0 x F_l)): the angular lane needs the linear half
        T Fl[3], Fa[3], cc[3], X[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) both_halves(Fw[k], Fl[k], Fa[k]);
        cross3(t0, Fl, cc);
#pragma unroll
        for (int k = 0; k < 3; ++k) X[k] = Fw[k] - hh * cc[k];
        mat3t_vec(R0, X, fi3);
      }
      {
        const T d3 = Sw3[0] * Fw[0] + Sw3[1] * Fw[1] + Sw3[2] * Fw[2];
        si = pair_sum(d3);  // S^T f (hxx:231-233): the pairing of a motion and a force does not depend on the frame
      }
      if constexpr (TAU_AHEAD) d32 = Sw3[0] * PB2[0] + Sw3[1] * PB2[1] + Sw3[2] * PB2[2];   // (its pair sum: in the fold's block, below)
#pragma unroll
      for (int k = 0; k < 3; ++k) SE3[k] = SEn[k];
    }
    TAIL_TP(5)
    // ================= what is left of the per-joint work: the norms that need f (hxx:137-146, :231-236) =========================
    {
      T df[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) df[k] = fi3[k] - f3[k];
      l_dfis = mass * hinf3(df);
      si += w;
      s_stf = si;
      if constexpr (JCL) { s_dstf = si - jc[j * 3 + 2]; jc[j * 3 + 2] = si; }
      else { s_dstf = si - s; s = si; }
#pragma unroll
      for (int k = 0; k < 3; ++k) f3[k] = fi3[k];
    }
    TAIL_TP(3)
    // ================= the scalars of the stopping logic, folded over the wavefront ===============================================
    // Main loop: four maxima (primal, dual, the two sides of the certificate's first test); the certificate's second test and the
    // tail solve's stopping rule need four more scalars, folded only when they are looked at.
    // (the norms only some iterations need: from the state this iteration left -- v3, g3, nu are the new iterates by now)
    auto l_hrefv_now = [&]() -> T {
      if constexpr (HD) { T hv[3]; hrefv_of(v3, hv); return mass * hinf3(hv); }
      else return mass * tabs(href_s) * hinf3(v3);
    };
    auto ub_lb_sums = [&](T& up, T& lm) {   // this lane's terms of ub^T [dz]_+ and lb^T [dz]_- (hpp:430-446): task rows and the joint's box
      const T bk = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(cdi) + field_here<0, 16>(cb_pk, itok) + C2_B * 8 + field_here<16, 6>(cb_pk, itok));
      up = bk * tmax(s_dy, T(0));
      lm = bk * tmin(s_dy, T(0));
      up += hz * ((JCL ? jc[j * 3 + 1] : ubi) * tmax(s_dw, T(0)));
      lm += hz * ((JCL ? jc[j * 3] : lbi) * tmin(s_dw, T(0)));
    };
    const bool fixed = P.mode & MODE_FIXED_ITERS;
    const bool in_tail = (status & ST_TAIL) != 0;
    const bool logic = !fixed && !in_tail;  // the main loop's stopping logic runs
#ifdef LOIKB_DBG_QUIET
    if (lane == 0 && in_tail) atomicAdd(&g_tail_prof_all[27], 1ull);
#endif
    T primal = T(0), dual = T(0), dyqp = T(0), atdy = T(0), dx = T(0), dz = T(0), ubp = T(0), lbm = T(0);
    T ntol_p = T(0), ntol_d = T(0);
    bool have_norms = false;
    if (logic) {
      T in[4] = {hmaxa(s_ek, s_prs), hmax_a(l_dualv, s_stf), hmax_a(hmax_a(l_dfis, s_dy), s_dw), hmax_a(l_dg, s_dstf)}, r[4];
#ifdef LOIKB_DBG_QUIET
      if (lane == 0) atomicAdd(&g_tail_prof_all[24], 1ull);
      if (quiet_f32(in, qth, iter, q_lim) && lane == 0) atomicAdd(&g_tail_prof_all[25], 1ull);
#endif
      if (LOIKB_QUIET32 && !LOG && MUR != 1 && quiet_f32(in, qth, iter, q_lim)) {   // (the quick look: see quiet_f32)
        ++iter;
        TAIL_TP(7)
        if (iter > q_lim) continue;   // (SLICED: the slice ends -- through the loop's top)
        goto next_iteration;
      }
      // (the next iteration's tau: its exchange between the halves runs beside the fold's -- two independent chains in one block)
      T wcn[NH], dinvn = T(0);   // (and its W entries: fetched here, so that the round trip to LDS is over when the iteration starts)
      if constexpr (TAU_AHEAD) {
        tau2 = (w - mu_in * z) + pair_sum(d32);
#pragma unroll
        for (int i = 0; i < NH; ++i) wcn[i] = wcur[(2 * i + (h ? 1 : 0)) * GW + j];
        dinvn = wcur[NA * GW + j];
      }
      wave_fold4<0u>(lane, in, r);
      primal = r[0]; dual = r[1]; dyqp = r[2]; atdy = r[3];
#ifdef LOIKB_DBG_QUIET
      if (lane == 0 && !(!((primal < P.tol_abs) & (dual < P.tol_abs)) & !((iter > 0) & (atdy <= P.tol_primal_inf * dyqp)) & !(primal > T(10) * dual) & !(dual > T(10) * primal) & (iter + 2 < P.max_iter))) atomicAdd(&g_tail_prof_all[26], 1ull);
      if (lane == 0 && ((primal > T(10) * dual) || (dual > T(10) * primal))) atomicAdd(&g_tail_prof_all[28], 1ull);
      if (lane == 0 && (iter > 0) & (atdy <= P.tol_primal_inf * dyqp)) atomicAdd(&g_tail_prof_all[29], 1ull);
#endif
      if (!LOG && P.tol_rel == T(0)) {
        // The common iteration decides nothing: not converged, the certificate's first test fails, mu stays where it is, not the
        // last iteration.  Five compares side by side and ONE branch; the stopping logic below is a chain of ~30 dependent
        // compare -> mask -> branch steps (900 cycles of a lone wavefront's 5400 per iteration) that only the other iterations walk.
        bool mu_stays;
        if constexpr (MUR == 1) {
          // OSQP's rule leaves mu alone while mu sqrt(x) stays within [0.2, 5] mu, x = r_p / (r_d + eps) the ratio of the normalised
          // residuals (update_mu): x in [0.04, 25].  Here WITHOUT its three divisions and the root: x = p (n_d + eps) / ((n_p + eps)
          // (d + eps (n_d + eps))), so x in (0.05, 24) is two products compared -- margins of 4 % where rounding moves 1e-16; an
          // iteration outside that band, or with mu outside the rule's clip range, walks the logic below, which evaluates the rule itself.
          T in2[4] = {hmaxa(s_av, nu), hmax_a(hmax(l_hrefv_now(), hinf3(g3)), s_stf), T(0), T(0)}, r2[4];
          wave_fold4<0u>(lane, in2, r2);
          ntol_p = r2[0]; ntol_d = r2[1];
          have_norms = true;
          const T e_ = T(1e-10), np_ = tmax(ntol_p, bnorm_r) + e_, nd_ = tmax(ntol_d, P.Hv_inf_norm) + e_;
          const T A_ = primal * nd_, B_ = np_ * (dual + e_ * nd_);
          mu_stays = (A_ < T(24) * B_) & (A_ > T(0.05) * B_) & (mu >= T(1e-6)) & (mu <= T(1e6));
        } else {
          mu_stays = !(primal > T(10) * dual) & !(dual > T(10) * primal);
        }
        const bool quiet = !((primal < tol_abs_h) & (dual < tol_abs_h)) & !((iter > 0) & (atdy <= tpi_h * dyqp)) & mu_stays & (iter + 2 < max_iter_h);
        if (quiet) {
          ++iter;
          TAIL_TP(7)
          if (LOIKB_QUIET_SKIPS_TOP && iter <= q_lim) {
            if constexpr (TAU_AHEAD) {
              tau = tau2; itok = my_iters + 1u; h3b = (opaque_here((unsigned int)lane, itok) >> 5) * 24u;
#pragma unroll
              for (int i = 0; i < NH; ++i) wc[i] = wcn[i];
              dinv = dinvn;
              goto after_tau;
            }
            goto next_iteration;
          }
          if constexpr (SLICED) { if (lane == 0) isc[FI_MARK1] = hmax(primal, dual); }   // (a slice ends here: what the probe launch's marks read)
          continue;
        }
      }
    }
    if ((MUR == 1 ? logic : P.tol_rel != T(0)) && !have_norms) {  // (uniform) relative tolerances -- and OSQP's rule -- need two more maxima
      T in2[4] = {hmaxa(s_av, nu), hmax_a(hmax(l_hrefv_now(), hinf3(g3)), s_stf), T(0), T(0)}, r2[4];
      wave_fold4<0u>(lane, in2, r2);
      ntol_p = r2[0]; ntol_d = r2[1];
    }
    TAIL_TP(6)
    // ================= CheckConvergence, CheckFeasibility, UpdateMu, the tail solve's stopping rule (hpp:377-454, :271-319) ====
    const T mu_used = mu;
    const T tol_p = P.tol_abs + P.tol_rel * tmax(ntol_p, isc[FI_BNORM]);
    const T tol_d = P.tol_abs + P.tol_rel * tmax(ntol_d, P.Hv_inf_norm);
    const int itn = iter + 1;
    if constexpr (LOG) {
      if (!in_tail) {   // (the nine lists of k_pass_solve, loik_passes.hpp, in k_flat's order)
        T inl[4] = {tabs(s_ek), tabs(s_prs), tabs(s_stf), l_dualv}, rl[4];
        wave_fold4<0u>(lane, inl, rl);
        const int row = itn - 1;
        if (lane == 0 && row < Bf.log_cap) {
          const double vals[9] = {rl[0], rl[1], hmax(rl[0], rl[1]), rl[2], rl[3], hmax(rl[3], rl[2]), mu_used, P.mu_scale * mu_used, mu_used};
#pragma unroll
          for (int l = 0; l < 9; ++l) Bf.log[((size_t)l * Bf.log_B + lidx) * Bf.log_cap + row] = vals[l];
        }
      }
    }
    const bool conv = logic && (primal < tol_p) && (dual < tol_d);
    const bool feas_chk = logic && itn > 1;
    const bool c1 = atdy <= P.tol_primal_inf * dyqp;
    bool have_b = false;
    auto fold_b = [&]() {
      T l_up, l_lm;
      ub_lb_sums(l_up, l_lm);
      T in[4] = {l_up, l_lm, hmax_a(l_dvis, s_dnu), tabs(s_dz)}, r[4];
      wave_fold4<0x3u>(lane, in, r);
      ubp = r[0]; lbm = r[1]; dx = r[2]; dz = r[3];
      have_b = true;
    };
    if (in_tail || (feas_chk && c1)) fold_b();
    bool c2 = (ubp + lbm) <= P.tol_primal_inf * dyqp;
    const bool infeas = feas_chk && c1 && c2;
    const bool enter_tail = infeas && !conv;
    const bool upd = logic && !conv && !infeas;
    bool mu_up = upd && (primal > T(10) * dual), mu_dn = upd && !mu_up && (dual > T(10) * primal);
    if constexpr (MUR == 1) {   // OSQP's rule (update_mu): the normalisers are CheckConvergence's (hxx:544-552), as in k_tail / k_solve
      mu_up = mu_dn = false;
      if (upd) {
        T m2 = mu;
        int kk = kexp;
        if (update_mu<T>(MODE_MU_OSQP, primal, dual, tmax(ntol_p, isc[FI_BNORM]), tmax(ntol_d, P.Hv_inf_norm), m2, kk)) {
          mu = m2;
          inv_mu = T(1) / mu;
          ++nflip;
          kslot = -(1 << 30);   // (the factors are those of the old mu: the next iteration starts with a build)
        }
      }
    }
    const bool tail_stop = !(dx >= P.tol_tail_solve || dz >= P.tol_tail_solve) || itn >= P.max_iter;  // (looked at with have_b only)
    const bool stop = conv || ((enter_tail || in_tail) && tail_stop) || ((upd || fixed) && itn + 1 >= P.max_iter);
    iter = itn;
    status |= (conv ? ST_CONVERGED : 0) | (infeas ? ST_PRIMAL_INF : 0) | (enter_tail ? ST_TAIL : 0) | (stop ? ST_DONE : 0);
    tail_it = enter_tail ? 0 : (in_tail ? tail_it + 1 : tail_it);
    mu = mu_up ? mu * T(10) : (mu_dn ? mu * T(0.1) : mu);
    if (mu_up || mu_dn) inv_mu = T(1) / mu;
    kexp += (mu_up ? 1 : 0) - (mu_dn ? 1 : 0);
    nflip += (mu_up || mu_dn) ? 1 : 0;
    done = stop;
    if constexpr (SLICED) { if (logic && lane == 0) isc[FI_MARK1] = hmax(primal, dual); }
    // ---- what the getters report: written when an instance stops (the certificate's scalars also when it enters the tail
    // solve: they keep the values of the last iteration that evaluated them) -------------------------------------------------
    T r1[8], r2[8];
    if (stop) {
      T in1[8] = {tabs(s_ek), tabs(s_prs), tabs(s_stf), l_dvis, tabs(s_dnu), l_dfis, tabs(s_dy), tabs(s_dw)};
      T in2[8] = {tabs(s_av), tabs(nu), l_hrefv_now(), hinf3(g3), l_dualv, T(0), T(0), T(0)};
      wave_fold8<false>(lane, in1, r1);
      wave_fold8<false>(lane, in2, r2);
      if (!logic) { primal = hmax(r1[0], r1[1]); dual = hmax(r2[4], r1[2]); }  // (the tail solve / a fixed count did not fold them)
      if (!have_b) { fold_b(); c2 = (ubp + lbm) <= P.tol_primal_inf * dyqp; }  // (reported; c1 was false: no decision hung on it)
    }
    if (stop || enter_tail) {
      if (lane == 0) {
        isc[FI_PRIMAL] = primal; isc[FI_DUAL] = dual; isc[FI_DX] = dx; isc[FI_DZ] = dz; isc[FI_MULAST] = mu_used;
        if (logic) { isc[FI_TOLP] = tol_p; isc[FI_TOLD] = tol_d; }
        if (feas_chk) {
          isc[FI_C1] = (T)(c1 ? 1 : 0); isc[FI_C2] = (T)(c2 ? 1 : 0); isc[FI_DYQP] = dyqp; isc[FI_ATDY] = atdy; isc[FI_UBP] = ubp; isc[FI_LBM] = lbm;
        }
      }
    }
    if (stop) {
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) isc[FI_RED + k] = r1[k];
#pragma unroll
        for (int k = 0; k < 5; ++k) isc[FI_RED + 8 + k] = r2[k];
      }
      tail_sync();
      TAIL_TP(18)
      break;
    }
    TAIL_TP(7)
   }
    bool build_req = false;
    if constexpr (SLICED && MUR == 2) {   // (parked with a build request: whoever unparks it builds the slot before the state is back in registers)
      if (need_build) { need_build = false; requeue = true; build_req = true; }
    }
    if (__builtin_expect(need_build, 0)) {   // (cold: the allocator's spills belong here, not in the iteration loop)
    if constexpr (BUILD && !(SLICED && MUR == 2)) {
      // the decade's own mu for the decade rule -- the columns k_fslots would have written --, the instance's mu for a rule off the
      // grid.  The builder's rows lie over BOTH slots in LDS: the other one is gone.
      T Sw6[6], hb[21], at[21], Wc[NA], dv;
#pragma unroll
      for (int k = 0; k < 3; ++k) both_halves(Sw3[k], Sw6[k], Sw6[3 + k]);
#pragma unroll
      for (int k = 0; k < 21; ++k) { hb[k] = T(0); at[k] = T(0); }
      if (!bc_valid) {
#pragma unroll
      for (int a_ = 0; a_ < 6; ++a_)   // (k_fslots' base term, read where it reads it)
#pragma unroll
        for (int b2 = a_; b2 < 6; ++b2)
          hb[sym(a_, b2)] = mass * ((a_ == b2 ? P.rho : T(0)) + (P.href_tab ? P.href_tab[(size_t)(jl + 1) * HREF_ROW + 6 * a_ + b2] : P.Href[6 * a_ + b2]));
      }
      if (!bc_valid && jcslot >= 0) {
        const char* crec = ip + (size_t)(L.off_c + jcslot * L.crec) * pair_bytes<T>();
        for (int k = 0; k < 21; ++k)
          at[k] = a_shared ? Bf.uni[L.nc * 36 + jcslot * 21 + k]
                           : *reinterpret_cast<const T*>(crec + (size_t)(CP_ATA + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T));
      }
      flat_build_slot<NA, G>(xb, lane, !h, j, L.nb, jd, topo, child_list, fl, maxdepth, R0, t0, Sw6, hb, at,
                             MUR == 1 ? mu : flat_decade_mu(P.mu0, kexp), P.mu_scale, Wc, dv, bcache, bc_valid);
      bc_valid = bcache != nullptr;
      wsel ^= 1;
      T* wdst = wl + (size_t)wsel * (NA + 1) * GW;
      if (!h) {
#pragma unroll
        for (int k = 0; k < NA; ++k) wdst[k * GW + j] = Wc[k];
        wdst[NA * GW + j] = dv;
      }
      kslot_o = -(1 << 30);
      kslot = kexp;
      if constexpr (MUR == 2) publish_slot(kexp, Wc, dv);
      if (lane == 0) {
        atomicAdd(&Bf.counters[FLAT_COUNTERS_BUILT], 1u);
        if (MUR != 1) atomicAdd(&Bf.counters[FLAT_COUNTERS_DEC + (kexp < -16 ? 0 : kexp > 15 ? 31 : kexp + 16)], 1u);
      }
      tail_sync();
      TAIL_TP(17)
    }
      continue;   // (back into the iteration loop with the instance the wavefront has)
    }
    if (SLICED && requeue) {
      // The switch in three round trips (the first version took ten, ~58 us of a wavefront under load): (1) the park stores and,
      // with them, the ticket for the NEXT entry; (2) once the stores have landed: this instance's place at the tail, and the entry
      // the ticket points at; (3) that entry's record and decade slot together (unpark).
      park_instance();
      const bool q_probe = (quantum & FLAT_Q_PROBE) != 0, q_deque = (quantum & FLAT_Q_FINISH) != 0;
      unsigned int tk = 0u;
      if (lane == 0 && !q_probe && !q_deque) tk = atomicAdd(q_head, 1u);
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): every store of the record has completed before the entry appears
      __builtin_amdgcn_wave_barrier();
      int got = -1;
      if (q_probe || q_deque) {
        // probe launch: the entry goes behind the fresh ones and stays there (k_probe_sort reads it).  The launch that finishes the
        // survivors parks for one reason only -- a decade slot to be built (MUR = 2) -- and the wavefront takes the instance up again itself.
        const int dslp = MUR == 1 ? (mu == P.mu0 ? 0 : 15) : kexp - kexp_lo;
        const int code = build_req ? (FLAT_BUILD_REQ | (((kexp + 8) & 15) << 24)) : ((dslp >= 0 && dslp < ndec ? dslp : 15) << 24);
        const int entry = lidx | FLAT_PARKED | code;
        if (q_probe) {
          if (lane == 0) {
            const unsigned int pos = atomicAdd(q_tail, 1u);
            __hip_atomic_store(ring + (pos & (unsigned int)ring_mask), entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          got = -2;   // (the next instance: an ordinary fetch)
        } else {
          got = entry;
        }
      } else
      if (lane == 0) {
        const unsigned int pos = atomicAdd(q_tail, 1u);
        int* en = ring + (tk & (unsigned int)ring_mask);
        got = __hip_atomic_load(en, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int* ep = ring + (pos & (unsigned int)ring_mask);
        if (pos > (unsigned int)ring_mask) {  // (the ring has wrapped: the place must have been consumed -- it has, long ago)
          for (unsigned int spins = 0; __hip_atomic_load(ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0; ++spins) {
            if (spins > (1u << 23)) { atomicOr(Bf.counters + FLAT_COUNTERS_ERR, 2u); break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        // (MUR = 1: the table holds mu0's slot and nothing else -- an instance whose mu has moved travels without a slot and rebuilds
        //  its factors when it is taken up again; found by the fuzz: it used to be handed mu0's factors, kexp being 0 for ever under that rule)
        const int dslp = MUR == 1 ? (mu == P.mu0 ? 0 : 15) : kexp - kexp_lo;
        const int code = build_req ? (FLAT_BUILD_REQ | (((kexp + 8) & 15) << 24)) : ((dslp >= 0 && dslp < ndec ? dslp : 15) << 24);
        __hip_atomic_store(ep, lidx | FLAT_PARKED | code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned int spins = 0; got < 0; ++spins) {   // (entries were waiting when the slice ended: normally it is there)
          if (__hip_atomic_load(q_retired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned int)nslots) break;
          if (spins > (1u << 23)) { atomicOr(Bf.counters + FLAT_COUNTERS_ERR, 1u); break; }
          __builtin_amdgcn_s_sleep(LOIKB_SPIN_SLEEP);
          got = __hip_atomic_load(en, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (got >= 0) __hip_atomic_store(en, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      next_slot = __builtin_amdgcn_readfirstlane(got);
      ++n_requeues;
      TAIL_TP(19)
    } else {
      store_instance();
      if constexpr (SLICED) { if (lane == 0) atomicAdd(q_retired, 1u); }
    }
  }
#ifdef LOIKB_TAIL_PROF
  TAIL_TP(12)   // (what is left: the last, empty fetch -- with time slices the wait for the launch's last instances)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int k = 0; k < 8; ++k) g_tail_prof[k] = prof_[k];
    for (int k = 8; k < 20; ++k) g_tail_prof[2 + k] = prof_[k];
    g_tail_prof[8] = n_wave_iters;
    g_tail_prof[9] = (clock64() - clk0_) * 100000ull / (wall_clock64() - wall0_ + 1);
  }
  if (threadIdx.x == 0) {
    for (int k = 0; k < 8; ++k) atomicAdd(&g_tail_prof_all[k], prof_[k]);
    for (int k = 8; k < 20; ++k) atomicAdd(&g_tail_prof_all[2 + k], prof_[k]);
    atomicAdd(&g_tail_prof_all[8], (unsigned long long)n_wave_iters);
    atomicAdd(&g_tail_prof_all[9], 1ull);
  }
#endif
  if (lane == 0) {
    atomicAdd(&Bf.counters[1], n_inst_iters);
    if (n_requeues) atomicAdd(&Bf.counters[LEAN_Q_REQUEUES], n_requeues);
    atomicAdd(&Bf.counters[5], n_wave_iters);
    atomicAdd(&Bf.counters[6], n_slot_loads >> 16);
    atomicAdd(&Bf.counters[FLAT_COUNTERS_SLOT_HITS], n_slot_hits);
    atomicOr(&Bf.counters[LEAN_DECADES_SEEN], n_slot_loads & 0xFFFFu);
    if constexpr (MUR == 2) {
      for (unsigned int m = dec_seen; m; m &= m - 1u) atomicOr(&Bf.counters[FLAT_COUNTERS_DEC + __builtin_ctz(m)], 1u);
    }
  }
}
#undef topo
#undef child_list
#undef fmask
#undef maxdepth
#undef win_bits

// ---- between the probe launch and the launch that finishes its survivors (FLAT_Q_PROBE / FLAT_Q_FINISH above) -----------------------
// The survivors' ring entries (ring[n_fresh .. tail)) sorted by the iterations they are predicted to need still, longest first, into
// ring[0 ..): a counting sort in PROBE_BINS classes of four iterations, one workgroup (the survivors are 5 % of a batch; their order
// inside a class is whatever the atomics make it -- the order decides when an instance runs, never what it computes).  The prediction:
// r = max(primal, dual) fell from r0 (first mark, iteration k0) over rm (middle mark, km) to r1 at the probe's end (k1); at the SLOWER of
// the two rates -- over the whole probe, over its last stretch -- the tolerance is log(r1 / tol) / rate iterations away (the residual of
// this ADMM falls in stretches and stalls between them, with jumps at the changes of mu: the rate over the whole probe alone ranks a few
// instances per batch that have just stalled as nearly done, and ONE long runner started last costs the launch its whole chain; a
// residual that stalls or rises counts as long as max_iter allows).  An instance parked BEFORE the probe's end (a decade slot to be
// built) has no marks and counts as long.  Also resets the queue's counters for the launch that follows.
#ifndef LOIKB_FLAT_KERNELS_TU
constexpr int PROBE_BINS = 256;
__global__ void __launch_bounds__(1024) k_probe_sort(int* __restrict__ ring, int n_fresh, int ring_mask, const double* __restrict__ park,
                                                     int park_stride, int isc_off, int scal_off, unsigned int* __restrict__ counters,
                                                     int k0, int km, int k1, int max_iter, double tol)
{
  __shared__ unsigned int h[PROBE_BINS], base[PROBE_BINS];
  const int t = threadIdx.x;
  if (t < PROBE_BINS) h[t] = 0u;
  __syncthreads();
  const int nsurv = (int)counters[LEAN_Q_TAIL] - n_fresh;
  auto bin_of = [&](int entry) -> int {
    const double* pk = park + (size_t)(entry & 0xFFFFF) * park_stride;
    const int it = (int)pk[scal_off + 2];
    float pr = 4.0f * PROBE_BINS;
    if (it >= k1 && k1 > k0) {
      const float r0 = (float)pk[isc_off + FI_MARK0], rm = (float)pk[isc_off + FI_MARKM], r1 = (float)pk[isc_off + FI_MARK1];
      const float tl = (float)(tol > 1e-30 ? tol : 1e-30);
      if (r0 > 0.f && r1 > 0.f) {
        const float l1 = __logf(r1), left = l1 - __logf(tl);
        float rate = (__logf(r0) - l1) / (float)(k1 - k0);
        if (km > k0 && km < k1 && rm > 0.f) { const float r2 = (__logf(rm) - l1) / (float)(k1 - km); rate = r2 < rate ? r2 : rate; }
        pr = left <= 0.f ? 0.f : (rate > 2e-4f ? left / rate : 4.0f * PROBE_BINS);
      }
    }
    const float cap = (float)(max_iter - it);
    if (pr > cap) pr = cap;
    int b = (int)(pr * 0.25f);
    b = b < 0 ? 0 : b > PROBE_BINS - 1 ? PROBE_BINS - 1 : b;
    return PROBE_BINS - 1 - b;   // (class 0 = the longest)
  };
  for (int i = t; i < nsurv; i += blockDim.x) atomicAdd(&h[bin_of(ring[(n_fresh + i) & ring_mask])], 1u);
  __syncthreads();
  if (t == 0) {
    unsigned int acc = 0u;
    for (int b = 0; b < PROBE_BINS; ++b) { base[b] = acc; acc += h[b]; }
  }
  __syncthreads();
  for (int i = t; i < nsurv; i += blockDim.x) {
    const int entry = ring[(n_fresh + i) & ring_mask];
    ring[atomicAdd(&base[bin_of(entry)], 1u)] = entry;
  }
  if (t == 0) {
    counters[FLAT_COUNTERS_FIN_N] = (unsigned int)(nsurv > 0 ? nsurv : 0);
    counters[LEAN_Q_HEAD] = 0u;
    counters[FLAT_COUNTERS_DRY] = 0u;   // (the queue of the launch that follows has not run dry yet)
  }
}
#endif


// ------------------------------------------------------------------------------------------------------------------------
// k_flat1<NA>: ONE instance per wavefront, one lane per joint -- the robots of 33..64 joints (G = 64: the 44-DoF whole-body Talos),
// which k_flat already ran one per wavefront.  Same iteration as k_flat on whole 6-vectors, with what k_flat2 showed to pay:
// load / iterate / store as nested loops over wave-uniform state, 1 / mu kept between changes of mu, the subtree sums as
// differences of a prefix sum along the 64 lanes (DPP), the stopping logic's scalars by the DPP transpose-reduce.  One wavefront
// per SIMD (the lane state is k_flat's).
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double prefix64(double x)
{
  x = prefix32(x);
  x += dpp_f64<0x143, 0xC, 0xF, false>(0.0, x);  // row_bcast:31 into rows 2 and 3: the total of the lower half
  return x;
}

template <int NA>
__host__ __device__ constexpr int flat1_xregion()
{
  int n = XROWS * 9;
  if (NA * WAVE + 2 > n) n = NA * WAVE + 2;
  if (2 * 6 * (WAVE + 2) > n) n = 2 * 6 * (WAVE + 2);   // two sets of prefix rows [6][66] (HD)
  return (n + 1) & ~1;
}
template <int NA>
__host__ __device__ __forceinline__ size_t flat1_lds_bytes(int nc, bool has_hv, int slot_bufs = 2)
{
  const size_t n = (size_t)flat1_xregion<NA>() + (size_t)slot_bufs * (NA + 1) * WAVE + 3 * (size_t)(WAVE + 2) + (has_hv ? (size_t)WAVE * 6 : 0) +
                   (size_t)nc * C2D + FISC + 36;
  return (n * sizeof(double) + 15) & ~(size_t)15;
}

// MUR = 1 (round 5): OSQP's rule, as in k_flat2 -- mu0's slot from the table, every change of mu one in-wave build (flat_build_slot on all 64
// lanes: its rows lie over the exchange area and the decade slots, so the launch keeps TWO slots in LDS), a division-free quiet test; unsliced.
// aux[0] = TailTopo*, aux[1] = the children's list (read by the builder only); has_hv_bits: bits 8..15 = the tree's depth.
// (MUR = 2, the lazily populated table of k_flat2, was built for this kernel in round 6 -- the builder as a function of its own, called between two
//  instances -- and taken out again: k_fslots' window saves the whole body 0.5 ms of 14.4, the kernel gave the same back (the instances' fmask words,
//  the coherent slot loads), and the build's register allocation tipped over with every small change of the kernel: 13 -> 43 -> 123 scratch reloads
//  per iteration, the last one a 68 ms launch.  DESIGN section 0.)
template <int NA, bool SLICED = false, int HM = 0, bool LOG = false, int MUR = 0>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_flat1(const Params<double> P, const Bufs<double> Bf, const JointDesc* __restrict__ jd, const FlatLane* __restrict__ fl, int nanc,
        int nscan, int njmp, int* ring, int nslots, const double* __restrict__ fslots, int frows, int kexp_lo,
        int ndec, double href_s, int has_hv_bits, int ring_mask, int quantum, const void* const* __restrict__ aux)
{
  static_assert(MUR == 0 || MUR == 1, "k_flat1 has no lazily populated table (see above)");
  static_assert(MUR == 0 || (!SLICED && !LOG), "the OSQP build of k_flat1 runs unsliced and writes no lists");
  static_assert(MUR == 0 || flat_build_scratch<WAVE>() <= flat1_xregion<NA>() + 2 * (NA + 1) * WAVE, "the in-wave builder's rows must end inside the two decade slots");
  // has_hv_bits: bit 0 = the reference target is not zero (H_ref v_ref rows in LDS), bit 1 = ONE decade slot in LDS instead of two
  // (a flip of mu back to the previous decade then reloads it, ~1.4 KB from the L2; the 5.6 KB it frees are worth two more
  // wavefronts per CU with four task constraints: whole body 23.6 -> see DESIGN)
  const int has_hv = has_hv_bits & 1;
  const bool one_buf = (has_hv_bits & 2) != 0;
  using T = double;
  constexpr int G = WAVE, cs = C2D;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (!lds_starts_at_zero(smem_raw)) {   // (never: see lds_abs -- reported as an error, not computed wrongly)
    if (threadIdx.x == 0) atomicOr(Bf.counters + FLAT_COUNTERS_ERR, 4u);
    return;
  }
  const Layout& L = P.L;
  const bool a_shared = P.mode & MODE_A_SHARED;
  const int lane = threadIdx.x;
  const QuietF32 qth = quiet_f32_thresholds(P.tol_abs, P.tol_rel, P.tol_primal_inf);
  // (iterations per time slice; none: never ends.  quantum = first | later << 16: an instance's FIRST slice and its later ones -- the later
  //  ones shorter, so that the instances still running when the queue is empty have done the same number of iterations to within that)
  const int slice_len = (SLICED && (quantum & 0xffff) > 0) ? (quantum & 0xffff) : 0x3fffffff;
  const int slice_len2 = (SLICED && ((quantum >> 16) & 0x1fff) > 0) ? ((quantum >> 16) & 0x1fff) : slice_len;
  const double tol_abs_h = held<SLICED>(P.tol_abs), tpi_h = held<SLICED>(P.tol_primal_inf);   // (see k_flat2)
  const int max_iter_h = held<SLICED>(P.max_iter);
  const int j = lane;  // lane <-> device joint j + 1
  T* const xb = reinterpret_cast<T*>(smem_raw);          // load-time rows | path rows [65][6] | W tau products [NA][64]
  T* const wl = xb + flat1_xregion<NA>();                // [2][NA + 1][64]
  T* const nbuf = wl + (one_buf ? 1 : 2) * (NA + 1) * G; // [66]
  T* const pbuf = nbuf + G + 2;                          // [66]
  T* const rbuf = pbuf + G + 2;                          // [66]
  T* const shv = rbuf + G + 2;                           // [64][6] (if H_ref v_ref != 0)
  T* const cdi = shv + (has_hv ? G * 6 : 0);             // [nc][C2D]
  T* const isc = cdi + (size_t)L.nc * cs;                // [FISC]

  const bool isj_lane = j < L.nb;
  const int jl = isj_lane ? j : 0;
  const int jflags = jd[jl + 1].flags, jcslot = isj_lane ? jd[jl + 1].cslot : -1;
  const T mass = (!isj_lane || (jflags & JF_MASSLESS)) ? T(0) : T(1);
  constexpr bool HD = HM > 0;
  T hd[6];  // the diagonal of H_ref (HM = 1: see k_flat2; = href_s for HM = 0)
#pragma unroll
  for (int k = 0; k < 6; ++k) hd[k] = HM == 1 ? P.Href[7 * k] : href_s;
  T* const hmat = isc + FISC;  // [36] H_ref (HM = 2)
  if (HM == 2 && lane < 36) hmat[lane] = P.Href[lane];
  const T* const hrow = HM == 3 ? P.href_tab + (size_t)(jl + 1) * HREF_ROW : nullptr;  // (H_ref_i, H_ref_i v_ref_i) of this link (HM = 3)
  T hvl[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) hvl[k] = HM == 3 ? hrow[36 + k] : P.Hv[k];
  int size, fcol, fdm1;  // (fcol, fdm1: this joint's column in a packed decade slot, its number of ancestors)
  bool helper;
  unsigned int jrow4[(FLAT_JMP + 3) / 4], ra2[FLAT_RED / 2], prow4[(FLAT_PART + 3) / 4], anc4[(NA + 3) / 4];
  // (at most 11 ancestors: the path sum's second round has the ancestors at distance 4 and 8 only, their lanes ride in the spare half of
  //  the last word of anc4, and there is no third round: flat_path_sum4<.., 2>)
  constexpr bool PB2F1 = NA <= 11 && (NA % 4) != 0 && (NA % 4) <= 2;
  unsigned int pathA, pathB, pathC;
  flat_path_rows4(fl, j, 0, pathA, pathB, pathC);
  {
    const FlatLane F = fl[j];
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
    for (int r = 0; r < FLAT_JMP; ++r) jrow4[r >> 2] |= (unsigned int)(F.jmp[r] >= 0 ? F.jmp[r] : WAVE) << (8 * (r & 3));
#pragma unroll
    for (int t = 0; t < FLAT_RED; ++t) ra2[t >> 1] |= (unsigned int)((F.red[t] >= 0 ? F.red[t] : flat1_xregion<NA>() + (one_buf ? 1 : 2) * (NA + 1) * G + G) * 8) << (16 * (t & 1));  // (byte offsets; none: nbuf's zero pad)
#pragma unroll
    for (int q = 0; q < FLAT_PART; ++q) prow4[q >> 2] |= (unsigned int)(F.part[q] >= 0 ? F.part[q] : WAVE) << (8 * (q & 3));
#pragma unroll
    for (int k = 0; k < NA; ++k) anc4[k >> 2] |= (unsigned int)((k < FLAT_MAXA && F.anc[k] >= 0) ? F.anc[k] : WAVE) << (8 * (k & 3));
    if constexpr (PB2F1) anc4[(NA + 3) / 4 - 1] |= (((pathB & 0x3FFu) / 8u) << 16) | ((((pathB >> 10) & 0x3FFu) / 8u) << 24);   // (ancestors at distance 4 and 8, as lanes)
  }
  if (lane < 2) { nbuf[G + lane] = T(0); pbuf[G + lane] = T(0); }
  for (int e = lane; e < (one_buf ? 1 : 2) * (NA + 1) * G; e += WAVE) wl[e] = T(0);

  bool has_inst = false, isj = false, done = true, any_iter = false;
  int lidx = 0;
  char *ip = Bf.tiles, *rec = Bf.tiles;
  T R0[9], t0[3], Sw[6], v[6], f[6], g[6], SE[6];
  T w = T(0), z = T(0), nu = T(0), s = T(0), lbi = T(0), ubi = T(0), mu = T(1);
  int kexp = 0, kslot = -(1 << 30), kslot_o = -(1 << 30), wsel = 0;
  int iter = 0, status = ST_DONE, tail_it = 0, nflip = 0;
  unsigned int my_iters = 0, n_wave_iters = 0, n_slot_loads = 0, n_slot_hits = 0;
  unsigned int* q_head = Bf.counters + LEAN_Q_HEAD;
  unsigned int cbits = 0u;
  // The constraints whose joint lies in this joint's subtree as a LIST (four bits each, 15 = no more; at most ten constraints reach the flat
  // engines) and `ncl`, the longest list of the wavefront.  The two sums over the constraints (p^base for tau, the force balance for f) visit
  // ncl slots, not all L.nc blocks: every block read is a broadcast that occupies the LDS pipe like any other read (scripts/ubench/lds_rate.hip) --
  // whole body, four tasks: 72 of an iteration's ~200 LDS reads and 72 multiply-adds, of which a joint's own chains need half (a joint of a
  // fixed-base humanoid lies on the chains of at most two of the four end effectors).  A slot past a lane's list reads a valid block with weight 0,
  // as the skipped blocks had: the sums are those of the loop over all blocks, term by term.
  unsigned int cpk_lo = ~0u, cpk_hi = ~0u;
  int ncl = 0;
  auto cslot_of = [&](int s_, unsigned int tok, const T*& c_, T& m) {
    const unsigned int wd = opaque_here(s_ < 8 ? cpk_lo : cpk_hi, tok);   // (tied to the iteration: the block addresses are not hoisted into registers)
    const unsigned int cc = (wd >> (4 * (s_ & 7))) & 15u;
    m = cc != 15u ? T(1) : T(0);
    c_ = cdi + (cc != 15u ? cc : 0u) * cs;
  };
  const int ccl = lane / 6, ckl = lane - 6 * ccl;
  const bool iscl = lane < 6 * L.nc;
  T* const ccb = cdi + (iscl ? ccl : 0) * cs;   // (no null block here as in k_flat2: its 1.3 KB cost the whole-body batch a wavefront per CU)
  // (AW y)_k and (A^T y)_k of the constraint of lane 6 c + k, in ONE order of operations wherever they are formed (see k_flat2)
  auto awy_k = [&]() -> T {
    const T* r = ccb + C2_AW + 6 * ckl;
    const T* y = ccb + C2_Y;
    return ((r[0] * y[0] + r[1] * y[1]) + r[2] * y[2]) + ((r[3] * y[3] + r[4] * y[4]) + r[5] * y[5]);
  };
  auto aty_k = [&]() -> T {
    const T* A_ = ccb + C2_A + ckl;
    const T* y = ccb + C2_Y;
    return ((A_[0] * y[0] + A_[6] * y[1]) + A_[12] * y[2]) + ((A_[18] * y[3] + A_[24] * y[4]) + A_[30] * y[5]);
  };
  bool resumed = false;  // (SLICED) the instance came back from the queue: no first-iteration corrections
  auto force_of_motion = [&](const T* vw, T* E) {  // E = mass * (R0 v_l, R0 v_a + t0 x R0 v_l) from the world-frame motion
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

  auto note_dry = [&]() {  // the first wavefront to find the queue empty notes the time (the launch's bulk phase ends here)
    if (lane == 0 && atomicCAS(Bf.counters + FLAT_COUNTERS_DRY, 0u, 1u) == 0u)
      __hip_atomic_store(Bf.counters + FLAT_COUNTERS_TDRY, (unsigned int)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // ---- work queue.  Plain: the listed instances in order, one atomic per fetch.  SLICED: the ring of k_lean (ticket t is served by
  // entry t; entries are consumed by resetting them to -1; LEAN_Q_RETIRED counts the instances that will not come back)
  unsigned int* q_tail = Bf.counters + LEAN_Q_TAIL;
  unsigned int* q_retired = Bf.counters + LEAN_Q_RETIRED;
  auto fetch = [&]() -> int {
    if constexpr (!SLICED) {
      int nx = 0;
      if (lane == 0) nx = (int)atomicAdd(q_head, 1u);
      nx = __builtin_amdgcn_readfirstlane(nx);
      if (nx >= nslots) note_dry();
      return nx < nslots ? ring[nx] : -1;
    } else {
      int got = -1;
      if (lane == 0) {
        const unsigned int ticket = atomicAdd(q_head, 1u);
        int* e = ring + (ticket & (unsigned int)ring_mask);
        bool noted = false;
        for (unsigned int spins = 0;; ++spins) {
          const int v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v >= 0) { __hip_atomic_store(e, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); got = v; break; }
          if (__hip_atomic_load(q_retired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned int)nslots) break;
          if (!noted) {  // (this ticket is beyond the tail: from now on wavefronts wait for entries)
            noted = true;
            if (atomicCAS(Bf.counters + FLAT_COUNTERS_DRY, 0u, 1u) == 0u)
              __hip_atomic_store(Bf.counters + FLAT_COUNTERS_TDRY, (unsigned int)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (spins > (1u << 23)) { atomicOr(Bf.counters + FLAT_COUNTERS_ERR, 1u); break; }  // (never: a lost entry must not hang the GPU)
          __builtin_amdgcn_s_sleep(LOIKB_SPIN_SLEEP);
        }
      }
      return __builtin_amdgcn_readfirstlane(got);
    }
  };
  auto q_waiting = [&]() -> bool {  // do entries wait for a wavefront?
    int wtg = 0;
    if (lane == 0)
      wtg = (int)(__hip_atomic_load(q_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                  __hip_atomic_load(q_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) > 0;
    return __builtin_amdgcn_readfirstlane(wtg) != 0;
  };
  auto q_push = [&](int slot) {  // (after store_instance: every store of the record has completed before the entry appears)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      const unsigned int pos = atomicAdd(q_tail, 1u);
      int* e = ring + (pos & (unsigned int)ring_mask);
      for (unsigned int spins = 0; __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0; ++spins) {
        if (spins > (1u << 23)) { atomicOr(Bf.counters + FLAT_COUNTERS_ERR, 2u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
      __hip_atomic_store(e, slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicAdd(&Bf.counters[LEAN_Q_REQUEUES], 1u);
    }
  };
  auto load_instance = [&]() {
    const int slot_in = fetch();
    has_inst = slot_in >= 0;
    if (!has_inst) return;
    isj = isj_lane;
    lidx = slot_in;
    ip = lane_ptr<T>(Bf.tiles, L, slot_in);
    rec = ip + (size_t)jl * JREC * pair_bytes<T>();
    const char* srec = ip + (size_t)L.off_s * pair_bytes<T>();
    T ax[3];
    const bool rev = jflags & JF_REVOLUTE;
    const int jflags_h = jflags;
    const T pitch_h = (jflags & JF_HELICAL) ? (T)jd[jl + 1].pitch : T(0);
    {
      // (straight from a cold reset -- the plain queue hands every instance out once -- vis, fis, g, w, z are zeros in every
      //  record: ten of the twelve pairs of a joint are not fetched.  A record's 16-byte pairs lie 1 KiB apart in the tiles of
      //  the streaming engine: every pair costs a 64-byte line of its own)
      const bool zero_state = !SLICED && (P.mode & MODE_ZERO_STATE);
      typename Vec2<T>::type wz;
      wz.x = T(0); wz.y = T(0);
      const typename Vec2<T>::type csn = ldp<T>(rec, JP_CS), nus = rldp<T, SLICED>(rec, JP_NUS);
      if (!zero_state) wz = rldp<T, SLICED>(rec, JP_WZ);
      const JointDesc d = jd[jl + 1];
#pragma unroll
      for (int k = 0; k < 3; ++k) ax[k] = isj_lane ? (T)d.axis[k] : T(0);
      joint_xform<T>(d, rec, csn.x, csn.y, R0, t0);  // liMi ...
      if (!zero_state) {
        rld6<T, SLICED>(rec, JP_V, v);
        rld6<T, SLICED>(rec, JP_F, f);
        rld6<T, SLICED>(rec, JP_G, g);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) { v[k] = T(0); f[k] = T(0); g[k] = T(0); }
      }
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
      lbi = ubi = T(0);
    }
    flat_world_placement<T>(xb, lane, lane, jrow4, njmp, R0, t0);  // ... -> oMi (FwdPassInit's oMi chain, hxx:265)
    {
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
    for (int c = 0; c < L.nc; ++c) {
      const char* crec = ip + (size_t)(L.off_c + c * L.crec) * pair_bytes<T>();
      T* c_ = cdi + c * cs;
      if (lane < 18) {
        const int which = lane / 6, k = lane % 6;
        const int pair = which == 0 ? CP_B : which == 1 ? CP_Y : CP_ATY;
        const int dst = which == 0 ? C2_B : which == 1 ? C2_Y : C2_ATY;
        c_[dst + k] = rld<T, SLICED>(crec + (size_t)(pair + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T));
      }
      for (int e = lane; e < LCA; e += WAVE)
        c_[C2_A + e] = a_shared ? Bf.uni[c * LCA + e]
                                : *reinterpret_cast<const T*>(crec + (size_t)(CP_A + e / 2) * pair_bytes<T>() + (e & 1) * sizeof(T));
    }
    if (jcslot >= 0) cdi[jcslot * cs + C2_LANE] = (T)j;
    tail_sync();
    cbits = 0u;
    for (int c = 0; c < L.nc; ++c) {
      const int cl = (int)cdi[c * cs + C2_LANE];
      if (isj_lane && cl >= j && cl < j + size) cbits |= 1u << c;
    }
    {
      unsigned long long pk = ~0ull;
      int nmine = 0;
      for (int c = 0; c < L.nc; ++c)
        if ((cbits >> c) & 1u) { pk = (pk & ~(15ull << (4 * nmine))) | ((unsigned long long)c << (4 * nmine)); ++nmine; }
      cpk_lo = (unsigned int)pk; cpk_hi = (unsigned int)(pk >> 32);
      ncl = 0;
      for (int t = L.nc; t > 0; --t)
        if (__any(nmine >= t)) { ncl = t; break; }
    }
    if (jcslot >= 0) {
      T* c_ = cdi + jcslot * cs;
      const T* A_ = c_ + C2_A;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        T aj[6], o[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) aj[k] = A_[6 * q + k];
        act_force(R0, t0, aj, o);
#pragma unroll
        for (int k = 0; k < 6; ++k) { c_[C2_AW + 6 * k + q] = o[k]; c_[C2_AWT + 6 * q + k] = o[k]; }
      }
      T ay[6], o[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) ay[k] = c_[C2_ATY + k];
      act_force(R0, t0, ay, o);
#pragma unroll
      for (int k = 0; k < 6; ++k) c_[C2_ATYW + k] = o[k];
    }
    tail_sync();
    resumed = SLICED && rldp<T, SLICED>(srec, SP_TAG).x == T(-3);
    if (iscl) {  // A^T b at the world origin and the first iteration's corrections (see k_flat2)
      const int k = ckl;
      T ab = T(0);
#pragma unroll
      for (int q = 0; q < 6; ++q) ab += ccb[C2_AW + 6 * k + q] * ccb[C2_B + q];
      ccb[C2_ATBW + k] = ab;
      const T aw = awy_k(), at = aty_k();
      ccb[C2_CW + k] = resumed ? T(0) : ccb[C2_ATYW + k] - aw;
      ccb[C2_DLT + k] = resumed ? T(0) : at - ccb[C2_ATY + k];
      if (resumed) ccb[C2_ATYW + k] = aw;
    }
    {
      T vw[6], E[6];
      T a[3], l[3], c[3];
      mat3_vec(R0, v, l);
      mat3_vec(R0, v + 3, a);
      cross3(t0, a, c);
#pragma unroll
      for (int k = 0; k < 3; ++k) { vw[k] = l[k] + c[k]; vw[3 + k] = a[k]; }
      force_of_motion(vw, E);
      flat_subtree_sum<T>(xb, lane, lane, G, size, nscan, E, SE);
      if (has_hv) {
        T hv[6], hw[6], Sh[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) hv[k] = mass * hvl[k];
        act_force(R0, t0, hv, hw);
        flat_subtree_sum<T>(xb, lane, lane, G, size, nscan, hw, Sh);
#pragma unroll
        for (int k = 0; k < 6; ++k) shv[lane * 6 + k] = Sh[k];
      }
    }
    if constexpr (SLICED) {  // an instance that comes back from the queue continues exactly where it left (see k_flat2)
      if (resumed) rld6<T, SLICED>(rec, JP_P, SE);
    }
    const typename Vec2<T>::type mu2 = rldp<T, SLICED>(srec, SP_MU), bi2 = rldp<T, SLICED>(srec, SP_BI), st2 = rldp<T, SLICED>(srec, SP_ST);
    mu = mu2.x;
    kexp = (int)mu2.y;
    kslot = -(1 << 30); kslot_o = -(1 << 30);
    status = (int)st2.x;
    iter = (int)bi2.y;
    tail_it = (int)rld_scal<T, SLICED>(srec, SC_TAIL_ITER);
    nflip = (int)rldp<T, SLICED>(srec, SP_FLIP).x;
    done = (status & ST_DONE) != 0;
    if (!done && !(status & ST_TAIL) && iter + 1 >= P.max_iter) { done = true; status |= ST_DONE; }
    if (lane == 0) {
      isc[FI_BNORM] = bi2.x; isc[FI_TGIN] = rldp<T, SLICED>(srec, SP_TAG).x; isc[FI_STY] = st2.y; isc[FI_MULAST] = T(-1);
      isc[FI_TOLP] = rld_scal<T, SLICED>(srec, SC_TOL_PRIMAL); isc[FI_TOLD] = rld_scal<T, SLICED>(srec, SC_TOL_DUAL);
      isc[FI_DYQP] = rld_scal<T, SLICED>(srec, SC_DELTA_Y_QP); isc[FI_ATDY] = rld_scal<T, SLICED>(srec, SC_AT_DELTA_Y_QP);
      isc[FI_UBP] = rld_scal<T, SLICED>(srec, SC_UB_DY_PLUS); isc[FI_LBM] = rld_scal<T, SLICED>(srec, SC_LB_DY_MINUS);
      isc[FI_C1] = rld_scal<T, SLICED>(srec, SC_COND1); isc[FI_C2] = rld_scal<T, SLICED>(srec, SC_COND2);
    }
    tail_sync();
    my_iters = 0;
    any_iter = false;
  };
  bool requeue = false;
  auto store_instance = [&]() {
    char* srec = ip + (size_t)L.off_s * pair_bytes<T>();
    if (SLICED && requeue && isj) rst6<T, SLICED>(rec, JP_P, SE);
    if (isj) {
      rst6<T, SLICED>(rec, JP_V, v);
      rst6<T, SLICED>(rec, JP_F, f);
      rst6<T, SLICED>(rec, JP_G, g);
      rstp<T, SLICED>(rec, JP_WZ, w, z);
      rstp<T, SLICED>(rec, JP_NUS, nu, s);
      if (any_iter) rstp<T, SLICED>(rec, JP_R, rbuf[lane], wl[(wsel * (NA + 1) + NA) * G + lane]);  // (r_i, Dinv_i; SP_TAG = -2: see k_flat)
    }
    if (iscl) {  // y and A^T y (an instance that did not iterate goes back with the A^T y it came with)
      char* crec = ip + (size_t)(L.off_c + ccl * L.crec) * pair_bytes<T>();
      const int k = ckl;
      rst<T, SLICED>(crec + (size_t)(CP_Y + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T), ccb[C2_Y + k]);
      rst<T, SLICED>(crec + (size_t)(CP_ATY + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T), any_iter ? aty_k() : ccb[C2_ATY + k]);
    }
    if (LOG && lane == 0 && any_iter) Bf.log_rows[lidx] = iter - ((status & ST_TAIL) ? tail_it : 0);  // (main-loop iterations)
    if (lane == 0) {
      rstp<T, SLICED>(srec, SP_MU, mu, (T)kexp);
      rstp<T, SLICED>(srec, SP_TAG, (SLICED && requeue) ? T(-3) : any_iter ? T(-2) : isc[FI_TGIN], T(0));
      rstp<T, SLICED>(srec, SP_BI, isc[FI_BNORM], (T)iter);
      rstp<T, SLICED>(srec, SP_FLIP, (T)nflip, T(0));
      rstp<T, SLICED>(srec, SP_ST, (T)(any_iter ? (status & ~ST_PFULL) : status), any_iter ? isc[FI_MULAST] : isc[FI_STY]);
      if (any_iter) {
        const T* rr = isc + FI_RED;
        const T mu_s = mu;
        rstp<T, SLICED>(srec, SP_SCAL + 0, isc[FI_PRIMAL], isc[FI_DUAL]);
        rstp<T, SLICED>(srec, SP_SCAL + 1, rr[0], rr[1]);
        rstp<T, SLICED>(srec, SP_SCAL + 2, rr[12], rr[2]);
        rstp<T, SLICED>(srec, SP_SCAL + 3, isc[FI_TOLP], isc[FI_TOLD]);
        rstp<T, SLICED>(srec, SP_SCAL + 4, mu_s, P.mu_scale * mu_s);
        rstp<T, SLICED>(srec, SP_SCAL + 5, mu_s, isc[FI_DX]);
        rstp<T, SLICED>(srec, SP_SCAL + 6, isc[FI_DZ], isc[FI_DYQP]);
        rstp<T, SLICED>(srec, SP_SCAL + 7, isc[FI_ATDY], isc[FI_UBP]);
        rstp<T, SLICED>(srec, SP_SCAL + 8, isc[FI_LBM], rr[5]);
        rstp<T, SLICED>(srec, SP_SCAL + 9, rr[6], rr[7]);
        rstp<T, SLICED>(srec, SP_SCAL + 10, rr[3], rr[4]);
        rstp<T, SLICED>(srec, SP_SCAL + 11, rr[8], rr[9]);
        rstp<T, SLICED>(srec, SP_SCAL + 12, rr[10], rr[11]);
        rstp<T, SLICED>(srec, SP_SCAL + 13, rr[2], isc[FI_C1]);
        rstp<T, SLICED>(srec, SP_SCAL + 14, isc[FI_C2], (T)tail_it);
      }
      if (my_iters) atomicAdd(&Bf.counters[1], my_iters);
    }
    tail_sync();
  };

  bool need_build = false;   // (MUR = 1: the iteration loop asks for a build and leaves; it runs out here and the loop is entered again: k_flat2)
  T inv_mu = T(1), bnorm_r = T(0);
  int q_lim_run = 0, slice_end = 0x7fffffff, q_lim = 0;
  while (true) {
    if (!need_build) {
    load_instance();
    if (!has_inst) break;
    inv_mu = T(1) / mu;
    if constexpr (MUR == 1) bnorm_r = isc[FI_BNORM];
    requeue = false;
    q_lim_run = quiet_limit(P.max_iter, P.max_launch_iters, iter);   // (see k_flat2)
    slice_end = SLICED ? iter + (iter >= slice_len ? slice_len2 : slice_len) : 0x7fffffff;
    q_lim = (SLICED && slice_end - 1 < q_lim_run) ? slice_end - 1 : q_lim_run;
    }
    need_build = false;
   while (true) {
    if (SLICED && !done && iter >= slice_end) {
      if (q_waiting()) { requeue = true; break; }
      slice_end = iter + (iter >= slice_len ? slice_len2 : slice_len);
      q_lim = (slice_end - 1 < q_lim_run) ? slice_end - 1 : q_lim_run;
    }
    bool exit_now = done || (int)my_iters >= P.max_launch_iters;
    if (!exit_now && kexp != kslot) {
      if (MUR != 1 && kexp == kslot_o && !one_buf) {
        { const int tk = kslot; kslot = kslot_o; kslot_o = tk; }
        wsel ^= 1;
        ++n_slot_hits;
      } else {
        const int dsl = MUR == 1 ? 0 : kexp - kexp_lo;
        const bool in_table = MUR == 1 ? (ndec > 0 && __builtin_amdgcn_readfirstlane((int)(mu == P.mu0)) != 0) : (dsl >= 0 && dsl < ndec);
        if (MUR == 1 && __builtin_expect(!in_table, 0)) {
          need_build = true;
          break;
        } else if (!in_table) {
          exit_now = true;
          if (lane == 0) atomicAdd(&Bf.counters[2], 1u);
        } else {
          if (!one_buf) { kslot_o = kslot; wsel ^= 1; }
          T* wdst = wl + (size_t)wsel * (NA + 1) * G;
          T in[NA + 1];
#pragma unroll
          for (int k = 0; k <= NA; ++k)
            in[k] = (isj && (k == NA || k < fdm1)) ? fslots[fslotW_at(lidx, ndec, dsl, frows, fcol + (k == NA ? fdm1 : k))] : T(0);
#pragma unroll
          for (int k = 0; k <= NA; ++k) wdst[k * G + lane] = in[k];
          kslot = kexp;
          n_slot_loads = (n_slot_loads + 0x10000u) | (1u << dsl);
        }
      }
    }
    if (exit_now) break;
    const T* wcur = wl + (size_t)wsel * (NA + 1) * G;
   next_iteration:   // (a quiet iteration comes straight back here: see k_flat2)
    const T mu_eq = P.mu_scale * mu, mu_in = mu;
    ++my_iters; any_iter = true;
    const unsigned int itok = my_iters;   // (see field_here)
    ++n_wave_iters;

    // ---- p^base summed over the subtrees; tau
    T wc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) wc[k] = wcur[k * G + lane];
    const T dinv = wcur[NA * G + lane];
    T tau;
    {
      T PB[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) PB[k] = -P.rho * SE[k];
      if (has_hv) {
#pragma unroll
        for (int k = 0; k < 6; ++k) PB[k] -= shv[lane * 6 + k];
      }
      for (int s_ = 0; s_ < ncl; ++s_) {
        const T* c_; T m;
        cslot_of(s_, itok, c_, m);
#pragma unroll
        for (int k = 0; k < 6; ++k) PB[k] += m * (c_[C2_ATYW + k] - mu_eq * c_[C2_ATBW + k]);
      }
      tau = (w - mu_in * z) + dot6_halves(Sw, PB);
    }
    // ---- r' = W tau
    tail_sync();
#pragma unroll
    for (int k = 0; k < NA; ++k) xb[k * G + lane] = wc[k] * tau;
    tail_sync();
    T rn;
    {
      T a[FLAT_RED];
      static_for<0, FLAT_RED>([&](auto t) { a[t] = lds_abs<0>(field_here<16 * (t & 1), 16>(ra2[t >> 1], itok)); });   // (xb)
      T acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      pbuf[lane] = helper ? acc : T(0);
      tail_sync();
      T pp[FLAT_PART];
      static_for<0, FLAT_PART>([&](auto q) { pp[q] = pbuf[field_here<8 * (q & 3), 8>(prow4[q >> 2], itok)]; });
      if (helper) acc = T(0);
#pragma unroll
      for (int q = 0; q < FLAT_PART; ++q) acc += pp[q];
      rn = tau + acc;
      rbuf[lane] = rn;
      nbuf[lane] = dinv * rn;
    }
    tail_sync();
    // ---- nu = -W^T (Dinv r')
    T nui;
    {
      T nb_[NA];
      static_for<0, NA>([&](auto k) { nb_[k] = nbuf[field_here<8 * (k & 3), 8>(anc4[k >> 2], itok)]; });
      T acc = dinv * rn;
#pragma unroll
      for (int k = 0; k < NA; ++k) acc += wc[k] * nb_[k];
      nui = -acc;
    }
    // ---- v = J nu
    T vi[6], E[6];
    {
      T vw[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) vw[k] = Sw[k] * nui;
      if constexpr (PB2F1) flat_path_sum4<T, 6, 2>(xb, lane, pathA, anc4[(NA + 3) / 4 - 1], 0u, njmp, vw, itok);
      else flat_path_sum4<T, 6>(xb, lane, pathA, pathB, pathC, njmp, vw, itok);
      if (jcslot >= 0) {  // the constrained links' velocities at the world origin: the task constraints' update starts from them
#pragma unroll
        for (int k = 0; k < 6; ++k) cdi[jcslot * cs + C2_VC + k] = vw[k];
      }
      actinv_motion(R0, t0, vw, vi);
      force_of_motion(vw, E);
    }
    auto hrefv_of = [&](const T* vv, T* out) {   // H_ref v (link frame)
      if constexpr (HM >= 2) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const T* hr = (HM == 3 ? hrow : hmat) + k * 6;
          out[k] = ((hr[0] * vv[0] + hr[1] * vv[1]) + hr[2] * vv[2]) + ((hr[3] * vv[3] + hr[4] * vv[4]) + hr[5] * vv[5]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = hd[k] * vv[k];
      }
    };
    T hv6[6];
    hrefv_of(vi, hv6);
    // (signed values for the four maxima of every iteration, the rest formed where it is looked at: see k_flat2)
    T s_dy = T(0), s_av = T(0), s_ek = T(0);
    T l_dfis = T(0), l_dvis = T(0), s_dnu = T(0), s_dz = T(0), s_dw = T(0), s_prs = T(0);
    T l_dg = T(0), s_stf = T(0), s_dstf = T(0), l_dualv = T(0);
    T fi[6], si;
    {
      T SEn[6], SHn[6], Fw[6];
      // ---- the task constraints' update (two dependent exchanges through the constraint blocks) with the subtree sums of E,
      // prefix-sum differences in registers, in its shadow
      const bool first = my_iters == 1u && !resumed;  // (the first iteration of a fresh record: see k_flat2's load_instance)
      tail_sync();
      if (iscl) {  // row ckl of the lane's constraint: A v - b = AW^T v^w - b, dy, y (hxx:410-451)
        const T* col = ccb + C2_AWT + 6 * ckl;
        const T* vc = ccb + C2_VC;
        const T avk = ((col[0] * vc[0] + col[1] * vc[1]) + col[2] * vc[2]) + ((col[3] * vc[3] + col[4] * vc[4]) + col[5] * vc[5]);
        const T bk = ccb[C2_B + ckl];
        const T ek = avk - bk;
        const T dy = mu_eq * ek;
        s_dy = dy; s_ek = ek; s_av = avk;
        ccb[C2_Y + ckl] += dy;
      }
      {
        T Pk[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) Pk[c] = prefix64(E[c]);
        const int src = lane + (size > 0 ? size - 1 : 0);
        T E2[6], P2[6];
        if constexpr (HD) {
          T fl[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) fl[k] = mass * hv6[k];
          act_force(R0, t0, fl, E2);
#pragma unroll
          for (int k = 0; k < 6; ++k) P2[k] = prefix64(E2[k]);
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) xb[c * PATH_RS + lane] = Pk[c];   // (component-major: a lane's six words 48 B apart put lanes l and l + 8 of a store, l and l + 16 of a load, on one bank)
        if constexpr (HD) {
#pragma unroll
          for (int c = 0; c < 6; ++c) xb[(6 + c) * PATH_RS + lane] = P2[c];
        }
        tail_sync();
#pragma unroll
        for (int c = 0; c < 6; ++c) SEn[c] = (xb[c * PATH_RS + src] - Pk[c]) + E[c];
        if constexpr (HD) {
#pragma unroll
          for (int c = 0; c < 6; ++c) SHn[c] = (xb[(6 + c) * PATH_RS + src] - P2[c]) + E2[c];
        }
      }
      if (iscl) {  // the constraint's force at the world origin: AW y (the next FwdPass1's A^T y; this iteration's share of f)
        const T aw = awy_k();
        const T cw = first ? ccb[C2_CW + ckl] : T(0);
        ccb[C2_ATYW + ckl] = aw;
        ccb[C2_ATYF + ckl] = aw + cw;
      }
      tail_sync();
      {
        T dv6[6], gi[6], dg[6], dvr[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          dv6[k] = vi[k] - v[k];
          gi[k] = -mass * (P.rho * dv6[k] + hv6[k]);
        }
        if (has_hv) {
#pragma unroll
          for (int k = 0; k < 6; ++k) gi[k] += mass * hvl[k];
        }
        if (first && jcslot >= 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) gi[k] += cdi[jcslot * cs + C2_DLT + k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          dg[k] = gi[k] - g[k];
          dvr[k] = mass * hv6[k] + gi[k];
        }
        if (has_hv) {
#pragma unroll
          for (int k = 0; k < 6; ++k) dvr[k] -= mass * hvl[k];
        }
        l_dualv = hinf6(dvr);
        l_dvis = mass * hinf6(dv6);
        s_dnu = nui - nu;
        const T x = nui + inv_mu * w;
        const T zi = hmin(ubi, hmax(lbi, x));
        s_dz = zi - z;
        s_prs = nui - zi;
        const T dwi = mu_in * (nui - zi);
        s_dw = dwi;
        w = w + dwi; z = zi; nu = nui;
        l_dg = hinf6(dg);
#pragma unroll
        for (int k = 0; k < 6; ++k) { v[k] = vi[k]; g[k] = gi[k]; }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) Fw[k] = HD ? P.rho * (SEn[k] - SE[k]) + SHn[k] : (P.rho + href_s) * SEn[k] - P.rho * SE[k];
      if (has_hv) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Fw[k] -= shv[lane * 6 + k];
      }
      for (int s_ = 0; s_ < ncl; ++s_) {
        const T* c_; T m;
        cslot_of(s_, itok, c_, m);
#pragma unroll
        for (int k = 0; k < 6; ++k) Fw[k] += m * c_[C2_ATYF + k];
      }
      actinv_force(R0, t0, Fw, fi);
      si = dot6_halves(Sw, Fw);
#pragma unroll
      for (int k = 0; k < 6; ++k) SE[k] = SEn[k];
    }
    {
      T df[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) df[k] = fi[k] - f[k];
      l_dfis = mass * hinf6(df);
      si += w;
      s_stf = si;
      s_dstf = si - s;
      s = si;
#pragma unroll
      for (int k = 0; k < 6; ++k) f[k] = fi[k];
    }
    // ================= the scalars of the stopping logic, folded over the wavefront ===============================================
    // Main loop: four maxima (primal, dual, the two sides of the certificate's first test); the certificate's second test and the
    // tail solve's stopping rule need four more scalars, folded only when they are looked at.
    auto l_hrefv_now = [&]() -> T {   // (from the state this iteration left: see k_flat2)
      if constexpr (HD) { T hv[6]; hrefv_of(v, hv); return mass * hinf6(hv); }
      else return mass * tabs(href_s) * hinf6(v);
    };
    auto ub_lb_sums = [&](T& up, T& lm) {
      up = T(0); lm = T(0);
      if (iscl) {
        const T bk = ccb[C2_B + ckl];
        up = bk * tmax(s_dy, T(0));
        lm = bk * tmin(s_dy, T(0));
      }
      up += ubi * tmax(s_dw, T(0));
      lm += lbi * tmin(s_dw, T(0));
    };
    const bool fixed = P.mode & MODE_FIXED_ITERS;
    const bool in_tail = (status & ST_TAIL) != 0;
    const bool logic = !fixed && !in_tail;  // the main loop's stopping logic runs
    T primal = T(0), dual = T(0), dyqp = T(0), atdy = T(0), dx = T(0), dz = T(0), ubp = T(0), lbm = T(0);
    T ntol_p = T(0), ntol_d = T(0);
    bool have_norms = false;
    if (logic) {
      T in[4] = {hmaxa(s_ek, s_prs), hmax_a(l_dualv, s_stf), hmax_a(hmax_a(l_dfis, s_dy), s_dw), hmax_a(l_dg, s_dstf)}, r[4];
      if (LOIKB_QUIET32 && !LOG && MUR != 1 && quiet_f32(in, qth, iter, q_lim)) {   // (the quick look: see quiet_f32)
        ++iter;
        if (iter > q_lim) continue;
        goto next_iteration;
      }
      wave_fold4<0u>(lane, in, r);
      primal = r[0]; dual = r[1]; dyqp = r[2]; atdy = r[3];
      if (!LOG && P.tol_rel == T(0)) {
        // The common iteration decides nothing: not converged, the certificate's first test fails, mu stays where it is, not the
        // last iteration.  Five compares side by side and ONE branch; the stopping logic below is a chain of ~30 dependent
        // compare -> mask -> branch steps (900 cycles of a lone wavefront's 5400 per iteration) that only the other iterations walk.
        bool mu_stays;
        if constexpr (MUR == 1) {   // (OSQP's band, without its divisions and root: see k_flat2)
          T in2[4] = {hmaxa(s_av, nu), hmax_a(hmax(l_hrefv_now(), hinf6(g)), s_stf), T(0), T(0)}, r2[4];
          wave_fold4<0u>(lane, in2, r2);
          ntol_p = r2[0]; ntol_d = r2[1];
          have_norms = true;
          const T e_ = T(1e-10), np_ = tmax(ntol_p, bnorm_r) + e_, nd_ = tmax(ntol_d, P.Hv_inf_norm) + e_;
          const T A_ = primal * nd_, B_ = np_ * (dual + e_ * nd_);
          mu_stays = (A_ < T(24) * B_) & (A_ > T(0.05) * B_) & (mu >= T(1e-6)) & (mu <= T(1e6));
        } else {
          mu_stays = !(primal > T(10) * dual) & !(dual > T(10) * primal);
        }
        const bool quiet = !((primal < tol_abs_h) & (dual < tol_abs_h)) & !((iter > 0) & (atdy <= tpi_h * dyqp)) & mu_stays & (iter + 2 < max_iter_h);
        if (quiet) {
          ++iter;
          if (LOIKB_QUIET_SKIPS_TOP && iter <= q_lim) goto next_iteration;
          continue;
        }
      }
    }
    if ((MUR == 1 ? logic : P.tol_rel != T(0)) && !have_norms) {  // (uniform) relative tolerances -- and OSQP's rule -- need two more maxima
      T in2[4] = {hmaxa(s_av, nu), hmax_a(hmax(l_hrefv_now(), hinf6(g)), s_stf), T(0), T(0)}, r2[4];
      wave_fold4<0u>(lane, in2, r2);
      ntol_p = r2[0]; ntol_d = r2[1];
    }
    // ================= CheckConvergence, CheckFeasibility, UpdateMu, the tail solve's stopping rule (hpp:377-454, :271-319) ====
    const T mu_used = mu;
    const T tol_p = P.tol_abs + P.tol_rel * tmax(ntol_p, isc[FI_BNORM]);
    const T tol_d = P.tol_abs + P.tol_rel * tmax(ntol_d, P.Hv_inf_norm);
    const int itn = iter + 1;
    if constexpr (LOG) {
      if (!in_tail) {   // (the nine lists of k_pass_solve, loik_passes.hpp, in k_flat's order)
        T inl[4] = {tabs(s_ek), tabs(s_prs), tabs(s_stf), l_dualv}, rl[4];
        wave_fold4<0u>(lane, inl, rl);
        const int row = itn - 1;
        if (lane == 0 && row < Bf.log_cap) {
          const double vals[9] = {rl[0], rl[1], hmax(rl[0], rl[1]), rl[2], rl[3], hmax(rl[3], rl[2]), mu_used, P.mu_scale * mu_used, mu_used};
#pragma unroll
          for (int l = 0; l < 9; ++l) Bf.log[((size_t)l * Bf.log_B + lidx) * Bf.log_cap + row] = vals[l];
        }
      }
    }
    const bool conv = logic && (primal < tol_p) && (dual < tol_d);
    const bool feas_chk = logic && itn > 1;
    const bool c1 = atdy <= P.tol_primal_inf * dyqp;
    bool have_b = false;
    auto fold_b = [&]() {
      T l_up, l_lm;
      ub_lb_sums(l_up, l_lm);
      T in[4] = {l_up, l_lm, hmax_a(l_dvis, s_dnu), tabs(s_dz)}, r[4];
      wave_fold4<0x3u>(lane, in, r);
      ubp = r[0]; lbm = r[1]; dx = r[2]; dz = r[3];
      have_b = true;
    };
    if (in_tail || (feas_chk && c1)) fold_b();
    bool c2 = (ubp + lbm) <= P.tol_primal_inf * dyqp;
    const bool infeas = feas_chk && c1 && c2;
    const bool enter_tail = infeas && !conv;
    const bool upd = logic && !conv && !infeas;
    bool mu_up = upd && (primal > T(10) * dual), mu_dn = upd && !mu_up && (dual > T(10) * primal);
    if constexpr (MUR == 1) {   // OSQP's rule (update_mu): see k_flat2
      mu_up = mu_dn = false;
      if (upd) {
        T m2 = mu;
        int kk = kexp;
        if (update_mu<T>(MODE_MU_OSQP, primal, dual, tmax(ntol_p, isc[FI_BNORM]), tmax(ntol_d, P.Hv_inf_norm), m2, kk)) {
          mu = m2;
          inv_mu = T(1) / mu;
          ++nflip;
          kslot = -(1 << 30);
        }
      }
    }
    const bool tail_stop = !(dx >= P.tol_tail_solve || dz >= P.tol_tail_solve) || itn >= P.max_iter;  // (looked at with have_b only)
    const bool stop = conv || ((enter_tail || in_tail) && tail_stop) || ((upd || fixed) && itn + 1 >= P.max_iter);
    iter = itn;
    status |= (conv ? ST_CONVERGED : 0) | (infeas ? ST_PRIMAL_INF : 0) | (enter_tail ? ST_TAIL : 0) | (stop ? ST_DONE : 0);
    tail_it = enter_tail ? 0 : (in_tail ? tail_it + 1 : tail_it);
    mu = mu_up ? mu * T(10) : (mu_dn ? mu * T(0.1) : mu);
    if (mu_up || mu_dn) inv_mu = T(1) / mu;
    kexp += (mu_up ? 1 : 0) - (mu_dn ? 1 : 0);
    nflip += (mu_up || mu_dn) ? 1 : 0;
    done = stop;
    // ---- what the getters report: written when an instance stops (the certificate's scalars also when it enters the tail
    // solve: they keep the values of the last iteration that evaluated them) -------------------------------------------------
    T r1[8], r2[8];
    if (stop) {
      T in1[8] = {tabs(s_ek), tabs(s_prs), tabs(s_stf), l_dvis, tabs(s_dnu), l_dfis, tabs(s_dy), tabs(s_dw)};
      T in2[8] = {tabs(s_av), tabs(nu), l_hrefv_now(), hinf6(g), l_dualv, T(0), T(0), T(0)};
      wave_fold8<false>(lane, in1, r1);
      wave_fold8<false>(lane, in2, r2);
      if (!logic) { primal = hmax(r1[0], r1[1]); dual = hmax(r2[4], r1[2]); }  // (the tail solve / a fixed count did not fold them)
      if (!have_b) { fold_b(); c2 = (ubp + lbm) <= P.tol_primal_inf * dyqp; }  // (reported; c1 was false: no decision hung on it)
    }
    if (stop || enter_tail) {
      if (lane == 0) {
        isc[FI_PRIMAL] = primal; isc[FI_DUAL] = dual; isc[FI_DX] = dx; isc[FI_DZ] = dz; isc[FI_MULAST] = mu_used;
        if (logic) { isc[FI_TOLP] = tol_p; isc[FI_TOLD] = tol_d; }
        if (feas_chk) {
          isc[FI_C1] = (T)(c1 ? 1 : 0); isc[FI_C2] = (T)(c2 ? 1 : 0); isc[FI_DYQP] = dyqp; isc[FI_ATDY] = atdy; isc[FI_UBP] = ubp; isc[FI_LBM] = lbm;
        }
      }
    }
    if (stop) {
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) isc[FI_RED + k] = r1[k];
#pragma unroll
        for (int k = 0; k < 5; ++k) isc[FI_RED + 8 + k] = r2[k];
      }
      tail_sync();
      break;
    }
   }
    if constexpr (MUR == 1) {
      if (__builtin_expect(need_build, 0)) {   // (W / Dinv for the instance's own mu: flat_build_slot, one joint per lane; both slots are gone afterwards)
        T hb[21], at[21], Wc[NA], dv;
#pragma unroll
        for (int a_ = 0; a_ < 6; ++a_)
#pragma unroll
          for (int b2 = a_; b2 < 6; ++b2)
            hb[sym(a_, b2)] = mass * ((a_ == b2 ? P.rho : T(0)) + (P.href_tab ? P.href_tab[(size_t)(jl + 1) * HREF_ROW + 6 * a_ + b2] : P.Href[6 * a_ + b2]));
#pragma unroll
        for (int k = 0; k < 21; ++k) at[k] = T(0);
        if (jcslot >= 0) {
          const char* crec = ip + (size_t)(L.off_c + jcslot * L.crec) * pair_bytes<T>();
          for (int k = 0; k < 21; ++k)
            at[k] = a_shared ? Bf.uni[L.nc * 36 + jcslot * 21 + k]
                             : *reinterpret_cast<const T*>(crec + (size_t)(CP_ATA + k / 2) * pair_bytes<T>() + (k & 1) * sizeof(T));
        }
        flat_build_slot<NA, WAVE>(xb, lane, true, lane, L.nb, jd, reinterpret_cast<const TailTopo*>(aux[0]), reinterpret_cast<const int*>(aux[1]), fl,
                                  (has_hv_bits >> 8) & 0xFF, R0, t0, Sw, hb, at, mu, P.mu_scale, Wc, dv);
        wsel = 0;
#pragma unroll
        for (int k = 0; k < NA; ++k) wl[k * G + lane] = Wc[k];
        wl[NA * G + lane] = dv;
        kslot_o = -(1 << 30);
        kslot = kexp;
        if (lane == 0) atomicAdd(&Bf.counters[FLAT_COUNTERS_BUILT], 1u);
        tail_sync();
        continue;   // (back into the iteration loop with the instance the wavefront has)
      }
    }
    store_instance();
    if constexpr (SLICED) {
      if (requeue) q_push(lidx);
      else if (lane == 0) atomicAdd(q_retired, 1u);
    }
  }
  if (lane == 0) {
    atomicAdd(&Bf.counters[5], n_wave_iters);
    atomicAdd(&Bf.counters[6], n_slot_loads >> 16);
    atomicAdd(&Bf.counters[FLAT_COUNTERS_SLOT_HITS], n_slot_hits);
    atomicOr(&Bf.counters[LEAN_DECADES_SEEN], n_slot_loads & 0xFFFFu);
  }
}

}  // namespace loikb
