// The flat iteration kernels (k_flat2 / k_flat1, loik_flat2.hpp) are compiled in a translation unit of their own, loik_flat_kernels.hip,
// with two code-generation switches the rest of the library does not want (see loik_amd/_build.py, FLAT_FLAGS): no merging of neighbouring
// LDS accesses into ds_read2_b64 / ds_write2_b64 / 128-bit ones.  On MI355X a ds_read2_b64 occupies the CU's LDS pipe for 8.3 cycles
// where two ds_read_b64 take 4.8 (scripts/ubench/lds_rate.hip, profiles/r05_i_lds_rate.txt), and the loops of these kernels are LDS-pipe
// bound beside their fp64 issue (k_flat1: pipe busy 63 % of the whole launch).  k_fslots, k_solve, k_lean and the record movers keep the
// default code generation: their global loads want the merging (k_fslots +0.2 ms without).
// This header lists the instantiations the host launches (loik_host.hip: LOIKB_LAUNCH_FLAT2 / LOIKB_LAUNCH_FLAT1); X(...) is
// `template` in loik_flat_kernels.hip and `extern template` in loik_host.hip.
#pragma once
#include "loik_flat2.hpp"

#define LOIKB_FLAT2_ARGS                                                                                                            \
  (const loikb::Params<double>, const loikb::Bufs<double>, const loikb::JointDesc* __restrict__, const loikb::FlatLane* __restrict__, int, int, int, int*, \
   int, const double* __restrict__, int, int, int, double, int, int, int, double* __restrict__, int, const void* const* __restrict__)
#define LOIKB_FLAT1_ARGS                                                                                                            \
  (const loikb::Params<double>, const loikb::Bufs<double>, const loikb::JointDesc* __restrict__, const loikb::FlatLane* __restrict__, int, int, int, int*, \
   int, const double* __restrict__, int, int, int, double, int, int, int, const void* const* __restrict__)

// k_flat2<NA, WPE, SLICED, HM, LOG, MUR>
#define LOIKB_FLAT2_INSTANCES(X)                                                                                                    \
  X(2, false, 0, false, 0) X(2, true, 0, false, 0) X(3, false, 0, false, 0) X(3, true, 0, false, 0)                                 \
  X(2, false, 1, false, 0) X(2, true, 1, false, 0) X(2, false, 2, false, 0) X(2, true, 2, false, 0)                                 \
  X(2, false, 3, false, 0) X(2, true, 3, false, 0)                                                                                  \
  X(2, false, 0, true, 0) X(2, false, 2, true, 0) X(2, false, 3, true, 0)                                                           \
  X(2, false, 0, false, 1) X(2, true, 0, false, 1) X(2, false, 1, false, 1) X(2, false, 2, false, 1) X(2, false, 3, false, 1)      \
  X(2, false, 0, false, 2) X(2, true, 0, false, 2)
// k_flat1<NA, SLICED, HM, LOG, MUR>, for NA = FLAT_NA_SMALL and FLAT_MAXA
#define LOIKB_FLAT1_INSTANCES_NA(X, NA)                                                                                             \
  X(NA, false, 0, false, 0) X(NA, true, 0, false, 0) X(NA, false, 1, false, 0) X(NA, true, 1, false, 0)                             \
  X(NA, false, 2, false, 0) X(NA, true, 2, false, 0) X(NA, false, 3, false, 0) X(NA, true, 3, false, 0)                             \
  X(NA, false, 0, true, 0) X(NA, false, 2, true, 0) X(NA, false, 3, true, 0)                                                        \
  X(NA, false, 0, false, 1) X(NA, false, 2, false, 1) X(NA, false, 3, false, 1)
#define LOIKB_FLAT1_INSTANCES(X) LOIKB_FLAT1_INSTANCES_NA(X, loikb::FLAT_NA_SMALL) LOIKB_FLAT1_INSTANCES_NA(X, loikb::FLAT_MAXA)

#define LOIKB_FLAT2_DECL(WPE, SLICED, HM, LOG, MUR) \
  extern template __global__ void loikb::k_flat2<loikb::FLAT_NA_SMALL, WPE, SLICED, HM, LOG, MUR> LOIKB_FLAT2_ARGS;
#define LOIKB_FLAT2_DEF(WPE, SLICED, HM, LOG, MUR) \
  template __global__ void loikb::k_flat2<loikb::FLAT_NA_SMALL, WPE, SLICED, HM, LOG, MUR> LOIKB_FLAT2_ARGS;
#define LOIKB_FLAT1_DECL(NA, SLICED, HM, LOG, MUR) extern template __global__ void loikb::k_flat1<NA, SLICED, HM, LOG, MUR> LOIKB_FLAT1_ARGS;
#define LOIKB_FLAT1_DEF(NA, SLICED, HM, LOG, MUR) template __global__ void loikb::k_flat1<NA, SLICED, HM, LOG, MUR> LOIKB_FLAT1_ARGS;
