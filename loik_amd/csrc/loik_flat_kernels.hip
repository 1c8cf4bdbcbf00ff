// loik_flat_kernels.hip -- the flat engine's iteration kernels (k_flat2: two lanes per joint, 17..32 joints; k_flat1: one lane per joint,
// 33..64 joints; loik_flat2.hpp) as a translation unit of their own, compiled with the LDS code-generation switches of
// loik_amd/_build.py::FLAT_FLAGS (why: loik_flat_inst.hpp).  The host side (loik_host.hip) launches them through `extern template`
// declarations of the same list; nothing else lives here.
// Math: /root/reference/include/loik/loik-loid-optimized.hxx:31-81, :102-163, :404-461 (as cited in loik_flat.hpp / loik_flat2.hpp).
#define LOIKB_FLAT_KERNELS_TU
#include "loik_flat_inst.hpp"

LOIKB_FLAT2_INSTANCES(LOIKB_FLAT2_DEF)
LOIKB_FLAT1_INSTANCES(LOIKB_FLAT1_DEF)

#ifdef LOIKB_TAIL_PROF
// (profile build of two units: this unit's copy of the phase counters -- the ones k_flat2 / k_flat1 write -- for the host unit's debug entry points.
//  which: 0 = g_tail_prof (wavefront 0 of the last launch), 1 = g_tail_prof_all (all wavefronts since the last reset; reset != 0 clears it))
int loikb_flat_prof_read(unsigned long long* out, int which, int reset)
{
  if (hipMemcpyFromSymbol(out, which ? HIP_SYMBOL(loikb::g_tail_prof_all) : HIP_SYMBOL(loikb::g_tail_prof), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
  if (which && reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(loikb::g_tail_prof_all), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
