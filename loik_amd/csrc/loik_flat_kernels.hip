// loik_flat_kernels.hip -- the flat engine's iteration kernels (k_flat2: two lanes per joint, 17..32 joints; k_flat1: one lane per joint,
// 33..64 joints; loik_flat2.hpp) as a translation unit of their own, compiled with the LDS code-generation switches of
// loik_amd/_build.py::FLAT_FLAGS (why: loik_flat_inst.hpp).  The host side (loik_host.hip) launches them through `extern template`
// declarations of the same list; nothing else lives here.
// Math: /root/reference/include/loik/loik-loid-optimized.hxx:31-81, :102-163, :404-461 (as cited in loik_flat.hpp / loik_flat2.hpp).
#define LOIKB_FLAT_KERNELS_TU
#include "loik_flat_inst.hpp"

LOIKB_FLAT2_INSTANCES(LOIKB_FLAT2_DEF)
LOIKB_FLAT1_INSTANCES(LOIKB_FLAT1_DEF)
