// bvq_fakequant_bwd_f16.hip -- the row-mapped backward kernel for float16 tensors with float16 arithmetic
// (explicit instantiations of launch_bwd: the long pole of the build, one translation unit per dtype family).
#include "bvq_fakequant_bwd.h"

namespace bvq {
template BVQ_LAUNCH_BWD(f16_t, f16_t);
}  // namespace bvq
