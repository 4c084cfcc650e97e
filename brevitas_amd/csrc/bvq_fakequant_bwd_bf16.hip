// bvq_fakequant_bwd_bf16.hip -- the row-mapped backward kernel for bf16 tensors with bf16 arithmetic
// (explicit instantiations of launch_bwd: the long pole of the build, one translation unit per dtype family).
#include "bvq_fakequant_bwd.h"

namespace bvq {
template BVQ_LAUNCH_BWD(bf16_t, bf16_t);
}  // namespace bvq
