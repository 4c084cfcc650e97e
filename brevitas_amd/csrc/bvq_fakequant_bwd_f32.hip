// bvq_fakequant_bwd_f32.hip -- the row-mapped backward kernel for float32 arithmetic (float32, and 16-bit tensors beside a float32 scale)
// (explicit instantiations of launch_bwd: the long pole of the build, one translation unit per dtype family).
#include "bvq_fakequant_bwd.h"

namespace bvq {
template BVQ_LAUNCH_BWD(float, float);
template BVQ_LAUNCH_BWD(bf16_t, float);
template BVQ_LAUNCH_BWD(f16_t, float);
}  // namespace bvq
