"""Exhaustive check of the claim behind the bf16 fast path of the quantizer kernels
(brevitas_amd/csrc/bvq_fakequant.hip, DivBf16): for every bf16 scale s in [2^-14, 2^14] and EVERY
bf16 numerator a (all 65536 bit patterns),

    RN_bf16( RN_f32(a) * RN_f32(1 / s) )  ==  RN_bf16( RN_f32(a / s) )

bit for bit -- i.e. multiplying by the correctly rounded reciprocal gives the same bf16 as the
reference's true division followed by its rounding to bf16.  numpy float32 arithmetic is IEEE
(correctly rounded), the same as the device's v_mul_f32 / IEEE division.
"""
import numpy as np


def rn_bf16(f):
    """float32 array -> bf16 bit patterns (round to nearest even, NaNs canonicalised)"""
    u = f.view(np.uint32)
    nan = (u & 0x7fffffff) > 0x7f800000
    r = ((u + (0x7fff + ((u >> 16) & 1))) >> 16).astype(np.uint16)
    r[nan] = 0x7fc0
    return r


def test_reciprocal_multiply_is_exact_for_bf16_over_bf16():
    a = (np.arange(65536, dtype=np.uint32) << 16).view(np.float32)
    exps = np.arange(-14, 14)
    mant = 1.0 + np.arange(128) / 128.0
    scales = np.concatenate([(mant[None, :] * (2.0 ** exps)[:, None]).reshape(-1), [2.0 ** 14]]).astype(np.float32)
    assert np.all((scales.view(np.uint32) & 0xffff) == 0)  # all are bf16 values
    bad = 0
    with np.errstate(all='ignore'):
        for s in scales:
            r = np.float32(1.0) / s
            fast = rn_bf16(a * r)
            ref = rn_bf16(a / s)
            bad += int(np.count_nonzero(fast != ref))
    assert bad == 0, '%d of %d quotients differ' % (bad, a.size * scales.size)
