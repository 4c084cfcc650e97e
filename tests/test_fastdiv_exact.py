"""Exhaustive check of the claim behind the bf16 fast path of the quantizer kernels
(brevitas_amd/csrc/bvq_fakequant.h, DivBf16): for every bf16 scale s in [2^-14, 2^14] and EVERY
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


def f16_guard(q):
    """the kernels' test for `0 < |q| < 2^-14` (with a margin): such quotients take the IEEE division"""
    ab = q.view(np.uint32) & np.uint32(0x7fffffff)
    return (ab - np.uint32(1)) < np.uint32(0x38810000 - 1)


def f16_scales(full_binades, stride):
    out = []
    for e in range(-14, 14):
        step = 1 if e in full_binades else stride
        out.extend((1.0 + m / 1024.0) * 2.0 ** e for m in range(0, 1024, step))
    out.append(2.0 ** 14)
    return np.array(out, dtype=np.float32)


def test_reciprocal_multiply_is_exact_for_f16_over_f16_outside_the_subnormal_guard():
    """DivF16 (backward kernels): for float16 a and s, RN_f16(a * RN_f32(1/s)) == RN_f16(RN_f32(a / s)) whenever
    the fast quotient is 0 or at least 2^-14 in magnitude -- every float16 numerator, two full binades of scales
    and a sample of the others here; tests/test_gpu_fastdiv.py runs all 28 673 scales on the GPU."""
    a = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    nan = np.isnan(a)
    bad = guarded = 0
    with np.errstate(all='ignore'):
        for s in f16_scales(full_binades=(-14, 0), stride=41):
            assert np.float32(np.float16(s)) == s
            q = a * (np.float32(1.0) / s)
            fast = q.astype(np.float16).view(np.uint16)
            ref = (a / s).astype(np.float16).view(np.uint16)
            g = f16_guard(q)
            bad += int(np.count_nonzero((fast != ref) & ~nan & ~g))
            guarded += int(np.count_nonzero(g))
    assert bad == 0
    assert guarded > 0  # the guard is exercised (it is what makes the subnormal quotients right)


def refined_quotient(a, s):
    """DivF16R of brevitas_amd/csrc/bvq_fakequant.h emulated exactly: q0 = RN32(a * r) with r = RN32(1 / s),
    rem = fma(-q0, s, a), q = fma(rem, r, q0), then what v_div_fixup_f32 does for zero / infinite / NaN numerators
    and for the sign.  The two fmas are evaluated in float64, where the first is exact (35-bit product, operands a
    few binades apart) and the second is exact up to ONE rounding to 53 bits: the rare results that land exactly on
    a float32 midpoint after that rounding are redone in rational arithmetic."""
    from fractions import Fraction
    a = a.astype(np.float32)
    s = np.float32(s)
    r = np.float32(1.0) / s
    q0 = a * r
    a64, s64, r64 = a.astype(np.float64), np.float64(s), np.float64(r)
    rem64 = -q0.astype(np.float64) * s64 + a64
    rem = rem64.astype(np.float32)
    finite = np.isfinite(a) & (a != 0)
    assert np.all(rem.astype(np.float64)[finite] == rem64[finite])  # the remainder of a faithful quotient is exact
    t64 = rem.astype(np.float64) * r64 + q0.astype(np.float64)
    q = t64.astype(np.float32)
    mid = finite & ((t64.view(np.uint64) & np.uint64((1 << 29) - 1)) == np.uint64(1 << 28))
    for i in np.nonzero(mid)[0]:
        exact = Fraction(float(rem[i])) * Fraction(float(r)) + Fraction(float(q0[i]))
        lo = np.nextafter(q[i], np.float32(-np.inf)) if Fraction(float(q[i])) > exact else q[i]
        hi = np.nextafter(lo, np.float32(np.inf))
        dl, dh = exact - Fraction(float(lo)), Fraction(float(hi)) - exact
        if dl != dh:
            q[i] = lo if dl < dh else hi
        else:  # a true tie: to even
            q[i] = lo if (int(np.float32(lo).view(np.uint32)) & 1) == 0 else hi
    sign = np.signbit(a) ^ np.signbit(s)
    out = np.where(sign, -np.abs(q), np.abs(q)).astype(np.float32)
    out = np.where(a == 0, np.where(sign, np.float32(-0.0), np.float32(0.0)), out)
    out = np.where(np.isinf(a), np.where(sign, np.float32(-np.inf), np.float32(np.inf)), out)
    return np.where(np.isnan(a), np.float32(np.nan), out).astype(np.float32)


def test_refined_reciprocal_product_is_the_ieee_quotient_for_f16_over_f16():
    """DivF16R (float16 forward kernel): the float32 quotient itself, bit for bit, for every float16 numerator
    (zeros, subnormals, infinities and NaNs included) -- two full binades of scales and a sample of the others
    here; tests/test_gpu_fastdiv.py runs all 28 673 scales through the device code."""
    a = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    bad = 0
    with np.errstate(all='ignore'):
        for s in f16_scales(full_binades=(-14, 0), stride=53):
            got = refined_quotient(a, s)
            want = a / s
            same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
            bad += int(np.count_nonzero(~same))
    assert bad == 0
