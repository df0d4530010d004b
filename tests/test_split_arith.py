"""The arithmetic claim behind csrc/wino_gemm_s3.hip, restated in numpy (no GPU): an fp32 value carried as three bf16 terms,
a product formed from its six leading partial products, fp32 accumulation.  The kernel's own error against float64 is measured
on the GPU (tests/test_gpu_parity.py::test_split_bf16_gemm_error_against_float64, tools/micro/gemm_s3_bench.hip)."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (the arithmetic of v_cvt_pk_bf16_f32 and of
    wino_gemm_s3.hip:s3_bf16_rne)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    x1 = bf16_rne(x)
    r1 = (x - x1).astype(np.float32)          # exact in fp32
    x2 = bf16_rne(r1)
    r2 = (r1 - x2).astype(np.float32)         # exact in fp32
    x3 = bf16_rne(r2)
    return x1, x2, x3


def test_three_bf16_terms_carry_an_fp32_value():
    rs = np.random.RandomState(0)
    x = (rs.randn(200000) * np.exp(rs.randn(200000) * 4)).astype(np.float32)       # many binades
    x1, x2, x3 = split3(x)
    # the subtractions inside the split are exact, so x1 + x2 + r2 == x; only the last rounding loses anything
    rec = x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64)
    err = np.abs(rec - x.astype(np.float64))
    assert np.all(err <= np.abs(x.astype(np.float64)) * 2.0 ** -24)
    assert np.median(err / np.abs(x)) < 2.0 ** -27
    # each term is a bf16 (low 16 bits of the fp32 pattern are zero) and the terms shrink by >= 2^-8 each
    for t in (x1, x2, x3):
        assert np.all((t.view(np.uint32) & 0xFFFF) == 0)
    nz = x2 != 0
    assert np.all(np.abs(x2[nz]) <= np.abs(x1[nz]) * 2.0 ** -8)


def test_six_partial_products_match_an_fp32_dot_product():
    """sum_k u_k v_k from the six products of weight >= 2^-16 per k, accumulated in float32 (smallest first, as the kernel
    orders them) -- error against float64 not above that of a plain float32 fmaf-style accumulation of u_k v_k."""
    rs = np.random.RandomState(1)
    K, R = 1024, 400
    u = (rs.randn(R, K) * 0.05).astype(np.float32)
    v = (rs.randn(R, K) * np.exp(rs.randn(R, K))).astype(np.float32)
    ref = np.sum(u.astype(np.float64) * v.astype(np.float64), axis=1)
    mag = np.sum(np.abs(u.astype(np.float64) * v.astype(np.float64)), axis=1)
    u1, u2, u3 = split3(u)
    v1, v2, v3 = split3(v)
    acc = np.zeros(R, dtype=np.float32)
    # the MFMA forms the 16 products of a k block exactly and adds them to the fp32 accumulator; model: float64 sum of a
    # block of 16 exact bf16 x bf16 products, rounded into the float32 accumulator once per product term and block
    for k0 in range(0, K, 16):
        s = slice(k0, k0 + 16)
        for a, b in ((u3, v1), (u2, v2), (u1, v3), (u2, v1), (u1, v2), (u1, v1)):
            blk = np.sum(a[:, s].astype(np.float64) * b[:, s].astype(np.float64), axis=1)
            acc = (acc.astype(np.float64) + blk).astype(np.float32)
    e_split = np.abs(acc.astype(np.float64) - ref) / mag
    f = np.zeros(R, dtype=np.float32)
    for k in range(K):                                   # the fp32 instruction's semantics: one fused multiply-add per k
        f = (f.astype(np.float64) + u[:, k].astype(np.float64) * v[:, k].astype(np.float64)).astype(np.float32)
    e_f32 = np.abs(f.astype(np.float64) - ref) / mag
    assert np.sqrt(np.mean(e_split ** 2)) <= 1.1 * np.sqrt(np.mean(e_f32 ** 2))
    assert e_split.max() < 5e-7
    # what the dropped products (u2 v3, u3 v2, u3 v3) amount to: below fp32's product rounding
    dropped = np.sum(np.abs(u2.astype(np.float64) * v3) + np.abs(u3.astype(np.float64) * v2) + np.abs(u3.astype(np.float64) * v3), axis=1) / mag
    assert dropped.max() < 2.0 ** -23


# ---- the fp16 two-term form (round 6; csrc/wino_gemm_s3.hip NT = 2, csrc/dt_internal.h: dt_h2_base) ---------------------------------
def h2_base(amax):
    """2^(14 - floor(log2 amax)), exponent field clamped to [27, 240] -- dt_internal.h: dt_h2_expfield / dt_h2_base."""
    bits = np.float32(amax).view(np.uint32)
    e = int((bits >> 23) & 0xFF)
    e = min(max(e, 27), 240)
    return np.float32(2.0) ** np.float32(14 - (e - 127)), np.float32(2.0) ** np.float32((e - 127) - 14)


def split2(xs):
    """x (already scaled) -> hi = fp16(x), lo = fp16(x - hi); the subtraction is exact in fp32 (wino_gemm_s3.hip: h2_split_pair)."""
    xs = np.asarray(xs, dtype=np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float32).astype(np.float16)
    return hi, lo


def test_h2_base_puts_the_maximum_below_fp16s_range():
    rs = np.random.RandomState(2)
    for amax in np.concatenate([np.exp(rs.randn(200) * 8), [1.0, 2.0 ** -20, 2.0 ** 30, 65504.0, 3.0e-31, 2.0 ** 113]]).astype(np.float32):
        base, inv = h2_base(amax)
        assert base * inv == 1.0                                     # exact inverse: the epilogue's multiply loses nothing
        s = np.float32(amax) * base
        assert s < 2.0 ** 15                                        # a factor 2 below 65504
        if amax >= 2.0 ** -100:                                     # below that the exponent field is clamped (27): the scale stops growing
            assert s >= 2.0 ** 14
    base, inv = h2_base(np.float32(0.0))                           # an all-zero tensor: clamped field, finite scale
    assert np.isfinite(base) and np.isfinite(inv) and base * inv == 1.0


def test_two_fp16_terms_carry_a_scaled_fp32_value():
    rs = np.random.RandomState(3)
    x = (rs.randn(400000) * np.exp(rs.randn(400000) * 1.5)).astype(np.float32)
    base, inv = h2_base(np.abs(x).max())
    xs = (x * base).astype(np.float32)                               # power of two: exact
    assert np.array_equal(xs * inv, x)
    hi, lo = split2(xs)
    assert np.all(np.isfinite(hi.astype(np.float32))) and np.abs(hi.astype(np.float32)).max() < 65504
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    err = np.abs(rec - xs.astype(np.float64))
    # 11 + 11 significand bits plus the sign of lo: at most one bit of fp32's 24 is lost while lo is a normal fp16; below that
    # the loss is absolute, half of fp16's subnormal step 2^-24 -- 2^-39 of the tensor's maximum
    assert np.all(err <= np.maximum(np.abs(xs.astype(np.float64)) * 2.0 ** -23, 2.0 ** -25))
    assert np.mean(err == 0) > 0.70                                  # most values survive exactly


def test_three_fp16_products_match_an_fp32_dot_product():
    """sum_k u_k v_k from lo_u hi_v + hi_u lo_v + hi_u hi_v (lo_u lo_v dropped), fp32 accumulation per 16-wide k block, smallest
    terms first as S3Prod<2> orders them -- error against float64 not above a plain fp32 accumulation's."""
    rs = np.random.RandomState(4)
    K, R = 1024, 400
    u = (rs.randn(R, K) * 0.05).astype(np.float32)
    v = (rs.randn(R, K) * np.exp(rs.randn(R, K))).astype(np.float32)
    ref = np.sum(u.astype(np.float64) * v.astype(np.float64), axis=1)
    mag = np.sum(np.abs(u.astype(np.float64) * v.astype(np.float64)), axis=1)
    bu, iu = h2_base(np.abs(u).max())
    bv, iv = h2_base(np.abs(v).max())
    uh, ul = split2(u * bu)
    vh, vl = split2(v * bv)
    acc = np.zeros(R, dtype=np.float32)
    for k0 in range(0, K, 16):
        s = slice(k0, k0 + 16)
        for a, b in ((ul, vh), (uh, vl), (uh, vh)):
            blk = np.sum(a[:, s].astype(np.float64) * b[:, s].astype(np.float64), axis=1)
            acc = (acc.astype(np.float64) + blk).astype(np.float32)
    assert np.all(np.isfinite(acc))                                  # 2^15 * 2^15 * K stays far inside fp32
    out = acc.astype(np.float64) * float(iu) * float(iv)
    e_split = np.abs(out - ref) / mag
    f = np.zeros(R, dtype=np.float32)
    for k in range(K):
        f = (f.astype(np.float64) + u[:, k].astype(np.float64) * v[:, k].astype(np.float64)).astype(np.float32)
    e_f32 = np.abs(f.astype(np.float64) - ref) / mag
    assert np.sqrt(np.mean(e_split ** 2)) <= 1.1 * np.sqrt(np.mean(e_f32 ** 2))
    assert e_split.max() < 5e-7
    dropped = np.sum(np.abs(ul.astype(np.float64) * vl.astype(np.float64)), axis=1) * float(iu) * float(iv) / mag
    assert dropped.max() < 2.0 ** -21


def test_an_outlier_costs_the_fp16_form_bits_gracefully():
    """One element 2^s times the rest moves the scale with it: the small values keep 22 - s bits (the curve
    tests/test_gpu_parity.py::test_h2_scale_follows_the_data measures on the device)."""
    rs = np.random.RandomState(5)
    x = rs.randn(100000).astype(np.float32)
    prev = 0.0
    for s in (0, 6, 10, 14):
        y = x.copy()
        y[0] = np.float32(2.0 ** s) * 4
        base, inv = h2_base(np.abs(y).max())
        hi, lo = split2(y * base)
        rec = (hi.astype(np.float64) + lo.astype(np.float64)) * float(inv)
        rel = np.sqrt(np.mean((rec[1:] - x[1:].astype(np.float64)) ** 2)) / np.sqrt(np.mean(x[1:].astype(np.float64) ** 2))
        assert rel < 2.0 ** (s - 19) and rel >= prev                # degrades by the outlier's exponent, never abruptly
        prev = rel
