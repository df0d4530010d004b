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
