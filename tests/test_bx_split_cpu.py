"""The arithmetic behind the bf16-piece attention kernels (efficient-slowfast_amd/csrc/bx.h), restated in numpy: an
fp32 value is the exact sum of three round-to-nearest bf16 pieces, and the six products the kernels keep reproduce the
fp32 product to fp32 rounding level.  (The kernels themselves are checked against the oracle and fp64 in the -m gpu
tests; this pins the claim their header makes.)"""
import numpy as np


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32 — what v_cvt_pk_bf16_f32 does for finite inputs."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    p1 = bf16_rne(x)
    r = (x - p1).astype(np.float32)
    p2 = bf16_rne(r)
    r = (r - p2).astype(np.float32)
    return p1, p2, bf16_rne(r)


def _samples(n, seed):
    g = np.random.default_rng(seed)
    mant = g.uniform(1.0, 2.0, n) * g.choice([-1.0, 1.0], n)
    return (mant * np.exp2(g.integers(-60, 60, n))).astype(np.float32)


def test_three_pieces_are_exact():
    x = np.concatenate([_samples(200000, 1), np.float32([0.0, 1.0, -1.0, 3.0e38, 1e-30, 255.99998, 1.0000001])])
    p1, p2, p3 = split3(x)
    assert np.array_equal((p1.astype(np.float64) + p2 + p3).astype(np.float32), x)
    assert np.array_equal(p1.astype(np.float64) + p2 + p3, x.astype(np.float64))      # exact, not just to fp32
    nz = x != 0
    assert np.all(np.abs(p2[nz]) <= np.abs(x[nz]) * 2.0 ** -8)
    assert np.all(np.abs(p3[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_products_reach_fp32_rounding_level():
    a, b = _samples(200000, 2), _samples(200000, 3)
    a = (a / np.exp2(np.floor(np.log2(np.abs(a))))).astype(np.float32)  # exponents 0: no overflow in the check
    b = (b / np.exp2(np.floor(np.log2(np.abs(b))))).astype(np.float32)
    a1, a2, a3 = [v.astype(np.float64) for v in split3(a)]
    b1, b2, b3 = [v.astype(np.float64) for v in split3(b)]
    six = a1 * b3 + a3 * b1 + a2 * b2 + a1 * b2 + a2 * b1 + a1 * b1      # each term exact in fp32, summed here in fp64
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -23                                       # the dropped a2 b3 + a3 b2 + a3 b3
    assert np.sqrt(np.mean(rel ** 2)) <= 2.0 ** -26
    # every kept term is a product of two 8-bit significands: exact in fp32
    for u, v in ((a1, b3), (a3, b1), (a2, b2), (a1, b2), (a2, b1), (a1, b1)):
        assert np.array_equal((u * v).astype(np.float32).astype(np.float64), u * v)
