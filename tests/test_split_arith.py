"""CPU pins of the split-MFMA arithmetic (oracle/split_pieces.py): what the two-piece fp16 / three-piece bf16 operand cuts give up
against an exact product, bounded without a GPU.  The device's pieces are checked against the same functions bit for bit in
tests/test_split_gemm_gpu.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import split_pieces as sp  # noqa: E402


def _wild(rng, shape, lo, hi, axis):
    x = rng.normal(size=shape).astype(np.float32)
    sh = [1, 1]
    sh[axis] = shape[axis]
    return x * np.ldexp(1.0, rng.integers(lo, hi, size=sh)).astype(np.float32)


def test_scale_exponent_puts_the_largest_element_at_2_13():
    rng = np.random.default_rng(0)
    mx = np.abs(_wild(rng, (4000, 1), -100, 100, 0)).ravel()
    e = sp.scale_exp(mx)
    s = mx.astype(np.float64) * np.exp2(e.astype(np.float64))
    assert np.all((s >= 2.0 ** 13) & (s < 2.0 ** 14))
    assert sp.scale_exp(np.float32(0)) == 0 and sp.scale_exp(np.float32(np.inf)) == 0 and sp.scale_exp(np.float32(np.nan)) == 0
    assert sp.scale_exp(np.float32(1.0)) == 13 and sp.scale_exp(np.float32(3.999)) == 12 and sp.scale_exp(np.float32(2.0 ** 13)) == 0


def test_two_fp16_pieces_lose_at_most_one_fp32_ulp():
    """h keeps 11 significant bits, the residual has at most 12, m keeps 11 of them: |x 2^e - h - m| <= 2^-23 |x 2^e| (one fp32
    ulp) while m is a normal fp16; for elements more than 2^15 below the row's largest, half a subnormal fp16 ulp (2^-25 scaled)."""
    rng = np.random.default_rng(1)
    X = _wild(rng, (300, 257), -60, 60, 0)
    X[:, :64] *= np.ldexp(1.0, rng.integers(-40, 0, size=(300, 64))).astype(np.float32)
    X[7] = 0
    h, m, e = sp.cut_rows_f16(X)
    assert e[7] == 0 and np.all(h[7] == 0) and np.all(m[7] == 0)
    assert np.all(np.isfinite(h.astype(np.float32))) and np.all(np.isfinite(m.astype(np.float32)))
    xs = np.ldexp(X.astype(np.float64), e[:, None].astype(np.int64))
    err = np.abs(xs - h.astype(np.float64) - m.astype(np.float64))
    assert np.all(err <= np.maximum(np.abs(xs) * 2.0 ** -23, 2.0 ** -25))
    big = np.abs(xs) >= 2.0 ** -2
    # on average two bits better than the bound (2^-25.5), and 2^-11 for h alone
    assert np.mean(err[big] / np.abs(xs[big])) < 2.0 ** -25
    assert np.all(np.abs(xs - h.astype(np.float64))[big] <= np.abs(xs[big]) * 2.0 ** -11)


def test_three_bf16_pieces_are_exact():
    rng = np.random.default_rng(2)
    x = _wild(rng, (200, 100), -100, 100, 0)
    h, m, l = sp.cut_bf16x3(x)
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), x)
    for p in (h, m, l):
        assert np.all((p.view(np.uint32) & np.uint32(0xFFFF)) == 0)            # every piece is a bf16 value


def test_what_the_two_piece_product_gives_up_is_below_one_fp32_accumulation():
    """Error of the two-piece three-product contraction with EXACT accumulation -- i.e. only what the arithmetic gives up (last
    operand bits + the m m' products) -- against float64, next to an fp32 FMA chain's own rounding on the same dot products:
    K = 602 like the Reddit pooling MLP; plain and with rows / columns spread over 2^+-40."""
    rng = np.random.default_rng(3)
    K, rows, cols = 602, 96, 64
    for wild in (False, True):
        X = rng.normal(size=(rows, K)).astype(np.float32)
        W = (rng.normal(size=(K, cols)) * 0.1).astype(np.float32)
        if wild:
            X = _wild(rng, (rows, K), -40, 40, 0)
            X[:, :200] *= np.ldexp(1.0, rng.integers(-12, 0, size=(rows, 200))).astype(np.float32)
            W = W * np.ldexp(1.0, rng.integers(-30, 30, size=(1, cols))).astype(np.float32)
        want = X.astype(np.float64) @ W.astype(np.float64)
        mag = np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64)
        e2 = (np.abs(sp.matmul_two_pieces(X, W) - want) / mag).max()
        e3 = (np.abs(sp.matmul_three_pieces(X, W) - want) / mag).max()
        # an fp32 FMA chain over k (the float32 yardstick): sequential float32 accumulation
        acc = np.zeros((rows, cols), np.float32)
        for k in range(K):
            acc = (acc + X[:, k:k + 1] * W[k:k + 1, :]).astype(np.float32)
        e32 = (np.abs(acc.astype(np.float64) - want) / mag).max()
        print("wild=%s: |error| / (|x|.|w|): two fp16 pieces %.3g, three bf16 pieces %.3g, fp32 chain %.3g" % (wild, e2, e3, e32))
        assert e3 < 2.0 ** -24                       # three pieces: the dropped ml, lm, ll terms, <= 3 * 2^-24 per product
        assert e2 < 2.0 ** -22                       # two pieces: worst case per product 2^-21; a random walk in a sum
        assert e2 < 0.25 * e32                       # ... well below what fp32 accumulation itself loses on the same sums
