"""CPU restatement (NumPy) of the operand cuts of the split-MFMA contractions, bit for bit:

  * two fp16 pieces under a power-of-two scale per row / column  (graphsage_amd/csrc/gs_split16.hip: gs_scale_exp, gs_cut16,
    split16_table_kernel, split16_rows_kernel) and the three-product contraction built on them;
  * three bf16 pieces by truncation                                (graphsage_amd/csrc/gs_split.hip: gs_split2).

TEST INFRASTRUCTURE ONLY (see oracle/graphsage_oracle.py header).  The reference has no counterpart: its pooling MLP is a
float32 tf.matmul (layers.py:104-116 via aggregators.py:176-179); these functions pin the ARITHMETIC the device uses in its
place, so that its error against float64 can be bounded on the CPU and the device's pieces checked bit-exactly on the GPU
(tests/test_split_arith.py, tests/test_split_gemm_gpu.py).
"""
import numpy as np


def scale_exp(mx):
    """gs_scale_exp: exponent e with mx * 2^e in [2^13, 2^14); 0 for an all-zero or non-finite row / column (a denormal
    maximum counts as 2^-127)."""
    mx = np.asarray(mx, dtype=np.float32)
    bits = mx.view(np.uint32)
    ex = ((bits >> np.uint32(23)) & np.uint32(0xFF)).astype(np.int64)
    e = 13 - (ex - 127)
    return np.where((ex == 0xFF) | (mx == 0), 0, e).astype(np.int32)


def cut16(x, e):
    """gs_cut16 on arrays: (h, m) float16 with x * 2^e = h + m + r, round to nearest even at both steps."""
    xs = np.ldexp(np.asarray(x, dtype=np.float32), np.asarray(e, dtype=np.int32)).astype(np.float32)
    h = xs.astype(np.float16)
    m = (xs - h.astype(np.float32)).astype(np.float16)
    return h, m


def cut_rows_f16(X):
    """split16_table_kernel: per-ROW exponents and the two pieces of X [rows, d]."""
    X = np.asarray(X, dtype=np.float32)
    e = scale_exp(np.abs(X).max(axis=1))
    h, m = cut16(X, e[:, None])
    return h, m, e


def cut_cols_f16(W):
    """split16_rows_kernel: per-COLUMN exponents and the two pieces of W [K, N]."""
    W = np.asarray(W, dtype=np.float32)
    e = scale_exp(np.abs(W).max(axis=0))
    h, m = cut16(W, e[None, :])
    return h, m, e


def matmul_two_pieces(X, W):
    """X . W the way gs_dense_fwd_rows_split16 forms it, with exact (float64) accumulation in place of the device's fp32
    accumulators: h h' + h m' + m h' of the scaled operands, scaled back by 2^-(e_row + e_col).  What differs from the exact
    product is exactly what the arithmetic gives up (the operands' last bit, the m m' products) -- the device adds its fp32
    accumulation rounding on top."""
    hx, mx, ex = cut_rows_f16(X)
    hw, mw, ew = cut_cols_f16(W)
    hx, mx, hw, mw = (a.astype(np.float64) for a in (hx, mx, hw, mw))
    acc = hx @ hw + (hx @ mw + mx @ hw)
    return np.ldexp(acc, -(ex[:, None].astype(np.int64) + ew[None, :].astype(np.int64)))


def cut_bf16x3(x):
    """gs_split2: (h, m, l) float32 arrays holding the top / middle / low 8 significant bits of x (truncation: h + m + l == x)."""
    x = np.asarray(x, dtype=np.float32)
    top = lambda v: (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    h = top(x)
    r = (x - h).astype(np.float32)
    m = top(r)
    l = (r - m).astype(np.float32)
    return h, m, l


def matmul_three_pieces(X, W):
    """X . W from the six kept piece products of gs_split.hip (hh, hm, mh, mm, hl, lh), exact accumulation."""
    hx, mx, lx = (a.astype(np.float64) for a in cut_bf16x3(X))
    hw, mw, lw = (a.astype(np.float64) for a in cut_bf16x3(W))
    return hx @ hw + (hx @ mw + mx @ hw + mx @ mw + hx @ lw + lx @ hw)
