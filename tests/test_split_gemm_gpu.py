"""-m gpu: the split-MFMA contractions (graphsage_amd/csrc/gs_split.hip): fp32 operands cut into three bf16 pieces, six bf16
MFMAs per product tile, fp32 accumulation.  Claims checked here:
  * the cut is EXACT: h + m + l == x bit for bit, for every fp32 value (normal range), in the layout the kernel reads;
  * the contraction has the accuracy of an fp32 FMA chain: its error against fp64 is of the size of the fp32 MFMA kernel's
    own error against fp64 (both ~1e-7 of the row's rms at K = 602), five hundred times below north_star's 1e-4;
  * shapes of the pooling MLP on distinct ids: ragged counts read from the device, K with a masked last stage, N not a multiple of
    the column tile, bias + relu / identity, the split-K tail round;
    (the register-streaming layer-0 form of this arithmetic and its tests were removed in round 6: benchmarks/variants/README.md)
  * run to run bit-identical."""
import numpy as np
import pytest
import torch

from graphsage_amd import ops
from graphsage_amd.ops import Mat

pytestmark = pytest.mark.gpu


def _i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


def _pieces(w3, K, N):
    """int32 device tensor of gs_split_rows -> float64 arrays h, m, l of shape [N, Kp] (the pieces of W^T)."""
    KG = w3.numel() // (12 * N)                           # groups of 8 k (zero-padded beyond K)
    assert KG >= ((K + 15) // 16) * 2 and KG % 4 == 0
    raw = w3.cpu().numpy().view(np.uint16).reshape(KG, 3, N, 8)
    as_f32 = (raw.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return [as_f32[:, p].transpose(1, 0, 2).reshape(N, KG * 8) for p in range(3)]


def test_split_rows_pieces_are_exact(dev):
    rng = np.random.default_rng(0)
    K, N = 602, 128
    W = rng.normal(size=(K, N)).astype(np.float32)
    W[0, :8] = [0.0, -0.0, 1.0, -1.0, 2.0 ** -100, -3.0e38, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24]    # edge values
    W[1] = np.ldexp(rng.normal(size=N), rng.integers(-60, 60, size=N)).astype(np.float32)       # wide exponent range
    Wd = Mat.from_numpy(W, dev)
    w3 = ops.split_rows(Wd)
    torch.cuda.synchronize()
    h, m, l = _pieces(w3, K, N)
    total = h + m + l                                               # float64: exact sum of three bf16 values
    assert np.array_equal(total[:, :K], W.T.astype(np.float64))     # nothing is lost
    assert np.all(total[:, K:] == 0)                                # zero padding up to the next multiple of 16
    # the pieces are what the header says: top / middle / low 8 significant bits (truncation: same sign, decreasing size)
    assert np.all(np.abs(m) <= np.abs(h) * 2.0 ** -7) and np.all(np.abs(l) <= np.abs(h) * 2.0 ** -15)
    assert np.all((np.sign(m) == np.sign(h)) | (m == 0)) and np.all((np.sign(l) == np.sign(h)) | (l == 0))


CASES = [
    # n, d, out, two, act, bias, gathered, riders
    (5632, 602, 128, True, ops.ACT_RELU, False, True, True),        # the Reddit step's layer 0
    (11484, 602, 128, True, ops.ACT_RELU, False, True, False),      # the unsupervised step's
    (5632, 256, 128, True, ops.ACT_RELU, False, True, False),       # RMAT's F = 256 (8 whole stages)
    (5633, 601, 128, True, ops.ACT_IDENTITY, True, False, True),    # ragged rows, 25 valid k in the masked last stage
    (17000, 608, 64, False, ops.ACT_RELU, True, False, False),      # one term (GCN form), half a column tile, K = 608 exactly
    (4100, 250, 192, True, ops.ACT_RELU, True, True, False),        # two column tiles per term (the second half empty), K tail
    (2050, 50, 100, True, ops.ACT_RELU, True, True, False),         # PPI's F = 50 (two stages); N = 100
    (33, 7, 8, False, ops.ACT_IDENTITY, False, False, False),       # one partial stage, one partial tile
    (5632, 602, 256, False, ops.ACT_RELU, False, False, False),     # GCN at Reddit's shape (dims 2 x 128)
]


@pytest.mark.parametrize("n,d,out,two,act,bias,gathered,riders", CASES)
def test_sage_dense_fwd_tiled3(dev, n, d, out, two, act, bias, gathered, riders):
    """gs_sage_dense_fwd_tiled3 (layer 0 of a mean / GCN step on the bf16 matrix pipe, three-piece operands, LDS-tiled 64 x 128
    workgroup tiles, aggregators.py:51-64 / :110-116) vs fp64 NumPy: the accuracy of an fp32 FMA chain, NaN in every pad column,
    gathered / dense self rows, one and two terms, bias + relu / identity, gather jobs riding, bit-identical reruns."""
    rng = np.random.default_rng(n + d + out)
    Nn = 6000
    X = rng.normal(size=(Nn + 1, d)).astype(np.float32); X[Nn] = 0
    self_m, mean = rng.normal(size=(n, d)).astype(np.float32), rng.normal(size=(n, d)).astype(np.float32)
    self_ids = rng.integers(0, Nn + 1, size=n).astype(np.int32)
    if gathered:
        self_m = X[self_ids]
    Ws, Wn = (rng.normal(size=(d, out)) * 0.1).astype(np.float32), (rng.normal(size=(d, out)) * 0.1).astype(np.float32)
    b = (rng.normal(size=((2 if two else 1) * out,)) * 0.1).astype(np.float32) if bias else None
    Xd = Mat.from_numpy(X, dev, 32)
    sd, md = Mat.from_numpy(self_m, dev, 32), Mat.from_numpy(mean, dev, 4)      # the means: ld = round_up(d, 4) only
    for mat in (Xd, sd, md):
        if mat.ld > d:
            mat.buf[:, d:] = float("nan")                   # nothing beyond K may leak
    Wsd, Wnd = Mat.from_numpy(Ws, dev), Mat.from_numpy(Wn, dev)
    w3s, w3n = ops.split_rows(Wsd), ops.split_rows(Wnd)
    bd = torch.from_numpy(b).to(dev) if bias else None
    idx = rng.integers(0, Nn + 1, size=(700, 25)).astype(np.int32)
    idx_d, sid_d = _i32(idx.reshape(-1), dev), _i32(self_ids, dev)
    Xclean = Mat.from_numpy(X, dev, 32)                      # the riders' table (NaN pads would enter their float4 sums)
    outs = []
    for rep in range(2):
        outm = Mat.zeros(n, (2 if two else 1) * out, dev)
        outm.buf.fill_(float("nan"))
        g_out = Mat.zeros(700, d, dev)
        jobs = [ops.gather_job(Xclean, idx_d, 700, 25, g_out)] if riders else []
        if two:
            ops.sage_dense_fwd_tiled3(Xd if gathered else sd, sid_d if gathered else None, md, n, Wsd, Wnd, out, act, bd, outm, jobs)
        else:
            ops.sage_dense_fwd_tiled3(None, None, md, n, None, Wnd, out, act, bd, outm, jobs)
        torch.cuda.synchronize()
        outs.append(outm.numpy())
        if riders:
            np.testing.assert_allclose(g_out.numpy(), X[idx].mean(axis=1), rtol=1e-4, atol=1e-4)
    assert np.array_equal(outs[0], outs[1])                  # deterministic
    want_n = mean.astype(np.float64) @ Wn.astype(np.float64)
    want = np.concatenate([self_m.astype(np.float64) @ Ws.astype(np.float64), want_n], axis=1) if two else want_n
    if bias:
        want = want + b
    pre = want.copy()
    if act == ops.ACT_RELU:
        want = np.maximum(want, 0)
    got = outs[0].astype(np.float64)
    assert np.isfinite(got).all()
    # fp32 accuracy: the error is a few fp32 roundings of the row's scale, like a float32 NumPy matmul's
    rms = np.sqrt((pre * pre).mean())
    err = np.abs(got - want).max() / rms
    f32 = mean @ Wn
    f32 = np.concatenate([self_m @ Ws, f32], axis=1) if two else f32
    if bias:
        f32 = f32 + b
    if act == ops.ACT_RELU:
        f32 = np.maximum(f32, 0)
    err32 = np.abs(f32.astype(np.float64) - want).max() / rms
    assert err <= max(4 * err32, 4e-7), (err, err32)
    assert err < 4e-6, err


def test_tiled3_fwd_is_as_accurate_as_the_fp32_mfma_kernel(dev):
    """Same operands through gs_sage_dense_fwd_stream (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains) and through the three-piece
    tiled form: both errors against fp64 are fp32 rounding noise of the same size."""
    rng = np.random.default_rng(9)
    n, d, out = 5632, 602, 128
    Xs, Mn = rng.normal(size=(n, d)).astype(np.float32), rng.normal(size=(n, d)).astype(np.float32)
    Ws, Wn = (rng.normal(size=(d, out)) * 0.1).astype(np.float32), (rng.normal(size=(d, out)) * 0.1).astype(np.float32)
    sd, md = Mat.from_numpy(Xs, dev, 32), Mat.from_numpy(Mn, dev, 32)
    Wsd, Wnd = Mat.from_numpy(Ws, dev), Mat.from_numpy(Wn, dev)
    a, b = Mat.zeros(n, 2 * out, dev), Mat.zeros(n, 2 * out, dev)
    ops.sage_dense_fwd_stream(sd, None, md, n, Wsd, Wnd, out, ops.ACT_IDENTITY, None, a, [])
    ops.sage_dense_fwd_tiled3(sd, None, md, n, Wsd, Wnd, out, ops.ACT_IDENTITY, None, b, [])
    torch.cuda.synchronize()
    want = np.concatenate([Xs.astype(np.float64) @ Ws, Mn.astype(np.float64) @ Wn], axis=1)
    rms = np.sqrt((want * want).mean())
    e_fp32 = np.abs(a.numpy() - want).max() / rms
    e_split = np.abs(b.numpy() - want).max() / rms
    print("max error / rms vs fp64: fp32 MFMA kernel %.3g, tiled three-piece kernel %.3g" % (e_fp32, e_split))
    assert e_split <= 2 * e_fp32 and e_split < 4e-6


def _wgrad_desc(A, a_idx, dZ, col0, out, n, d, ns, dev, a_rows=0):
    from graphsage_amd import _lib
    ld_slab = (out + 3) & ~3
    sl = torch.full((ns * d * ld_slab,), float("nan"), device=dev)
    q = _lib.WgradDesc()
    q.A, q.a_idx, q.dZ, q.slabs = A.ptr, ops.ptr(a_idx), dZ.ptr, sl.data_ptr()
    q.lda, q.ldz, q.ld_slab, q.n = A.ld, dZ.ld, ld_slab, n
    q.d, q.col0, q.out_dim, q.n_slabs, q.a_rows = d, col0, out, ns, a_rows
    return q, sl, ld_slab


def test_tiled3_wgrad_is_as_accurate_as_the_fp32_mfma_kernel(dev):
    """Same operands through gs_dense_wgrad_grouped_stream (fp32 MFMA) and gs_dense_wgrad_grouped_tiled3 (three bf16 pieces): both
    errors against fp64 are fp32 rounding noise of the same size; NaN in the operands' pad columns reaches no output."""
    import ctypes
    from graphsage_amd import _lib
    rng = np.random.default_rng(19)
    n, d, out = 5632, 602, 128
    A = rng.normal(size=(n, d)).astype(np.float32)
    dZ = (rng.normal(size=(n, 2 * out)) * 0.1).astype(np.float32)
    Ad, Zd = Mat.from_numpy(A, dev, 32), Mat.from_numpy(dZ, dev, 32)
    Ad.buf[:, d:] = float("nan")
    want = A.astype(np.float64).T @ dZ[:, out:].astype(np.float64)
    errs = {}
    jn = (_lib.GatherDesc * 1)()
    for entry, ns in (("gs_dense_wgrad_grouped_stream", 22), ("gs_dense_wgrad_grouped_tiled3", 11)):
        q, sl, ld_slab = _wgrad_desc(Ad, None, Zd, out, out, n, d, ns, dev)
        arr = (_lib.WgradDesc * 1)(q)
        ops.call(entry, ctypes.addressof(arr), 1, ctypes.addressof(jn), 0, ops.current_stream())
        torch.cuda.synchronize()
        got = sl.cpu().numpy().reshape(ns, d, ld_slab)[:, :, :out].astype(np.float64).sum(axis=0)
        assert np.isfinite(got).all(), entry
        errs[entry] = np.abs(got - want).max() / np.sqrt((want * want).mean())
    print("max error / rms vs fp64:", errs)
    assert errs["gs_dense_wgrad_grouped_tiled3"] <= 2 * errs["gs_dense_wgrad_grouped_stream"] and errs["gs_dense_wgrad_grouped_tiled3"] < 4e-6


@pytest.mark.parametrize("n,d,out,col0,ns,gathered", [
    (5632, 602, 128, 128, 11, True),        # the Reddit step's layer-0 self term: 512-row slices, ragged last m tile
    (11484, 602, 128, 0, 12, True),         # the unsupervised step's: 960-row slices, the last one 924 rows (a partial last stage)
    (1000, 50, 256, 0, 1, False),           # PPI's F = 50 (one m tile, columns clamped), two n tiles, one 1000-row slice (1024 > n)
    (77, 256, 41, 0, 2, False),             # head-shaped: 41 columns, 64-row slices, the second one 13 rows
    (33, 1, 7, 4, 1, False),                # a bias gradient: A = ones [n, 1] (lda 4), 7 columns at an offset
    (300, 130, 200, 0, 3, True)])           # three m tiles (the last 2 rows), two n tiles (the last 72 columns), 128-row slices (44 in the last)
def test_dense_wgrad_grouped_tiled3_shapes(dev, n, d, out, col0, ns, gathered):
    """gs_dense_wgrad_grouped_tiled3 on ragged shapes vs fp64: every slab element inside [d, out] written, nothing outside it."""
    import ctypes
    from graphsage_amd import _lib
    rng = np.random.default_rng(n + d + out)
    rows = 4000 if gathered else n
    T = np.ones((rows, 1), np.float32) if d == 1 else rng.normal(size=(rows, d)).astype(np.float32)
    aidx = rng.integers(0, rows, size=n).astype(np.int32) if gathered else None
    A = T[aidx] if gathered else T
    dZ = (rng.normal(size=(n, col0 + out)) * 0.1).astype(np.float32)
    Td, Zd = Mat.from_numpy(T, dev), Mat.from_numpy(dZ, dev)
    if Td.ld > d:
        Td.buf[:, d:] = float("nan")
    if Zd.ld > col0 + out:
        Zd.buf[:, col0 + out:] = float("nan")
    idx_d = _i32(aidx, dev) if gathered else None
    q, sl, ld_slab = _wgrad_desc(Td, idx_d, Zd, col0, out, n, d, ns, dev)
    arr = (_lib.WgradDesc * 1)(q)
    jn = (_lib.GatherDesc * 1)()
    ops.call("gs_dense_wgrad_grouped_tiled3", ctypes.addressof(arr), 1, ctypes.addressof(jn), 0, ops.current_stream())
    torch.cuda.synchronize()
    raw = sl.cpu().numpy().reshape(ns, d, ld_slab)
    assert np.isfinite(raw[:, :, :out]).all()
    assert np.isnan(raw[:, :, out:]).all()               # pad columns of a slab row are not touched
    want = A.astype(np.float64).T @ dZ[:, col0:].astype(np.float64)
    got = raw[:, :, :out].astype(np.float64).sum(axis=0)
    rms = np.sqrt((want * want).mean()) + 1e-30
    assert np.abs(got - want).max() / rms < 1e-5, np.abs(got - want).max() / rms
    # deterministic
    sl2 = sl.clone()
    ops.call("gs_dense_wgrad_grouped_tiled3", ctypes.addressof(arr), 1, ctypes.addressof(jn), 0, ops.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(sl), torch.nan_to_num(sl2))


def test_dense_wgrad_grouped_tiled3_refuses_slices_beyond_its_row_list(dev):
    """A slice's source rows live in a 1024-entry LDS list -- for dense problems too: longer slices are refused, not mis-addressed."""
    import ctypes
    from graphsage_amd import _lib
    n, d, out = 4100, 64, 64
    Ad, Zd = Mat.zeros(n, d, dev), Mat.zeros(n, out, dev)
    jn = (_lib.GatherDesc * 1)()
    q, sl, _ = _wgrad_desc(Ad, None, Zd, 0, out, n, d, 4, dev)             # 4 slices of 1056 rows
    arr = (_lib.WgradDesc * 1)(q)
    with pytest.raises(_lib.GraphsageAmdError, match="at most 1024 rows"):
        ops.call("gs_dense_wgrad_grouped_tiled3", ctypes.addressof(arr), 1, ctypes.addressof(jn), 0, ops.current_stream())
    q, sl, _ = _wgrad_desc(Ad, None, Zd, 0, out, n, d, 5, dev)             # 5 slices of 832 rows: fine
    arr = (_lib.WgradDesc * 1)(q)
    ops.call("gs_dense_wgrad_grouped_tiled3", ctypes.addressof(arr), 1, ctypes.addressof(jn), 0, ops.current_stream())
    torch.cuda.synchronize()
    assert float(sl.abs().max()) == 0.0


@pytest.mark.parametrize("n_max,count,d,out,act,bias", [
    (6000, 5000, 602, 512, ops.ACT_RELU, True),      # the pooling MLP's shape (fewer rows), device-side row count
    (300, 300, 602, 512, ops.ACT_RELU, True),        # three row tiles, the last one ragged
    (1000, 777, 50, 512, ops.ACT_RELU, True),        # PPI's F = 50: two stages, the second with a 2-k tail
    (700, 700, 40, 100, ops.ACT_IDENTITY, False),    # ceil(K/16) odd: the last stage reads two k-groups only; N = 100
    (260, 130, 256, 1024, ops.ACT_RELU, True)])      # model_size "big": hidden 1024
def test_dense_fwd_rows_split(dev, n_max, count, d, out, act, bias):
    """gs_dense_fwd_rows_split (LDS-tiled split-MFMA GEMM on gathered rows with a device-side row count) vs fp64 and vs the
    fp32 MFMA kernel gs_dense_fwd_rows_dev: same accuracy class; rows beyond the count are not written; deterministic."""
    rng = np.random.default_rng(n_max + d + out)
    Nn = 5000
    X = rng.normal(size=(Nn + 1, d)).astype(np.float32); X[Nn] = 0
    ids = rng.integers(0, Nn + 1, size=n_max).astype(np.int32)
    W = (rng.normal(size=(d, out)) * 0.1).astype(np.float32)
    b = (rng.normal(size=(out,)) * 0.1).astype(np.float32) if bias else None
    Xd, Wd = Mat.from_numpy(X, dev, 32), Mat.from_numpy(W, dev)
    if Xd.ld > d:
        Xd.buf[:, d:] = float("nan")
    w3 = ops.split_rows(Wd)
    ids_d = _i32(ids, dev)
    cnt = _i32(np.asarray([count]), dev)
    bd = torch.from_numpy(b).to(dev) if bias else None
    outs = []
    for rep in range(2):
        o = Mat.zeros(n_max, out, dev)
        o.buf.fill_(-7.0)
        ops.call("gs_dense_fwd_rows_split", Xd.ptr, Xd.ld, ops.ptr(ids_d), d, n_max, ops.ptr(cnt), ops.ptr(w3), out, act,
                 ops.ptr(bd), o.ptr, o.ld, ops.current_stream())
        torch.cuda.synchronize()
        outs.append(o.numpy())
    assert np.array_equal(outs[0], outs[1])
    if n_max > 2048:                                         # the fp32 MFMA kernel of the same call (tiled kernels: n_max > 2048)
        ref = Mat.zeros(n_max, out, dev)
        Xc = Mat.from_numpy(X, dev, 32)
        ops.call("gs_dense_fwd_rows_dev", Xc.ptr, Xc.ld, ops.ptr(ids_d), d, n_max, ops.ptr(cnt), Wd.ptr, Wd.ld, out, act,
                 ops.ptr(bd), ref.ptr, ref.ld, ops.current_stream())
        torch.cuda.synchronize()
        ref = ref.numpy()
    else:                                                    # a float32 NumPy matmul as the fp32 yardstick
        ref = X[ids] @ W
        if bias:
            ref = ref + b
        if act == ops.ACT_RELU:
            ref = np.maximum(ref, 0)
    want = X[ids].astype(np.float64) @ W.astype(np.float64)
    if bias:
        want = want + b
    pre = want.copy()
    if act == ops.ACT_RELU:
        want = np.maximum(want, 0)
    got = outs[0].astype(np.float64)
    assert np.all(outs[0][count:] == -7.0)                                    # rows beyond the device-side count untouched
    rms = np.sqrt((pre[:count] ** 2).mean())
    err = np.abs(got[:count] - want[:count]).max() / rms
    err32 = np.abs(ref[:count].astype(np.float64) - want[:count]).max() / rms
    print("d=%d: max error / rms vs fp64: fp32 MFMA kernel %.3g, split kernel %.3g" % (d, err32, err))
    assert err <= max(2.5 * err32, 5e-7) and err < 4e-6
    # the same call with a workspace (ABI 8): the last round of the wide form's workgroups is cut along K (these shapes have fewer
    # tiles than CUs: EVERY tile takes that path) -- deterministic, same accuracy class, rows beyond the count untouched
    ws = torch.full((ops.split_tiled_ws_words(),), float("nan"), dtype=torch.float32, device=dev)
    outs_ws = []
    for rep in range(2):
        o = Mat.zeros(n_max, out, dev)
        o.buf.fill_(-7.0)
        ops.call("gs_dense_fwd_rows_split_ws", Xd.ptr, Xd.ld, ops.ptr(ids_d), d, n_max, ops.ptr(cnt), ops.ptr(w3), out, act,
                 ops.ptr(bd), o.ptr, o.ld, ops.ptr(ws), 4 * ws.numel(), ops.current_stream())
        torch.cuda.synchronize()
        outs_ws.append(o.numpy())
    assert np.array_equal(outs_ws[0], outs_ws[1])
    assert np.all(outs_ws[0][count:] == -7.0)
    err_ws = np.abs(outs_ws[0][:count].astype(np.float64) - want[:count]).max() / rms
    print("      with the split-K tail round: %.3g" % err_ws)
    assert err_ws <= max(2.5 * err32, 5e-7) and err_ws < 4e-6


def test_dense_fwd_rows_split_tail_round_mixes_whole_and_split_tiles(dev):
    """More tiles than CUs: whole tiles in the full rounds, K parts + fix-up launch for the rest (the pooling MLP's situation: 1300
    tiles on 256 CUs).  Against fp64 and against the call without a workspace: equal bits on the rows of whole tiles."""
    n_cu = ops.split_tiled_ws_words() // (128 * 256)
    d, out = 96, 512
    count = 128 * (n_cu // 2 + 9) - 37                      # tiles = 2 (n_cu / 2 + 9) = n_cu + 18: one full round + 18 tail tiles
    n_max = count + 300
    rng = np.random.default_rng(5)
    Nn = 3000
    X = rng.normal(size=(Nn + 1, d)).astype(np.float32)
    ids = rng.integers(0, Nn + 1, size=n_max).astype(np.int32)
    W = (rng.normal(size=(d, out)) * 0.1).astype(np.float32)
    b = (rng.normal(size=(out,)) * 0.1).astype(np.float32)
    Xd, Wd = Mat.from_numpy(X, dev, 32), Mat.from_numpy(W, dev)
    w3 = ops.split_rows(Wd)
    ids_d, cnt, bd = _i32(ids, dev), _i32(np.asarray([count]), dev), torch.from_numpy(b).to(dev)
    ws = torch.full((ops.split_tiled_ws_words(),), float("nan"), dtype=torch.float32, device=dev)
    res = {}
    for name in ("plain", "ws", "ws2"):
        o = Mat.zeros(n_max, out, dev)
        o.buf.fill_(-7.0)
        if name == "plain":
            ops.call("gs_dense_fwd_rows_split", Xd.ptr, Xd.ld, ops.ptr(ids_d), d, n_max, ops.ptr(cnt), ops.ptr(w3), out, ops.ACT_RELU,
                     ops.ptr(bd), o.ptr, o.ld, ops.current_stream())
        else:
            ops.call("gs_dense_fwd_rows_split_ws", Xd.ptr, Xd.ld, ops.ptr(ids_d), d, n_max, ops.ptr(cnt), ops.ptr(w3), out, ops.ACT_RELU,
                     ops.ptr(bd), o.ptr, o.ld, ops.ptr(ws), 4 * ws.numel(), ops.current_stream())
        torch.cuda.synchronize()
        res[name] = o.numpy()
    assert np.array_equal(res["ws"], res["ws2"])
    assert np.all(res["ws"][count:] == -7.0)
    full_rows = 128 * (n_cu // 2)                              # the first n_cu tiles = n_cu / 2 row tiles x 2 column tiles
    assert np.array_equal(res["ws"][:full_rows], res["plain"][:full_rows])
    want = np.maximum(X[ids[:count]].astype(np.float64) @ W.astype(np.float64) + b, 0)
    rms = np.sqrt((want ** 2).mean())
    assert np.abs(res["ws"][:count] - want).max() / rms < 4e-6
    assert np.abs(res["plain"][:count] - want).max() / rms < 4e-6


@pytest.mark.parametrize("n_max,count,d,out,act,bias,wild", [
    (6000, 5000, 602, 512, ops.ACT_RELU, True, False),      # the pooling MLP's shape (fewer rows), device-side row count
    (300, 300, 602, 512, ops.ACT_RELU, True, True),         # rows / columns / elements over a wide exponent range
    (1000, 777, 50, 512, ops.ACT_RELU, True, False),        # PPI's F = 50: one stage pair
    (700, 700, 40, 100, ops.ACT_IDENTITY, False, True),     # N = 100: a ragged column tile
    (260, 130, 256, 1024, ops.ACT_RELU, True, False)])      # model_size "big": hidden 1024
def test_dense_fwd_rows_split16(dev, n_max, count, d, out, act, bias, wild):
    """gs_dense_fwd_rows_split16 (two fp16 pieces per operand under row / column scales, three products) vs fp64, beside the fp32
    yardstick and the three-piece bf16 kernel: same accuracy class; rows beyond the count untouched; deterministic; with and
    without the split-K tail workspace."""
    rng = np.random.default_rng(n_max + d + out + 16)
    Nn = 5000
    X = rng.normal(size=(Nn + 1, d)).astype(np.float32); X[Nn] = 0
    W = (rng.normal(size=(d, out)) * 0.1).astype(np.float32)
    if wild:
        X *= np.ldexp(1.0, rng.integers(-40, 40, size=(Nn + 1, 1))).astype(np.float32)             # row magnitudes 2^-40 .. 2^40
        X[:, : d // 3] *= np.ldexp(1.0, rng.integers(-12, 0, size=(Nn + 1, d // 3))).astype(np.float32)   # small elements within a row
        W *= np.ldexp(1.0, rng.integers(-30, 30, size=(1, out))).astype(np.float32)                # column magnitudes
        X[7] = 0                                                                                   # an all-zero row
        W[:, 3] = 0                                                                                # an all-zero column
    ids = rng.integers(0, Nn + 1, size=n_max).astype(np.int32)
    ids[:4] = [7, Nn, 7, 0]
    b = (rng.normal(size=(out,)) * 0.1).astype(np.float32) if bias else None
    Xd, Wd = Mat.from_numpy(X, dev, 32), Mat.from_numpy(W, dev)
    X2, rexp = ops.split_table_f16(Xd)
    w2 = ops.split_rows_f16(Wd)
    w3 = ops.split_rows(Wd)
    ids_d = _i32(ids, dev)
    cnt = _i32(np.asarray([count]), dev)
    bd = torch.from_numpy(b).to(dev) if bias else None
    ws = torch.full((ops.split_tiled_ws_words(),), float("nan"), dtype=torch.float32, device=dev)
    res = {}
    for name in ("plain", "plain2", "ws", "ws2", "bf16x3"):
        o = Mat.zeros(n_max, out, dev)
        o.buf.fill_(-7.0)
        if name == "bf16x3":
            ops.call("gs_dense_fwd_rows_split", Xd.ptr, Xd.ld, ops.ptr(ids_d), d, n_max, ops.ptr(cnt), ops.ptr(w3), out, act,
                     ops.ptr(bd), o.ptr, o.ld, ops.current_stream())
        else:
            w = ws if name.startswith("ws") else None
            ops.call("gs_dense_fwd_rows_split16", ops.ptr(X2), ops.ptr(rexp), ops.ptr(ids_d), d, n_max, ops.ptr(cnt), ops.ptr(w2), out, act,
                     ops.ptr(bd), o.ptr, o.ld, ops.ptr(w), 4 * ws.numel() if w is not None else 0, ops.current_stream())
        torch.cuda.synchronize()
        res[name] = o.numpy()
    assert np.array_equal(res["plain"], res["plain2"]) and np.array_equal(res["ws"], res["ws2"])
    want = X[ids].astype(np.float64) @ W.astype(np.float64)
    if bias:
        want = want + b
    pre = want.copy()
    if act == ops.ACT_RELU:
        want = np.maximum(want, 0)
    ref32 = X[ids] @ W
    if bias:
        ref32 = ref32 + b
    if act == ops.ACT_RELU:
        ref32 = np.maximum(ref32, 0)
    # errors relative to the scale of each output's own dot product (|x| . |w|): row and column magnitudes differ by 2^+-70 here
    mag = np.abs(X[ids]).astype(np.float64) @ np.abs(W).astype(np.float64) + (np.abs(b) if bias else 0) + 1e-300
    errs = {}
    for name in ("plain", "ws", "bf16x3"):
        got = res[name].astype(np.float64)
        assert np.all(res[name][count:] == -7.0), name
        assert np.all(np.isfinite(res[name][:count])), name
        errs[name] = (np.abs(got[:count] - want[:count]) / mag[:count]).max()
    e32 = (np.abs(ref32[:count].astype(np.float64) - want[:count]) / mag[:count]).max()
    print("d=%d wild=%s: max |error| / (|x|.|w|) vs fp64: fp32 %.3g, three bf16 pieces %.3g, two fp16 pieces %.3g (with the tail split %.3g)"
          % (d, wild, e32, errs["bf16x3"], errs["plain"], errs["ws"]))
    for name in ("plain", "ws"):
        assert errs[name] <= max(2.5 * e32, 2.5 * errs["bf16x3"], 2e-7), (name, errs, e32)
        assert errs[name] < 1e-6


def test_split_pieces_f16_reconstruct_to_one_rounding(dev):
    """h + m of gs_split_table_f16 / gs_split_rows_f16 under their scales == the fp32 value to within ONE fp32 ulp (2^-23 relative:
    h keeps 11 bits, the residual has at most 12 and m keeps 11 of them) for elements within 2^-15 of the row's / column's
    largest, and to half a scaled fp16 subnormal ulp below that; exponents put the largest element at 2^13 .. 2^14; all-zero rows /
    columns get exponent 0."""
    rng = np.random.default_rng(3)
    rows, d = 200, 77
    X = rng.normal(size=(rows, d)).astype(np.float32) * np.ldexp(1.0, rng.integers(-30, 30, size=(rows, 1))).astype(np.float32)
    X[:, :20] *= np.ldexp(1.0, rng.integers(-30, 0, size=(rows, 20))).astype(np.float32)
    X[5] = 0
    Xd = Mat.from_numpy(X, dev, 32)
    X2, rexp = ops.split_table_f16(Xd)
    torch.cuda.synchronize()
    KP = X2.numel() * 2 // (rows * 2)
    pieces = X2.view(torch.float16).cpu().numpy().reshape(rows, 2, KP).astype(np.float64)
    e = rexp.cpu().numpy().astype(np.int64)
    assert KP % 64 == 0 and KP >= d
    assert np.all(pieces[:, :, d:] == 0)
    assert e[5] == 0
    mx = np.abs(X).max(axis=1).astype(np.float64)
    nz = mx > 0
    scaled = mx[nz] * np.exp2(e[nz].astype(np.float64))
    assert np.all((scaled >= 2.0 ** 13) & (scaled < 2.0 ** 14))
    # bit for bit the CPU restatement (oracle/split_pieces.py): same exponents, same round-to-nearest-even cuts, fp16 subnormals kept
    from oracle import split_pieces as sp
    h_o, m_o, e_o = sp.cut_rows_f16(X)
    assert np.array_equal(e, e_o)
    assert np.array_equal(X2.view(torch.float16).cpu().numpy().reshape(rows, 2, KP)[:, 0, :d].view(np.uint16), h_o.view(np.uint16))
    assert np.array_equal(X2.view(torch.float16).cpu().numpy().reshape(rows, 2, KP)[:, 1, :d].view(np.uint16), m_o.view(np.uint16))
    rec = (pieces[:, 0, :d] + pieces[:, 1, :d]) * np.exp2(-e.astype(np.float64))[:, None]
    err = np.abs(rec - X.astype(np.float64))
    tol = np.maximum(np.abs(X).astype(np.float64) * 2.0 ** -23, (mx * 2.0 ** -13 * 2.0 ** -25)[:, None])   # half the scaled fp16 subnormal ulp (2^-24 at a row maximum of 2^13..2^14)
    assert np.all(err <= tol * 1.0001)
    W = X.T.copy()                                              # columns of W = rows of X
    Wd = Mat.from_numpy(W, dev)
    w2 = ops.split_rows_f16(Wd)
    torch.cuda.synchronize()
    K, N = d, rows
    body = w2[: w2.numel() - N].view(torch.float16).cpu().numpy().reshape(KP // 8, 2, N, 8).astype(np.float64)
    ce = w2[w2.numel() - N:].cpu().numpy().astype(np.int64)
    assert np.array_equal(ce, e)
    recw = (body[:, 0] + body[:, 1]).transpose(0, 2, 1).reshape(KP, N)[:K] * np.exp2(-ce.astype(np.float64))[None, :]
    assert np.all(np.abs(recw - W.astype(np.float64)) <= tol.T * 1.0001)


def test_sampler_riding_in_the_tiled3_weight_gradient_launch_draws_what_the_standalone_launch_draws(dev):
    """gs_dense_wgrad_grouped_tiled3_sample under stress: the fan-out sampler (reference law on the materialised table, batch +
    label staging, one root per WAVE of a rider workgroup) riding behind the Reddit step's two layer-0 problems and a gather
    job, 400 launches with a moving cursor / clock while a second stream keeps the chip busy.  Every launch: ids and labels
    bit-equal to gs_sample_fanout_desc at the same cursor / clock, slabs bit-equal to the launch without a rider; a problem
    that gathers through the id buffer the sampler fills is refused."""
    import ctypes
    from graphsage_amd import _lib
    rng = np.random.default_rng(5)
    N, B, fans, C, M = 30000, 512, (10, 25), 41, 128
    deg = rng.integers(1, 200, size=N)
    rowptr_np = np.zeros(N + 1, np.int64)
    rowptr_np[1:] = np.cumsum(deg)
    col_np = rng.integers(0, N, size=int(rowptr_np[-1])).astype(np.int32)
    rowptr, col = torch.from_numpy(rowptr_np).to(dev), torch.from_numpy(col_np).to(dev)
    table = ops.build_padded_table(rowptr, col, N, N, M, 77)
    order = torch.from_numpy(rng.permutation(N).astype(np.int32)).to(dev)
    labels = Mat.from_numpy(rng.normal(size=(N + 1, C)).astype(np.float32), dev)
    offsets = [0, B, B + B * fans[0], B + B * fans[0] + B * fans[0] * fans[1]]
    cursor = torch.zeros(1, dtype=torch.int64, device=dev)
    clock = torch.zeros(1, dtype=torch.int64, device=dev)

    def desc(ids, lab):
        return ops.fanout_desc(rowptr, col, N, N, list(fans), offsets, ids, B, 123, step_dev=clock, order=order, cursor_dev=cursor,
                               label_table=labels, labels_out=lab, law=1,                     # GS_LAW_REFERENCE
                               max_degree=M, padded_table=table)

    ids_a, ids_b = (torch.full((offsets[-1],), -7, dtype=torch.int32, device=dev) for _ in range(2))
    lab_a, lab_b = Mat.zeros(B, C, dev), Mat.zeros(B, C, dev)
    qa, qb = desc(ids_a, lab_a), desc(ids_b, lab_b)
    # the Reddit step's layer-0 problems (gathered self term through a PRIVATE id vector, dense neighbor term) + a gather job
    n, d, out = 5632, 602, 128
    X = Mat.from_numpy(rng.normal(size=(N + 1, d)).astype(np.float32), dev, 32)
    means = Mat.from_numpy(rng.normal(size=(n, d)).astype(np.float32), dev, 32)
    dZ = Mat.from_numpy((rng.normal(size=(n, 2 * out)) * 0.1).astype(np.float32), dev, 32)
    aidx = torch.from_numpy(rng.integers(0, N, size=n).astype(np.int32)).to(dev)
    gidx = torch.from_numpy(rng.integers(0, N, size=2048 * 25).astype(np.int32)).to(dev)
    gout = Mat.zeros(2048, d, dev, 32)
    jobs = (_lib.GatherDesc * 1)(ops.gather_job(X, gidx, 2048, 25, gout))

    def problems():
        q0, s0, _ = _wgrad_desc(X, aidx, dZ, 0, out, n, d, 11, dev)
        q1, s1, _ = _wgrad_desc(means, None, dZ, out, out, n, d, 11, dev)
        return (_lib.WgradDesc * 2)(q0, q1), (s0, s1)

    arr_ref, slabs_ref = problems()
    ops.call("gs_dense_wgrad_grouped_tiled3", ctypes.addressof(arr_ref), 2, ctypes.addressof(jobs), 1, ops.current_stream())
    torch.cuda.synchronize()
    want_slabs = [s.clone() for s in slabs_ref]
    side = torch.cuda.Stream()
    arr, slabs = problems()
    for it in range(400):
        cursor.fill_(int(rng.integers(0, N)))
        clock.fill_(it)
        for s in slabs:
            s.fill_(float("nan"))
        ids_b.fill_(-7)
        torch.cuda.synchronize()
        if it % 3 == 0:
            with torch.cuda.stream(side):                                    # concurrent work on another stream
                ops.gather_mean_fwd(X, gidx, 2048, 25, out=Mat.zeros(2048, d, dev, 32), stream=side.cuda_stream)
        ops.sample_fanout_desc(qa)
        ops.call("gs_dense_wgrad_grouped_tiled3_sample", ctypes.addressof(arr), 2, ctypes.addressof(jobs), 1, ctypes.addressof(qb),
                 ops.current_stream())
        torch.cuda.synchronize()
        assert torch.equal(ids_a, ids_b), "launch %d: ids differ at %d slots" % (it, int((ids_a != ids_b).sum()))
        assert torch.equal(lab_a.buf, lab_b.buf), "launch %d: labels differ" % it
        for got, want in zip(slabs, want_slabs):
            assert torch.equal(got, want), "launch %d: slabs differ" % it
    # a gathered problem whose row ids ARE the sampler's id buffer is refused
    qbad, _, _ = _wgrad_desc(X, ids_b[:n], dZ, 0, out, n, d, 11, dev)
    bad = (_lib.WgradDesc * 1)(qbad)
    with pytest.raises(_lib.GraphsageAmdError):
        ops.call("gs_dense_wgrad_grouped_tiled3_sample", ctypes.addressof(bad), 1, ctypes.addressof(jobs), 1, ctypes.addressof(qb),
                 ops.current_stream())
