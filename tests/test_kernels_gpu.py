"""-m gpu parity tests: every HIP kernel (called through the C ABI) vs the NumPy oracle.

Tolerances: bit-exact for int32 index outputs; rtol/atol 1e-4 for fp32 (north_star tolerance;
TF/Eigen summation order is unspecified, ours is fixed -- see DESIGN.md).
"""
import numpy as np
import pytest
import torch

from graphsage_amd import ops
from graphsage_amd.ops import Mat
from oracle import graphsage_oracle as orc
from oracle import sampler_hash

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)


def _i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


def _sync():
    torch.cuda.synchronize()


def assert_close_rownorm(got, want, tol=1e-4, err_msg=""):
    """|got - want| <= tol * |want| + tol * rms(want row): 1e-4 (north_star) RELATIVE to the magnitude of the output row,
    instead of an absolute 1e-4 * sqrt(K) that presumes unit-variance operands (a contraction's rounding error scales with
    the magnitude of its row, and an entry that cancels to ~0 cannot be held to 1e-4 of itself)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    rms = np.sqrt((want * want).mean(axis=-1, keepdims=True))
    bad = np.abs(got - want) > tol * np.abs(want) + tol * rms
    assert not bad.any(), "%s: %d / %d elements outside %g of their row's rms (max abs err %.3g, row rms %.3g)" % (
        err_msg, int(bad.sum()), bad.size, tol, float(np.abs(got - want).max()), float(rms.mean()))


# ----------------------------------------------------------------------------- K1
@pytest.mark.parametrize("n,s,max_deg", [(512, 10, 128), (5120, 25, 128), (7, 3, 5), (1, 128, 128), (100, 1, 4)])
def test_sample_padded_bit_exact(dev, n, s, max_deg):
    rng = np.random.default_rng(n * 131 + s)
    N = 1000
    adj = rng.integers(0, N + 1, size=(N + 1, max_deg)).astype(np.int32)
    adj[N, :] = N
    ids = rng.integers(0, N + 1, size=n).astype(np.int32)
    perm = rng.permutation(max_deg).astype(np.int32)
    want = orc.uniform_neighbor_sampler(adj, ids, s, perm)
    got = ops.sample_padded(_i32(adj, dev), _i32(ids, dev), _i32(perm, dev), s)
    _sync()
    assert np.array_equal(got.cpu().numpy().reshape(n, s), want)


def _rand_csr(rng, N, max_deg, frac_zero=0.2):
    deg = rng.integers(1, max_deg + 1, size=N)
    deg[rng.random(N) < frac_zero] = 0
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, N, size=int(rowptr[-1])).astype(np.int32)
    return rowptr, col


@pytest.mark.parametrize("n,s", [(512, 10), (5120, 25), (3, 1), (64, 1), (65, 1), (1000, 64), (37, 100), (13, 7)])
def test_sample_csr_bit_exact(dev, n, s):
    rng = np.random.default_rng(n * 7 + s)
    N = 5000
    rowptr, col = _rand_csr(rng, N, 300)
    ids = rng.integers(0, N + 1, size=n).astype(np.int32)  # includes the pad id N
    for hop, step, off in [(0, 0, 0), (1, 17, 12345)]:
        want = sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids, s, 123, step, hop, off)
        got = ops.sample_uniform_csr(torch.from_numpy(rowptr).to(dev), _i32(col, dev), N, N, _i32(ids, dev), s, 123,
                                     step=step, hop=hop, global_row_offset=off)
        _sync()
        assert np.array_equal(got.cpu().numpy().reshape(n, s), want)


def test_sample_csr_step_dev_and_sharding(dev):
    """step read from device memory == step by value; a row's draw is independent of sharding."""
    rng = np.random.default_rng(5)
    N = 3000
    rowptr, col = _rand_csr(rng, N, 50)
    ids = rng.integers(0, N, size=512).astype(np.int32)
    rp, cl, idd = torch.from_numpy(rowptr).to(dev), _i32(col, dev), _i32(ids, dev)
    full = ops.sample_uniform_csr(rp, cl, N, N, idd, 10, 99, step=7, hop=1).cpu().numpy()
    step_dev = torch.tensor([7], dtype=torch.int64, device=dev)
    via_dev = ops.sample_uniform_csr(rp, cl, N, N, idd, 10, 99, step=0, step_dev=step_dev, hop=1).cpu().numpy()
    assert np.array_equal(full, via_dev)
    half = ops.sample_uniform_csr(rp, cl, N, N, idd[256:].contiguous(), 10, 99, step=7, hop=1,
                                  global_row_offset=256).cpu().numpy()
    assert np.array_equal(full[2560:], half)


def test_sample_csr_properties(dev):
    """every pick is a true neighbor, degree-0 rows give the pad id, marginal is ~uniform (chi^2)."""
    rng = np.random.default_rng(11)
    N = 200
    rowptr, col = _rand_csr(rng, N, 20)
    # make adjacency lists duplicate-free so the uniformity test is clean
    for i in range(N):
        d = rowptr[i + 1] - rowptr[i]
        col[rowptr[i]:rowptr[i + 1]] = rng.choice(N, size=d, replace=False)
    node = int(np.argmax(np.diff(rowptr)))
    d = int(rowptr[node + 1] - rowptr[node])
    n = 20000
    ids = np.full(n, node, dtype=np.int32)
    got = ops.sample_uniform_csr(torch.from_numpy(rowptr).to(dev), _i32(col, dev), N, N, _i32(ids, dev), 8, 2024)
    got = got.cpu().numpy()
    neigh = col[rowptr[node]:rowptr[node + 1]]
    assert np.isin(got, neigh).all()
    counts = np.array([(got == v).sum() for v in neigh], dtype=np.float64)
    exp = got.size / d
    chi2 = ((counts - exp) ** 2 / exp).sum()
    assert chi2 < d + 6 * np.sqrt(2 * d), (chi2, d)  # mean d-1, sd sqrt(2(d-1)); 6 sigma
    zero = int(np.where(np.diff(rowptr) == 0)[0][0])
    got0 = ops.sample_uniform_csr(torch.from_numpy(rowptr).to(dev), _i32(col, dev), N, N,
                                  _i32(np.array([zero, N]), dev), 5, 1).cpu().numpy()
    assert (got0 == N).all()


def test_select_batch_and_counter(dev):
    order = _i32(np.arange(100)[::-1].copy(), dev)
    cur = torch.tensor([90], dtype=torch.int64, device=dev)
    out = torch.empty(20, dtype=torch.int32, device=dev)
    ops.select_batch(order, cur, 20, out)
    ops.advance_counter(cur, 20)
    _sync()
    want = np.arange(100)[::-1][(90 + np.arange(20)) % 100]
    assert np.array_equal(out.cpu().numpy(), want)
    assert int(cur.item()) == 110


# ----------------------------------------------------------------------------- K2
@pytest.mark.parametrize("n,s,d,ldm", [(5120, 25, 602, 32), (512, 10, 602, 32), (37, 3, 50, 4), (9, 1, 7, 4),
                                       (64, 70, 130, 4), (5, 8, 256, 4), (3, 130, 1024, 4)])
def test_gather_mean_fwd(dev, n, s, d, ldm):
    rng = np.random.default_rng(n + s + d)
    N = 3000
    X = rng.normal(size=(N + 1, d)).astype(np.float32)
    X[N] = 0
    idx = rng.integers(0, N + 1, size=n * s).astype(np.int32)
    Xd = Mat.from_numpy(X, dev, ld_multiple=ldm)
    out = ops.gather_mean_fwd(Xd, _i32(idx, dev), n, s)
    _sync()
    want = X[idx].reshape(n, s, d).mean(axis=1, dtype=np.float32)
    np.testing.assert_allclose(out.numpy(), want, **TOL)
    assert (out.buf[:, d:].cpu().numpy() == 0).all()
    # GCN mean: (sum neigh + self) / (s + 1), self rows gathered by index
    sidx = rng.integers(0, N, size=n).astype(np.int32)
    out2 = ops.gather_mean_fwd(Xd, _i32(idx, dev), n, s, self_src=Xd, self_idx=_i32(sidx, dev))
    _sync()
    want2 = (X[idx].reshape(n, s, d).sum(axis=1, dtype=np.float32) + X[sidx]) / np.float32(s + 1)
    np.testing.assert_allclose(out2.numpy(), want2, **TOL)


def test_gather_mean_contiguous_and_rows(dev):
    rng = np.random.default_rng(3)
    H = rng.normal(size=(5120, 256)).astype(np.float32)
    Hd = Mat.from_numpy(H, dev)
    out = ops.gather_mean_fwd(Hd, None, 512, 10)
    _sync()
    np.testing.assert_allclose(out.numpy(), H.reshape(512, 10, 256).mean(axis=1), **TOL)
    ids = rng.integers(0, 5120, size=777).astype(np.int32)
    rows = ops.gather_rows(Hd, _i32(ids, dev))
    _sync()
    assert np.array_equal(rows.numpy(), H[ids])  # pure copy: bit exact


def test_gather_mean_empty(dev):
    Xd = Mat.zeros(10, 8, dev)
    out = Mat.zeros(0, 8, dev)
    ops.gather_mean_fwd(Xd, torch.empty(0, dtype=torch.int32, device=dev), 0, 5, out=out)
    _sync()


def test_mean_bwd(dev):
    rng = np.random.default_rng(4)
    n, s, d = 512, 10, 256
    dm = rng.normal(size=(n, d)).astype(np.float32)
    y = rng.normal(size=(n * s, d)).astype(np.float32)
    dn = Mat.zeros(n * s, d, dev)
    ops.mean_bwd(Mat.from_numpy(dm, dev), n, s, 1.0 / s, dn, mask_y=Mat.from_numpy(y, dev))
    _sync()
    want = np.repeat(dm / np.float32(s), s, axis=0) * (y > 0)
    np.testing.assert_allclose(dn.numpy(), want, **TOL)
    ops.mean_bwd(Mat.from_numpy(dm, dev), n, s, 1.0 / s, dn, accumulate=True)
    _sync()
    np.testing.assert_allclose(dn.numpy(), want + np.repeat(dm / np.float32(s), s, axis=0), **TOL)


# ----------------------------------------------------------------------------- K3
def _asym(rng, shape):
    return rng.normal(size=shape).astype(np.float32)


@pytest.mark.parametrize("tA,tB", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (100, 41, 602), (512, 256, 602), (33, 7, 5), (1300, 130, 70)])
def test_gemm_all_layouts(dev, tA, tB, M, N, K):
    rng = np.random.default_rng(M + N + K + tA * 2 + tB)
    A = _asym(rng, (M, K))
    B = _asym(rng, (K, N))
    bias = _asym(rng, (N,))
    Ad = Mat.from_numpy(A.T.copy() if tA else A, dev)
    Bd = Mat.from_numpy(B.T.copy() if tB else B, dev)
    C = Mat.zeros(M, N, dev)
    C.buf.fill_(7.0)
    ops.gemm(tA, tB, M, N, K, Ad, Bd, C, bias=torch.from_numpy(bias).to(dev), act=ops.ACT_RELU)
    _sync()
    want = np.maximum(A.astype(np.float64) @ B.astype(np.float64) + bias, 0)
    assert_close_rownorm(C.numpy(), want)
    assert (C.buf[:, N:ops.round_up(N, 4)].cpu().numpy() == 0).all()


def test_gemm_big_tile_and_gather(dev):
    """128x128 tile path (>= 1024 tiles) with gathered A rows: the MaxPool MLP shape in miniature."""
    rng = np.random.default_rng(8)
    Nn, d, hid, rows = 4000, 602, 512, 128 * 260
    X = _asym(rng, (Nn, d))
    W = _asym(rng, (d, hid)) * 0.05
    b = _asym(rng, (hid,))
    idx = rng.integers(0, Nn, size=rows).astype(np.int32)
    C = Mat.zeros(rows, hid, dev)
    ops.gemm(0, 0, rows, hid, d, Mat.from_numpy(X, dev, 32), Mat.from_numpy(W, dev), C, a_row_idx=_i32(idx, dev),
             bias=torch.from_numpy(b).to(dev), act=ops.ACT_RELU)
    _sync()
    sel = rng.choice(rows, size=512, replace=False)
    want = np.maximum(X[idx[sel]].astype(np.float64) @ W.astype(np.float64) + b, 0)
    np.testing.assert_allclose(C.numpy()[sel], want, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("concat", [True, False])
@pytest.mark.parametrize("n,d,out", [(5120, 602, 128), (512, 256, 128), (77, 50, 12)])
def test_sage_dense_fwd(dev, concat, n, d, out):
    rng = np.random.default_rng(n + d)
    Nn = 3000
    X = _asym(rng, (Nn, d))
    mean = _asym(rng, (n, d))
    Ws, Wn = _asym(rng, (d, out)) * 0.1, _asym(rng, (d, out)) * 0.1
    sidx = rng.integers(0, Nn, size=n).astype(np.int32)
    o = Mat.zeros(n, out * (2 if concat else 1), dev)
    ops.sage_dense_fwd(Mat.from_numpy(X, dev, 32), _i32(sidx, dev), Mat.from_numpy(mean, dev), None, n,
                       Mat.from_numpy(Ws, dev), Mat.from_numpy(Wn, dev), out, concat, ops.ACT_RELU, None, o)
    _sync()
    want, _ = orc.mean_aggregator_fwd(X[sidx].astype(np.float64), mean[:, None, :].astype(np.float64),
                                      Ws.astype(np.float64), Wn.astype(np.float64), concat, "relu")
    np.testing.assert_allclose(o.numpy(), want, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("n,d,out,col0,slabs", [(5120, 602, 128, 128, 16), (512, 602, 128, 0, 2), (100, 50, 41, 0, 3),
                                                 (31, 256, 128, 0, 1)])
def test_dense_wgrad_slabs(dev, n, d, out, col0, slabs):
    rng = np.random.default_rng(n + d + out)
    Nn = 2000
    X = _asym(rng, (Nn, d))
    idx = rng.integers(0, Nn, size=n).astype(np.int32)
    dZ = _asym(rng, (n, col0 + out))
    ld_slab = ops.round_up(out, 4)
    sl = torch.full((slabs * d * ld_slab,), 3.0, dtype=torch.float32, device=dev)
    ops.dense_wgrad(Mat.from_numpy(X, dev, 32), _i32(idx, dev), Mat.from_numpy(dZ, dev), col0, out, n, slabs, sl, ld_slab)
    w = np.ones((d, out), np.float32)
    grad = torch.zeros((d, ld_slab), dtype=torch.float32, device=dev)
    ops.reduce_slabs(sl, slabs, d * ld_slab, d, out, ld_slab, 0.5, Mat.from_numpy(w, dev).ptr, ld_slab, ops.ptr(grad), ld_slab)
    _sync()
    want = X[idx].astype(np.float64).T @ dZ[:, col0:].astype(np.float64) + 0.5
    assert_close_rownorm(grad.cpu().numpy()[:, :out], want)
    assert (grad.cpu().numpy()[:, out:] == 0).all()


def test_dense_dgrad(dev):
    rng = np.random.default_rng(21)
    n, d, out, col0 = 512, 256, 128, 128
    dZ = _asym(rng, (n, col0 + out))
    W = _asym(rng, (d, out))
    dX = Mat.from_numpy(np.ones((n, d), np.float32), dev)
    ops.dense_dgrad(Mat.from_numpy(dZ, dev), col0, out, n, Mat.from_numpy(W, dev), dX, accumulate=True)
    _sync()
    want = 1.0 + dZ[:, col0:].astype(np.float64) @ W.astype(np.float64).T
    np.testing.assert_allclose(dX.numpy(), want, rtol=1e-4, atol=1e-3)
    # head-shaped: [512, 41] x [256, 41]^T
    dZ2, W2 = _asym(rng, (n, 41)), _asym(rng, (256, 41))
    dX2 = Mat.zeros(n, 256, dev)
    ops.dense_dgrad(Mat.from_numpy(dZ2, dev), 0, 41, n, Mat.from_numpy(W2, dev), dX2)
    _sync()
    np.testing.assert_allclose(dX2.numpy(), dZ2.astype(np.float64) @ W2.astype(np.float64).T, rtol=1e-4, atol=1e-3)


def test_act_bwd_and_colsum(dev):
    rng = np.random.default_rng(22)
    n, c = 777, 41
    dY, Y = _asym(rng, (n, c)), _asym(rng, (n, c))
    dZ = Mat.zeros(n, c, dev)
    ops.act_bwd(Mat.from_numpy(dY, dev), Mat.from_numpy(Y, dev), n, c, ops.ACT_RELU, dZ)
    sl = torch.zeros((5 * 44,), dtype=torch.float32, device=dev)
    ops.colsum_slabs(dZ, n, c, 5, sl, 44)
    _sync()
    want = dY * (Y > 0)
    np.testing.assert_allclose(dZ.numpy(), want, **TOL)
    np.testing.assert_allclose(sl.cpu().numpy().reshape(5, 44)[:, :c].sum(0), want.sum(0), rtol=1e-4, atol=1e-3)


# ----------------------------------------------------------------------------- K4
def test_segment_max(dev):
    rng = np.random.default_rng(23)
    n, s, hid = 300, 25, 512
    H = np.maximum(_asym(rng, (n * s, hid)), 0)
    pooled, arg = Mat.zeros(n, hid, dev), torch.zeros((n, hid), dtype=torch.int32, device=dev)
    ops.segment_max_fwd(Mat.from_numpy(H, dev), n, s, pooled, arg)
    _sync()
    H3 = H.reshape(n, s, hid)
    assert np.array_equal(pooled.numpy(), H3.max(axis=1))
    assert np.array_equal(arg.cpu().numpy(), H3.argmax(axis=1))
    dP = _asym(rng, (n, hid))
    dH = Mat.zeros(n * s, hid, dev)
    ops.segment_max_bwd(Mat.from_numpy(dP, dev), pooled, arg, n, s, dH)
    _sync()
    want = np.zeros_like(H3)
    np.put_along_axis(want, H3.argmax(axis=1)[:, None, :], (dP * (H3.max(axis=1) > 0))[:, None, :], axis=1)
    assert np.array_equal(dH.numpy(), want.reshape(n * s, hid))


@pytest.mark.parametrize("n,s,d,hid,gathered", [(5120, 25, 602, 512, True), (512, 10, 602, 512, True), (203, 25, 50, 130, False),
                                                 (37, 7, 33, 64, True), (3, 64, 20, 40, False), (1, 1, 9, 12, True)])
def test_dense_pool_max_fwd(dev, n, s, d, hid, gathered):
    """gs_dense_pool_max_fwd == gs_sage_dense_fwd(relu) + gs_segment_max_fwd (pooled values within GEMM tolerance, the
    arg-max equal wherever the top two activations of a group are not a near tie), 128- and 64-row tile forms, groups that
    do not divide the tile, ragged columns, last partial tile."""
    rng = np.random.default_rng(n + s)
    Nn = 3000
    X = _asym(rng, (Nn + 1, d)); X[Nn] = 0
    idx = rng.integers(0, Nn + 1, size=n * s).astype(np.int32)
    W, b = _asym(rng, (d, hid)) * 0.2, _asym(rng, (hid,)) * 0.1
    Xd = Mat.from_numpy(X, dev, 32) if gathered else Mat.from_numpy(X[idx], dev)
    idx_d = _i32(idx, dev) if gathered else None
    Wd, bd = Mat.from_numpy(W, dev), torch.from_numpy(b).to(dev)
    pooled, arg = Mat.zeros(n, hid, dev), torch.full((n, hid), -1, dtype=torch.int32, device=dev)
    ops.dense_pool_max_fwd(Xd, idx_d, n, s, Wd, bd, pooled, arg)
    _sync()
    H = np.maximum(X[idx].astype(np.float64) @ W + b, 0).reshape(n, s, hid)
    assert_close_rownorm(pooled.numpy(), H.max(axis=1))
    got = arg.cpu().numpy()
    assert got.min() >= 0 and got.max() < s
    # the device's choice attains the maximum (up to fp32 summation noise) ...
    picked = np.take_along_axis(H, got[:, None, :].astype(np.int64), axis=1)[:, 0, :]
    assert_close_rownorm(picked, H.max(axis=1))
    # ... and is the FIRST such row wherever the runner-up is clearly smaller (or everything is clamped to 0 -> row 0)
    srt = np.sort(H, axis=1)
    clear = (srt[:, -1, :] - (srt[:, -2, :] if s > 1 else -1.0)) > 1e-3
    assert np.array_equal(got[clear], H.argmax(axis=1)[clear])
    allzero = H.max(axis=1) == 0
    assert not got[allzero].any()


@pytest.mark.parametrize("m,nv", [(133120, 232966), (5000, 3001), (2049, 70000), (300000, 1000)])
def test_unique_ids(dev, m, nv):
    """gs_unique_ids (flag array + prefix sum: distinct ids ascending, inverse map, count as a device word) vs np.unique --
    heavy duplication, sparse occupancy, the pad id, more ids than values."""
    rng = np.random.default_rng(m + nv)
    ids = rng.integers(0, nv, size=m).astype(np.int32)
    ids[:7] = [nv - 1, 0, nv - 1, 5 % nv, 5 % nv, 0, nv - 1]
    ids_d = _i32(ids, dev)
    rank = torch.zeros(2 * nv, dtype=torch.int32, device=dev)          # [flags | ranks]: the flag half zero on first use, self-cleaning
    rank[nv:] = -5
    sums = torch.zeros(256, dtype=torch.int32, device=dev)
    uniq = torch.full((m,), -1, dtype=torch.int32, device=dev)
    inv = torch.full((m,), -1, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    for _ in range(2):                                    # reusable without host-side resets
        ops.call("gs_unique_ids", ops.ptr(ids_d), m, nv, ops.ptr(rank), ops.ptr(sums), ops.ptr(uniq), ops.ptr(inv), ops.ptr(cnt), None)
        _sync()
        want_u, want_inv = np.unique(ids, return_inverse=True)
        U = int(cnt.item())
        assert U == len(want_u)
        assert np.array_equal(uniq.cpu().numpy()[:U], want_u)
        assert np.array_equal(inv.cpu().numpy(), want_inv.astype(np.int32))
        assert int(rank[:nv].abs().sum().item()) == 0                  # the flag half is left zero


@pytest.mark.parametrize("groups,d,hid", [([(5120, 25), (512, 10)], 602, 512), ([(700, 10)], 50, 130), ([(90, 64), (30, 3)], 33, 64)])
def test_pool_max_on_unique_ids_equals_fused_launch(dev, groups, d, hid):
    """The de-duplicated pooling path (gs_unique_ids -> gs_dense_fwd_rows_dev on the distinct ids -> gs_segment_max_gather_fwd)
    gives the pooled values and arg-max rows of gs_dense_pool_max_fwd on the expanded rows, bit for bit."""
    rng = np.random.default_rng(d + hid)
    Nn = 4000
    X = _asym(rng, (Nn + 1, d)); X[Nn] = 0
    m = sum(n * s for n, s in groups)
    idx = rng.integers(0, Nn + 1, size=m).astype(np.int32)
    idx[rng.random(m) < 0.3] = idx[0]                                     # heavy duplication
    W, b = _asym(rng, (d, hid)) * 0.2, _asym(rng, (hid,)) * 0.1
    Xd, idx_d = Mat.from_numpy(X, dev, 32), _i32(idx, dev)
    Wd, bd = Mat.from_numpy(W, dev), torch.from_numpy(b).to(dev)
    n_total = sum(n for n, _ in groups)
    p1, a1 = Mat.zeros(n_total, hid, dev), torch.full((n_total, hid), -1, dtype=torch.int32, device=dev)
    p2, a2 = Mat.zeros(n_total, hid, dev), torch.full((n_total, hid), -1, dtype=torch.int32, device=dev)
    r = hr = 0
    for n, s in groups:
        ops.dense_pool_max_fwd(Xd, idx_d[hr:hr + n * s], n, s, Wd, bd, p1.rows_slice(r, r + n), a1[r:r + n])
        r, hr = r + n, hr + n * s
    nv = Nn + 1
    rank, sums = torch.zeros(2 * nv, dtype=torch.int32, device=dev), torch.zeros(256, dtype=torch.int32, device=dev)
    uniq, inv = torch.zeros(m, dtype=torch.int32, device=dev), torch.zeros(m, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.call("gs_unique_ids", ops.ptr(idx_d), m, nv, ops.ptr(rank), ops.ptr(sums), ops.ptr(uniq), ops.ptr(inv), ops.ptr(cnt), None)
    Hu = Mat.zeros(max(m, 2049), hid, dev)
    Hu.buf.fill_(float("nan"))
    ops.call("gs_dense_fwd_rows_dev", Xd.ptr, Xd.ld, ops.ptr(uniq), d, max(m, 2049), ops.ptr(cnt), Wd.ptr, Wd.ld, hid, ops.ACT_RELU,
             ops.ptr(bd), Hu.ptr, Hu.ld, None)
    r = hr = 0
    for n, s in groups:
        pr = p2.rows_slice(r, r + n)
        ops.call("gs_segment_max_gather_fwd", Hu.ptr, Hu.ld, inv.data_ptr() + 4 * hr, n, s, hid, pr.ptr, pr.ld,
                 a2[r:r + n].data_ptr(), a2.stride(0), None)
        r, hr = r + n, hr + n * s
    _sync()
    U = int(cnt.item())
    assert U == len(np.unique(idx)) and np.isfinite(Hu.numpy()[:U]).all()
    assert np.array_equal(p1.numpy(), p2.numpy())
    assert np.array_equal(a1.cpu().numpy(), a2.cpu().numpy())


@pytest.mark.parametrize("n,s,d,hid,k", [(203, 25, 602, 512, 7), (37, 10, 50, 128, 5), (9, 3, 70, 100, 2),
                                          (64, 16, 33, 64, 3), (21, 25, 40, 1100, 2),
                                          # the LDS-DMA pipeline (hidden >= 512, s <= 32): 1, 2, 3, 4 row segments per wave, a slice
                                          # longer than the ids staged in LDS at a time (300 groups x 25 > 3072 ids: three chunks),
                                          # fewer groups than pipeline stages, a ragged last slice, s = 32
                                          (50, 3, 602, 512, 3), (512, 10, 602, 512, 16), (77, 20, 130, 640, 4), (600, 25, 100, 512, 2),
                                          (2, 25, 64, 512, 1), (3, 8, 65, 512, 2), (45, 32, 70, 512, 4)])
def test_maxpool_sparse_wgrad(dev, n, s, d, hid, k):
    """dW = X[ids]^T . dH with dH one-hot per (group, column) == the dense product, without ever forming dH."""
    rng = np.random.default_rng(29)
    N = 500
    X = _asym(rng, (N, d))
    ids = rng.integers(0, N, size=n * s).astype(np.int32)
    arg = rng.integers(0, s, size=(n, hid)).astype(np.int32)
    dpm = _asym(rng, (n, hid)) * (rng.random((n, hid)) > 0.4)
    ld = (hid + 3) // 4 * 4
    slabs = torch.full((k, d, ld), float("nan"), device=dev)
    ops.maxpool_sparse_wgrad(Mat.from_numpy(X, dev), torch.from_numpy(ids).to(dev), n, s, torch.from_numpy(arg).to(dev),
                             Mat.from_numpy(dpm.astype(np.float32), dev), hid, k, slabs.data_ptr(), ld)
    _sync()
    got = slabs.cpu().numpy()[:, :, :hid].astype(np.float64).sum(axis=0)
    dH = np.zeros((n, s, hid))
    np.put_along_axis(dH, arg[:, None, :].astype(np.int64), dpm[:, None, :].astype(np.float64), axis=1)
    want = X[ids].astype(np.float64).T @ dH.reshape(n * s, hid)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)


def test_device_hash_known_answers(dev):
    """The device sampler and dropout kernels against tests/golden/hash_kat.npz (computed with Python big ints)."""
    import os
    k = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hash_kat.npz"))
    ns, seed, step, hop, row_off, pad = [int(v) for v in k["csr_args"]]
    out = ops.sample_uniform_csr(torch.from_numpy(k["rowptr"]).to(dev), torch.from_numpy(k["col"]).to(dev), 4, pad,
                                 torch.from_numpy(k["ids"]).to(dev), ns, seed, step=step, hop=hop, global_row_offset=row_off)
    _sync()
    assert np.array_equal(out.cpu().numpy().reshape(-1, ns), k["picked"])
    dseed, clock, site, row0, n_rows, d = [int(v) for v in k["drop_args"]]
    rate = float(k["drop_rate"])
    clk = torch.tensor([clock], dtype=torch.int64, device=dev)
    ones = Mat.from_numpy(np.ones((n_rows, d), np.float32), dev)
    ops.dropout_rows(ones, None, n_rows, ops.dropout_desc(dseed, clk, site, rate, row0), ones)
    _sync()
    assert np.array_equal(ones.numpy() > 0, k["keep"] == 1)
    assert np.allclose(ones.numpy()[k["keep"] == 1], 1.0 / (1.0 - rate))


# ----------------------------------------------------------------------------- dropout
@pytest.mark.parametrize("rate", [0.1, 0.5, 0.93])
def test_dropout_rows_matches_hash(dev, rate):
    """gs_dropout_rows == X[ids] * mask / keep_prob with the counter-hash mask of oracle/sampler_hash.py, bit exact;
    masks depend on (clock, site, row0); the kept fraction is 1 - rate."""
    rng = np.random.default_rng(31)
    N, n, d = 300, 257, 50
    X = _asym(rng, (N, d))
    ids = rng.integers(0, N, size=n).astype(np.int32)
    clock = torch.tensor([5], dtype=torch.int64, device=dev)
    Xd, ids_d = Mat.from_numpy(X, dev), torch.from_numpy(ids).to(dev)
    outs = {}
    for (site, row0, clk) in [(19, 1000, 5), (20, 1000, 5), (19, 1001, 5), (19, 1000, 6)]:
        clock.fill_(clk)
        out = Mat.zeros(n, d, dev)
        ops.dropout_rows(Xd, ids_d, n, ops.dropout_desc(77, clock, site, rate, row0), out)
        _sync()
        m = sampler_hash.dropout_mask(77, clk, site, row0, n, d, rate)
        assert np.array_equal(out.numpy(), X[ids] * m)
        outs[(site, row0, clk)] = m
    base = outs[(19, 1000, 5)]
    assert abs((base > 0).mean() - (1 - rate)) < 0.02
    assert not np.array_equal(base, outs[(20, 1000, 5)]) and not np.array_equal(base, outs[(19, 1000, 6)])
    assert np.array_equal(base[1:], outs[(19, 1001, 5)][:-1])          # row0 shifts the global row index
    # in place, no ids: the backward use (same mask applied to a gradient)
    g = Mat.from_numpy(X[:n], dev)
    clock.fill_(5)
    ops.dropout_rows(g, None, n, ops.dropout_desc(77, clock, 19, rate, 1000), g)
    _sync()
    assert np.array_equal(g.numpy(), X[:n] * base)


@pytest.mark.parametrize("n,s,d,gcn", [(65, 25, 602, False), (33, 10, 50, True), (7, 3, 8, False), (5, 70, 36, False)])
def test_gather_mean_dropout(dev, n, s, d, gcn):
    """K2 with dropout of every gathered row before the mean (aggregators.py:46-48)."""
    rng = np.random.default_rng(32)
    N, rate = 500, 0.4
    X = _asym(rng, (N, d))
    idx = rng.integers(0, N, size=n * s).astype(np.int32)
    clock = torch.tensor([3], dtype=torch.int64, device=dev)
    selfm = _asym(rng, (n, d)) if gcn else None
    out = Mat.zeros(n, d, dev)
    ops.gather_mean_fwd(Mat.from_numpy(X, dev), torch.from_numpy(idx).to(dev), n, s, out=out,
                        self_src=Mat.from_numpy(selfm, dev) if gcn else None,
                        drop=ops.dropout_desc(9, clock, 33, rate, 4096))
    _sync()
    m = sampler_hash.dropout_mask(9, 3, 33, 4096, n * s, d, rate)
    rows = (X[idx] * m).reshape(n, s, d).astype(np.float64)
    want = (rows.sum(axis=1) + selfm) / (s + 1) if gcn else rows.mean(axis=1)
    np.testing.assert_allclose(out.numpy(), want, rtol=1e-5, atol=1e-6)


# ----------------------------------------------------------------------------- K5
def test_l2norm_fwd_bwd(dev):
    rng = np.random.default_rng(24)
    n, d = 512, 256
    x = _asym(rng, (n, d))
    x[3] = 0  # clamped row (sum sq < 1e-12)
    dy = _asym(rng, (n, d))
    y, inv, dx = Mat.zeros(n, d, dev), torch.zeros(n, device=dev), Mat.zeros(n, d, dev)
    ops.l2norm_fwd(Mat.from_numpy(x, dev), n, y, inv)
    ops.l2norm_bwd(Mat.from_numpy(dy, dev), y, inv, n, dx)
    _sync()
    wy, cache = orc.l2_normalize_fwd(x.astype(np.float64))
    np.testing.assert_allclose(y.numpy(), wy, **TOL)
    np.testing.assert_allclose(dx.numpy()[4:], orc.l2_normalize_bwd(dy.astype(np.float64), cache)[4:], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("sig,C", [(False, 41), (True, 121), (False, 3), (True, 64)])
def test_class_loss(dev, sig, C):
    rng = np.random.default_rng(C)
    n = 300
    x = _asym(rng, (n, C)) * 3
    z = (rng.random((n, C)) > 0.5).astype(np.float32) if sig else np.eye(C, dtype=np.float32)[rng.integers(0, C, n)]
    lr, pr, dl = torch.zeros(n, device=dev), Mat.zeros(n, C, dev), Mat.zeros(n, C, dev)
    ops.class_loss(Mat.from_numpy(x, dev), Mat.from_numpy(z, dev), n, C, sig, lr, pr, dl)
    _sync()
    loss, dlog = orc.classification_loss(x.astype(np.float64), z.astype(np.float64), sig)
    np.testing.assert_allclose(lr.cpu().numpy().mean(), loss, rtol=1e-4)
    np.testing.assert_allclose(dl.numpy(), dlog, rtol=1e-4, atol=1e-7)
    wp = orc.sigmoid(x.astype(np.float64)) if sig else orc.softmax(x.astype(np.float64))
    np.testing.assert_allclose(pr.numpy(), wp, **TOL)


# ----------------------------------------------------------------------------- K6
def test_adam_matches_tf_formula(dev):
    rng = np.random.default_rng(25)
    cnt = 230185
    p = _asym(rng, (cnt,))
    g = _asym(rng, (cnt,)) * 8  # exercises the +-5 clip
    pd, m, v = torch.from_numpy(p.copy()).to(dev), torch.zeros(cnt, device=dev), torch.zeros(cnt, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    po, mo, vo = p.copy(), np.zeros(cnt, np.float32), np.zeros(cnt, np.float32)
    for t in range(1, 4):
        ops.adam_step(pd, torch.from_numpy(g).to(dev), m, v, cnt, 0.01, step)
        ops.advance_counter(step, 1)
        orc.adam_tf_update(po, orc.clip_by_value(g), mo, vo, t, 0.01)
    _sync()
    np.testing.assert_allclose(pd.cpu().numpy(), po, rtol=1e-5, atol=1e-6)


def test_input_grad_pull(dev):
    """gs_input_grad_pull == act_bwd(self rows) followed by one accumulate mean_bwd per hop (3-layer shape: the rows
    of the middle hop are self rows AND neighbor rows), up to fma contraction of scale*x + y (last bit)."""
    rng = np.random.default_rng(41)
    B, s2, s1, d = 6, 3, 5, 22
    n_self = B + B * s2                       # self rows of hops 0 and 1
    rows = n_self + B * s2 * s1               # + neighbor rows of hop 1
    d_self = _asym(rng, (n_self, d))
    dm = _asym(rng, (n_self, d))              # d_means of hop 0 (B rows) and hop 1 (B*s2 rows)
    y = _asym(rng, (rows, d))
    dS, dM, Y = Mat.from_numpy(d_self, dev), Mat.from_numpy(dm, dev), Mat.from_numpy(y, dev)
    want = Mat.zeros(rows, d, dev)
    ops.act_bwd(dS, Y.rows_slice(0, n_self), n_self, d, ops.ACT_RELU, want.rows_slice(0, n_self))
    ops.mean_bwd(dM.rows_slice(0, B), B, s2, 1.0 / s2, want.rows_slice(B, B + B * s2), mask_y=Y.rows_slice(B, B + B * s2),
                 accumulate=True)
    ops.mean_bwd(dM.rows_slice(B, n_self), B * s2, s1, 1.0 / s1, want.rows_slice(n_self, rows),
                 mask_y=Y.rows_slice(n_self, rows), accumulate=False)
    got = Mat.from_numpy(np.full((rows, d), np.nan, np.float32), dev)
    ops.input_grad_pull(got, rows, d, d_self=dS, n_self=n_self,
                        segments=[(dM.rows_slice(0, B), B, B, s2, 1.0 / s2), (dM.rows_slice(B, n_self), n_self, B * s2, s1, 1.0 / s1)],
                        mask_y=Y)
    _sync()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    ref = np.zeros((rows, d), np.float32)
    ref[:n_self] += d_self
    ref[B:B + B * s2] += np.repeat(dm[:B] * np.float32(1.0 / s2), s2, axis=0)
    ref[n_self:] += np.repeat(dm[B:] * np.float32(1.0 / s1), s1, axis=0)
    np.testing.assert_allclose(got.numpy(), ref * (y > 0), rtol=1e-5, atol=1e-6)


def test_scatter_add_rows_and_copy_cols(dev):
    """Identity features: gradient scatter (atomics; duplicate ids accumulate) and the table refresh."""
    rng = np.random.default_rng(42)
    N, n, s, c, ld = 50, 40, 7, 6, 16
    d = _asym(rng, (n, c))
    ids = rng.integers(0, N, size=n * s).astype(np.int32)
    table = Mat.zeros(N, ld, dev)
    ops.scatter_add_rows(Mat.from_numpy(d, dev), n, s, c, 0.25, torch.from_numpy(ids).to(dev), table)
    _sync()
    want = np.zeros((N, ld), np.float64)
    np.add.at(want, (ids[:, None], np.arange(c)[None, :]), np.repeat(d.astype(np.float64) * 0.25, s, axis=0))
    np.testing.assert_allclose(table.numpy(), want, rtol=1e-5, atol=1e-6)
    dst = Mat.from_numpy(np.full((N, 24), 7.0, np.float32), dev)
    ops.copy_cols(table, dst, N, c)
    _sync()
    out = dst.numpy()
    assert np.array_equal(out[:, :c], table.numpy()[:, :c]) and np.all(out[:, c:] == 7.0)


def test_sum_kernels(dev):
    x = torch.arange(10000, dtype=torch.float32, device=dev) / 1000.0
    out = torch.zeros(1, device=dev)
    ops.sum_scaled(x, 10000, 0.5, out)
    ops.sumsq_scaled(x, 10000, 2.0, out, accumulate=True)
    _sync()
    xn = x.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(out.item(), 0.5 * xn.sum() + 2.0 * (xn ** 2).sum(), rtol=1e-5)


# ----------------------------------------------------------------------------- graphs
def test_graph_capture_replay(dev):
    st = ops.Stream()
    x = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    g = ops.Graph(st.handle)
    g.begin()
    ops.advance_counter(x, 3, stream=st.handle)
    ops.advance_counter(x, 4, stream=st.handle)
    g.end()
    for _ in range(5):
        g.launch()
    st.sync()
    assert int(x.item()) == 35


# ----------------------------------------------------------------------------- grouped / fused launches
def test_grouped_wgrad_and_flat_reduce_adam(dev):
    """All weight gradients in one launch (+ bias gradient as ones^T dZ), then slab reduce + clip + Adam in one."""
    import ctypes
    from graphsage_amd import _lib
    rng = np.random.default_rng(31)
    Nn, d, out, n = 1500, 602, 128, 700
    X = _asym(rng, (Nn, d))
    idx = rng.integers(0, Nn, size=n).astype(np.int32)
    mean = _asym(rng, (n, d))
    dZ = _asym(rng, (n, 2 * out))
    Xd, Md, dZd, idxd = Mat.from_numpy(X, dev, 32), Mat.from_numpy(mean, dev), Mat.from_numpy(dZ, dev), _i32(idx, dev)
    ones = Mat(torch.ones((n, 4), device=dev), 1)
    # flat layout: W_self [d,out] | W_neigh [d,out] | bias [1, 2*out]
    sizes = [d * out, d * out, 2 * out]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    nsl = [3, 5, 2]
    slabs = [torch.zeros(nsl[i] * sizes[i], device=dev) for i in range(3)]
    descs = (_lib.WgradDesc * 3)()
    for i, (A, aidx, col0, dd, od) in enumerate([(Xd, idxd, 0, d, out), (Md, None, out, d, out), (ones, None, 0, 1, 2 * out)]):
        descs[i].A, descs[i].a_idx, descs[i].dZ, descs[i].slabs = A.ptr, ops.ptr(aidx), dZd.ptr, slabs[i].data_ptr()
        descs[i].lda, descs[i].ldz, descs[i].ld_slab, descs[i].n = A.ld, dZd.ld, od, n
        descs[i].d, descs[i].col0, descs[i].out_dim, descs[i].n_slabs = dd, col0, od, nsl[i]
    ops.call("gs_dense_wgrad_grouped", ctypes.addressof(descs), 3, ops.current_stream())
    total = int(offs[-1])
    p0 = _asym(rng, (total,))
    params, grads = torch.from_numpy(p0.copy()).to(dev), torch.zeros(total, device=dev)
    m, v = torch.zeros(total, device=dev), torch.zeros(total, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    vd = (_lib.VarDesc * 3)()
    for i in range(3):
        vd[i].offset, vd[i].size, vd[i].slabs, vd[i].n_slabs, vd[i].decay = int(offs[i]), sizes[i], slabs[i].data_ptr(), nsl[i], int(i < 2)
    wd = 0.01
    # the step's scalar loss is folded into the same launch (workgroup 0): loss_out = 7 + mean(loss_rows)
    loss_rows_h = rng.random(700).astype(np.float32)
    loss_rows = torch.from_numpy(loss_rows_h).to(dev)
    loss_out = torch.full((1,), 7.0, device=dev)
    ops.call("gs_flat_reduce_adam", ctypes.addressof(vd), 3, ops.ptr(params), ops.ptr(grads), ops.ptr(m), ops.ptr(v), total,
             wd, 1, 0.01, 0.9, 0.999, 1e-8, 5.0, 1.0, ops.ptr(step), 1, ops.ptr(loss_rows), 700, 1.0 / 700, ops.ptr(loss_out), 1,
             ops.current_stream())
    _sync()
    np.testing.assert_allclose(float(loss_out.item()), 7.0 + loss_rows_h.astype(np.float64).mean(), rtol=1e-5)
    want = np.concatenate([(X[idx].astype(np.float64).T @ dZ[:, :out]).reshape(-1) + wd * p0[:sizes[0]],
                           (mean.astype(np.float64).T @ dZ[:, out:]).reshape(-1) + wd * p0[offs[1]:offs[2]],
                           dZ.astype(np.float64).sum(0)])
    assert_close_rownorm(grads.cpu().numpy(), want)
    pw = p0.copy()
    orc.adam_tf_update(pw, orc.clip_by_value(want.astype(np.float32)), np.zeros(total, np.float32), np.zeros(total, np.float32), 1, 0.01)
    np.testing.assert_allclose(params.cpu().numpy(), pw, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("concat", [True, False])
def test_sage_dense_dgrad_two_outputs(dev, concat):
    rng = np.random.default_rng(32)
    n, d_in, o = 5632, 256, 128
    dZ = _asym(rng, (n, o * (2 if concat else 1)))
    Ws, Wn = _asym(rng, (d_in, o)), _asym(rng, (d_in, o))
    t2 = Mat.zeros(n, 2 * d_in, dev)
    ops.sage_dense_dgrad(Mat.from_numpy(dZ, dev), n, o, concat, Mat.from_numpy(Ws, dev), Mat.from_numpy(Wn, dev), d_in, t2)
    _sync()
    zs, zn = (dZ[:, :o], dZ[:, o:]) if concat else (dZ, dZ)
    np.testing.assert_allclose(t2.cols_slice(0, d_in).numpy(), zs.astype(np.float64) @ Ws.T.astype(np.float64), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(t2.cols_slice(d_in, 2 * d_in).numpy(), zn.astype(np.float64) @ Wn.T.astype(np.float64), rtol=1e-4, atol=1e-3)


def test_stage_batch_and_counters(dev):
    rng = np.random.default_rng(33)
    N, C, n = 500, 41, 64
    table = _asym(rng, (N + 1, C))
    order = rng.permutation(N)[:200].astype(np.int32)
    cur = torch.tensor([190], dtype=torch.int64, device=dev)
    c2 = torch.tensor([5], dtype=torch.int64, device=dev)
    batch = torch.zeros(n, dtype=torch.int32, device=dev)
    lab = Mat.zeros(n, C, dev)
    ops.stage_batch(_i32(order, dev), cur, n, batch, Mat.from_numpy(table, dev), lab)
    ops.call("gs_advance_counters", ops.ptr(cur), n, None, 0, ops.ptr(c2), 7, ops.current_stream())
    _sync()
    want = order[(190 + np.arange(n)) % 200]
    assert np.array_equal(batch.cpu().numpy(), want)
    assert np.array_equal(lab.numpy(), table[want])
    assert int(cur.item()) == 254 and int(c2.item()) == 12


@pytest.mark.parametrize("fans,B", [([10, 25], 512), ([3, 2, 4], 37), ([7], 100), ([64, 2], 5)])
def test_fused_fanout_sampler_bit_exact(dev, fans, B):
    """ONE launch == hop-by-hop gs_sample_uniform_csr == the oracle hash; plus batch/label staging."""
    rng = np.random.default_rng(sum(fans) + B)
    N, C = 4000, 41
    rowptr, col = _rand_csr(rng, N, 60)
    order = rng.permutation(N).astype(np.int32)
    table = _asym(rng, (N + 1, C))
    sizes = [B]
    for f in fans:
        sizes.append(sizes[-1] * f)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ids_all = torch.full((int(offs[-1]),), -7, dtype=torch.int32, device=dev)
    cur = torch.tensor([N - 20], dtype=torch.int64, device=dev)   # wraps around the epoch order
    clock = torch.tensor([5], dtype=torch.int64, device=dev)
    lab = Mat.zeros(B, C, dev)
    rp, cl = torch.from_numpy(rowptr).to(dev), _i32(col, dev)
    ops.sample_fanout_csr(rp, cl, N, N, fans, offs.tolist(), ids_all, B, 123, step_dev=clock, hop0=1, root_offset=1000,
                          order=_i32(order, dev), cursor_dev=cur, label_table=Mat.from_numpy(table, dev), labels_out=lab)
    _sync()
    got = ids_all.cpu().numpy()
    roots = order[(N - 20 + np.arange(B)) % N]
    assert np.array_equal(got[:B], roots)
    assert np.array_equal(lab.numpy(), table[roots])
    prev, support = roots, 1
    for h, f in enumerate(fans):
        want = sampler_hash.sample_uniform_csr(rowptr, col, N, N, prev, f, 123, 5, 1 + h, global_row_offset=1000 * support)
        seg = got[offs[h + 1]:offs[h + 2]]
        assert np.array_equal(seg.reshape(-1, f), want), "hop %d" % h
        dev_hop = ops.sample_uniform_csr(rp, cl, N, N, _i32(prev, dev), f, 123, step=5, hop=1 + h,
                                         global_row_offset=1000 * support).cpu().numpy()
        assert np.array_equal(seg, dev_hop)
        prev, support = want.reshape(-1), support * f


@pytest.mark.parametrize("sig,C,d,n", [(False, 41, 256, 512), (True, 121, 256, 300), (False, 7, 64, 37), (True, 64, 64, 5000),
                                       (False, 3, 128, 9), (True, 100, 256, 70), (False, 60, 512, 33)])
def test_fused_head_fwd_bwd(dev, sig, C, d, n):
    rng = np.random.default_rng(C + d)
    x = _asym(rng, (n, d))
    x[min(3, n - 1)] = 0                                            # clamped row (sum sq < 1e-12)
    W, b = _asym(rng, (d, C)) * 0.3, _asym(rng, (C,)) * 0.1
    z = (rng.random((n, C)) > 0.5).astype(np.float32) if sig else np.eye(C, dtype=np.float32)[rng.integers(0, C, n)]
    y, lo, pr, dl = Mat.zeros(n, d, dev), Mat.zeros(n, C, dev), Mat.zeros(n, C, dev), Mat.zeros(n, C, dev)
    lr, dx = torch.zeros(n, device=dev), Mat.zeros(n, d, dev)
    ops.head_fwd_bwd(Mat.from_numpy(x, dev), n, Mat.from_numpy(W, dev), torch.from_numpy(b).to(dev), Mat.from_numpy(z, dev),
                     C, sig, y, lo, pr, dl, lr, dx)
    _sync()
    x64, W64 = x.astype(np.float64), W.astype(np.float64)
    wy, cache = orc.l2_normalize_fwd(x64)
    logits = wy @ W64 + b
    loss, dlog = orc.classification_loss(logits, z.astype(np.float64), sig)
    np.testing.assert_allclose(y.numpy(), wy, **TOL)
    np.testing.assert_allclose(lo.numpy(), logits, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lr.cpu().numpy().mean(), loss, rtol=1e-4)
    np.testing.assert_allclose(dl.numpy(), dlog, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(pr.numpy(), orc.sigmoid(logits) if sig else orc.softmax(logits), **TOL)
    want_dx = orc.l2_normalize_bwd(dlog @ W64.T, cache)
    keep = np.ones(n, bool); keep[min(3, n - 1)] = False
    np.testing.assert_allclose(dx.numpy()[keep], want_dx[keep], rtol=1e-4, atol=1e-6)


def test_sage_dense_cogather_equals_separate_calls(dev):
    """Horizontally fused launch (GEMM tiles + gather waves) == the two separate launches, bitwise."""
    rng = np.random.default_rng(41)
    Nn, d, out, n = 3000, 602, 128, 2500   # > 2048 rows: the fused 64x64 path (smaller n falls back to separate launches)
    X = _asym(rng, (Nn + 1, d)); X[Nn] = 0
    mean = _asym(rng, (n, d))
    Ws, Wn = _asym(rng, (d, out)) * 0.1, _asym(rng, (d, out)) * 0.1
    sidx = rng.integers(0, Nn, size=n).astype(np.int32)
    idx_a = rng.integers(0, Nn + 1, size=700 * 25).astype(np.int32)
    idx_b = rng.integers(0, Nn + 1, size=90 * 10).astype(np.int32)
    Xd, Md, Wsd, Wnd = Mat.from_numpy(X, dev, 32), Mat.from_numpy(mean, dev), Mat.from_numpy(Ws, dev), Mat.from_numpy(Wn, dev)
    ia, ib, si = _i32(idx_a, dev), _i32(idx_b, dev), _i32(sidx, dev)
    o1, o2 = Mat.zeros(n, 2 * out, dev), Mat.zeros(n, 2 * out, dev)
    ma1, mb1, ma2, mb2 = Mat.zeros(700, d, dev), Mat.zeros(90, d, dev), Mat.zeros(700, d, dev), Mat.zeros(90, d, dev)
    ops.sage_dense_fwd(Xd, si, Md, None, n, Wsd, Wnd, out, True, ops.ACT_RELU, None, o1)
    ops.gather_mean_fwd(Xd, ia, 700, 25, out=ma1)
    ops.gather_mean_fwd(Xd, ib, 90, 10, out=mb1)
    jobs = [ops.gather_job(Xd, ia, 700, 25, ma2), ops.gather_job(Xd, ib, 90, 10, mb2)]
    ops.sage_dense_fwd_cogather(Xd, si, Md, None, n, Wsd, Wnd, out, True, ops.ACT_RELU, None, o2, jobs)
    _sync()
    assert np.array_equal(o1.numpy(), o2.numpy())
    assert np.array_equal(ma1.numpy(), ma2.numpy()) and np.array_equal(mb1.numpy(), mb2.numpy())
    np.testing.assert_allclose(ma2.numpy(), X[idx_a].reshape(700, 25, d).mean(1), **TOL)


@pytest.mark.parametrize("n,s,D,O,C,sig,train", [(512, 10, 256, 128, 41, False, True), (37, 3, 128, 64, 7, True, True),
                                                 (100, 10, 256, 64, 33, True, True), (48, 5, 128, 128, 64, False, True),
                                                 (33, 4, 256, 128, 41, False, False),
                                                 # two class groups (64 < C <= 128): PPI's 121 sigmoid labels at the
                                                 # reference example's shapes (example_supervised.sh), softmax, ragged C
                                                 (512, 10, 256, 128, 121, True, True), (70, 6, 128, 64, 100, False, True),
                                                 (45, 3, 256, 64, 65, True, True), (90, 11, 128, 128, 128, False, True),
                                                 (21, 2, 256, 128, 127, False, False),
                                                 (3000, 10, 256, 128, 41, False, True)])     # 940 workgroups > 256 CUs
def test_fused_tail_fwd_bwd(dev, n, s, D, O, C, sig, train):
    """gs_sage_tail_fwd_bwd (layer 1 + l2_normalize + head + loss + every input gradient, ONE launch) vs the oracle's
    MeanAggregator / head restatements (aggregators.py:43-64, supervised_models.py:85-126) in fp64; ragged n, both
    losses, C inside one 64-class group and across two (C <= 128), device counters."""
    rng = np.random.default_rng(n + C)
    rows = n + n * s
    h0 = np.maximum(_asym(rng, (rows, D)), 0).astype(np.float32)            # relu outputs of layer 0 (zeros included)
    Ws, Wn = _asym(rng, (D, O)) * 0.2, _asym(rng, (D, O)) * 0.2
    Wh, bh = _asym(rng, (2 * O, C)) * 0.3, _asym(rng, (C,)) * 0.1
    lab = (rng.random((n, C)) > 0.5).astype(np.float32) if sig else np.eye(C, dtype=np.float32)[rng.integers(0, C, n)]
    Z = 2 * O
    assert ops.sage_tail_supported(D, O, C) and not ops.sage_tail_supported(D, O, 129) and not ops.sage_tail_supported(96, O, C)
    means, z, y = Mat.zeros(n, D, dev), Mat.zeros(n, Z, dev), Mat.zeros(n, Z, dev)
    lo, pr, dl = Mat.zeros(n, C, dev), Mat.zeros(n, C, dev), Mat.zeros(n, C, dev)
    lr = torch.zeros(n, device=dev)
    dz, dh0 = Mat.zeros(n, Z, dev), Mat.zeros(rows, D, dev)
    c0, c1 = torch.full((1,), 5, dtype=torch.int64, device=dev), torch.full((1,), 100, dtype=torch.int64, device=dev)
    ops.sage_tail_fwd_bwd(Mat.from_numpy(h0, dev), n, s, Mat.from_numpy(Ws, dev), Mat.from_numpy(Wn, dev), O,
                          Mat.from_numpy(Wh, dev), torch.from_numpy(bh).to(dev), Mat.from_numpy(lab, dev), C, sig, means, z, y,
                          lo, pr, dl, lr, dz=dz if train else None, d_h0=dh0 if train else None,
                          counters=[(c0, 1), (None, 3), (c1, n)])
    _sync()
    assert int(c0.item()) == 6 and int(c1.item()) == 100 + n
    h64 = h0.astype(np.float64)
    self_v, neigh = h64[:n], h64[n:].reshape(n, s, D)
    zz, cache = orc.mean_aggregator_fwd(self_v, neigh, Ws.astype(np.float64), Wn.astype(np.float64), True, "id")
    np.testing.assert_allclose(means.numpy(), neigh.mean(1), **TOL)
    np.testing.assert_allclose(z.numpy(), zz, rtol=1e-4, atol=1e-4)
    wy, ncache = orc.l2_normalize_fwd(zz)
    np.testing.assert_allclose(y.numpy(), wy, **TOL)
    logits = wy @ Wh.astype(np.float64) + bh
    loss, dlog = orc.classification_loss(logits, lab.astype(np.float64), sig)
    np.testing.assert_allclose(lo.numpy(), logits, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(pr.numpy(), orc.sigmoid(logits) if sig else orc.softmax(logits), **TOL)
    np.testing.assert_allclose(lr.cpu().numpy().astype(np.float64).mean(), loss, rtol=1e-4)
    np.testing.assert_allclose(dl.numpy(), dlog, rtol=1e-4, atol=1e-7)
    if not train:
        assert float(dz.buf.abs().max().item()) == 0.0 and float(dh0.buf.abs().max().item()) == 0.0
        return
    d_z = orc.l2_normalize_bwd(dlog @ Wh.astype(np.float64).T, ncache)
    np.testing.assert_allclose(dz.numpy(), d_z, rtol=1e-4, atol=1e-4 * np.abs(d_z).max())
    d_self, d_neigh, _ = orc.mean_aggregator_bwd(d_z, cache, Ws.astype(np.float64), Wn.astype(np.float64), True, "id")
    want = np.concatenate([d_self, d_neigh.reshape(n * s, D)], axis=0) * (h64 > 0)
    np.testing.assert_allclose(dh0.numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(want).max())


@pytest.mark.parametrize("n,s,D,O,C,sig", [(512, 10, 256, 128, 41, False), (3000, 10, 256, 128, 41, False),
                                           (200, 7, 128, 64, 121, True), (77, 3, 256, 64, 9, False)])
def test_fused_tail_split_form_is_bit_identical(dev, n, s, D, O, C, sig):
    """The split form of the fused tail (gs_sage_tail_z: lean z-helper launch, then gs_sage_tail_fwd_bwd with z_ready: no
    helper workgroups, no in-kernel hand-over, no sync buffer) gives the bits of the one-launch form -- every output, with
    gather riders in both launches -- and the device counters advance once."""
    rng = np.random.default_rng(n + C)
    rows, Z = n + n * s, 2 * O
    h0 = Mat.from_numpy(np.maximum(_asym(rng, (rows, D)), 0).astype(np.float32), dev)
    Ws, Wn = Mat.from_numpy(_asym(rng, (D, O)) * 0.2, dev), Mat.from_numpy(_asym(rng, (D, O)) * 0.2, dev)
    Wh, bh = Mat.from_numpy(_asym(rng, (Z, C)) * 0.3, dev), torch.from_numpy(_asym(rng, (C,)) * 0.1).to(dev)
    labn = (rng.random((n, C)) > 0.5).astype(np.float32) if sig else np.eye(C, dtype=np.float32)[rng.integers(0, C, n)]
    lab = Mat.from_numpy(labn, dev)
    Xg = Mat.from_numpy(_asym(rng, (5000, 602)), dev, ld_multiple=32)
    idx = _i32(rng.integers(0, 5000, size=1200 * 25), dev)
    outs = []
    for split in (False, True):
        means, z, y = Mat.zeros(n, D, dev), Mat.zeros(n, Z, dev), Mat.zeros(n, Z, dev)
        lo, pr, dl = Mat.zeros(n, C, dev), Mat.zeros(n, C, dev), Mat.zeros(n, C, dev)
        lr, dz, dh0 = torch.zeros(n, device=dev), Mat.zeros(n, Z, dev), Mat.zeros(rows, D, dev)
        g1, g2 = Mat.zeros(700, 602, dev), Mat.zeros(500, 602, dev)
        c0 = torch.full((1,), 5, dtype=torch.int64, device=dev)
        j1 = [ops.gather_job(Xg, idx[:700 * 25], 700, 25, g1)]
        j2 = [ops.gather_job(Xg, idx[700 * 25:], 500, 25, g2)]
        _sync()
        ops.sage_tail_fwd_bwd(h0, n, s, Ws, Wn, O, Wh, bh, lab, C, sig, means, z, y, lo, pr, dl, lr, dz=dz, d_h0=dh0,
                              counters=[(c0, 2)], jobs=j1 + ([] if split else j2), split=split, jobs_z=j2 if split else ())
        _sync()
        assert int(c0.item()) == 7
        outs.append([m.numpy() for m in (means, z, y, lo, pr, dl, dz, dh0, g1, g2)] + [lr.cpu().numpy()])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert np.abs(outs[1][7]).max() > 0 and np.abs(outs[1][9]).max() > 0


@pytest.mark.parametrize("n,s,D,O", [(1044, 10, 256, 128), (37, 3, 128, 64), (3000, 11, 256, 64), (100, 5, 128, 128)])
def test_last_layer_z_and_dh0_launches(dev, n, s, D, O):
    """gs_sage_tail_z (neighbor mean + both contractions + concat of a last mean layer, aggregators.py:48-58) and its
    backward twin gs_sage_tail_dh0 (dz . W^T for both terms + relu mask + 1/s broadcast) as stand-alone launches vs the
    oracle's MeanAggregator forward / backward in fp64 -- ragged n, a gather job riding in each launch."""
    rng = np.random.default_rng(n + D + O)
    rows, Z = n + n * s, 2 * O
    h0n = np.maximum(_asym(rng, (rows, D)), 0).astype(np.float32)
    h0 = Mat.from_numpy(h0n, dev)
    Wsn, Wnn = _asym(rng, (D, O)) * 0.2, _asym(rng, (D, O)) * 0.2
    Ws, Wn = Mat.from_numpy(Wsn, dev), Mat.from_numpy(Wnn, dev)
    Xg = Mat.from_numpy(_asym(rng, (3000, 602)), dev, ld_multiple=32)
    idx = rng.integers(0, 3000, size=(400, 25)).astype(np.int32)
    idx_d = _i32(idx.reshape(-1), dev)
    g1, g2 = Mat.zeros(400, 602, dev), Mat.zeros(400, 602, dev)
    means, z = Mat.zeros(n, D, dev, 32), Mat.zeros(n, Z, dev)
    ops.sage_tail_z(h0, n, s, Ws, Wn, O, means, z, jobs=[ops.gather_job(Xg, idx_d, 400, 25, g1)])
    _sync()
    h64 = h0n.astype(np.float64)
    self_v, neigh = h64[:n], h64[n:].reshape(n, s, D)
    zz, cache = orc.mean_aggregator_fwd(self_v, neigh, Wsn.astype(np.float64), Wnn.astype(np.float64), True, "id")
    np.testing.assert_allclose(means.numpy(), neigh.mean(1), **TOL)
    np.testing.assert_allclose(z.numpy(), zz, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(g1.numpy(), Xg.numpy()[idx].mean(axis=1), **TOL)
    dzn = _asym(rng, (n, Z)).astype(np.float32)
    dz, dh0 = Mat.from_numpy(dzn, dev), Mat.zeros(rows, D, dev)
    dh0.buf.fill_(float("nan"))                                             # every row must be written
    ops.sage_tail_dh0(h0, n, s, Ws, Wn, O, dz, dh0, jobs=[ops.gather_job(Xg, idx_d, 400, 25, g2)])
    _sync()
    d_self, d_neigh, _ = orc.mean_aggregator_bwd(dzn.astype(np.float64), cache, Wsn.astype(np.float64), Wnn.astype(np.float64), True, "id")
    want = np.concatenate([d_self, d_neigh.reshape(n * s, D)], axis=0) * (h64 > 0)
    np.testing.assert_allclose(dh0.numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(want).max())
    assert np.array_equal(g1.numpy(), g2.numpy())


def test_fused_tail_handover_stress(dev):
    """The in-kernel hand-over of the fused tail (helper workgroups -> row-group workgroups through tagged 8-byte granules,
    per-group launch epochs, bounded wait, error word) under stress: 4000 rows = 1250 workgroups (far more than resident at once),
    30 back-to-back launches on ONE hand-over buffer while a second stream keeps the chip busy with gathers, plus gather
    riders in the launch itself.  Every launch must give the same bits, the error word stays 0, and a second model
    sharing the device (own buffer) is not disturbed."""
    rng = np.random.default_rng(7)
    n, s, D, O, C = 4000, 10, 256, 128, 41
    rows, Z = n + n * s, 2 * O
    h0 = Mat.from_numpy(np.maximum(_asym(rng, (rows, D)), 0).astype(np.float32), dev)
    Ws, Wn = Mat.from_numpy(_asym(rng, (D, O)) * 0.2, dev), Mat.from_numpy(_asym(rng, (D, O)) * 0.2, dev)
    Wh, bh = Mat.from_numpy(_asym(rng, (Z, C)) * 0.3, dev), torch.from_numpy(_asym(rng, (C,)) * 0.1).to(dev)
    lab = Mat.from_numpy(np.eye(C, dtype=np.float32)[rng.integers(0, C, n)], dev)
    Xg = Mat.from_numpy(_asym(rng, (20000, 602)), dev, ld_multiple=32)
    idx = _i32(rng.integers(0, 20000, size=5120 * 25), dev)
    side = torch.cuda.Stream()
    sync = torch.zeros(ops.tail_sync_words(n, O), dtype=torch.int32, device=dev)
    main = ops.Stream()
    outs = []
    for it in range(30):
        means, z, y = Mat.zeros(n, D, dev), Mat.zeros(n, Z, dev), Mat.zeros(n, Z, dev)
        lo, pr, dl = Mat.zeros(n, C, dev), Mat.zeros(n, C, dev), Mat.zeros(n, C, dev)
        lr, dz, dh0 = torch.zeros(n, device=dev), Mat.zeros(n, Z, dev), Mat.zeros(rows, D, dev)
        gm = Mat.zeros(5120, 602, dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):                                    # concurrent work on another stream
            for _ in range(3):
                ops.gather_mean_fwd(Xg, idx, 5120, 25, out=Mat.zeros(5120, 602, dev), stream=side.cuda_stream)
        jobs = [ops.gather_job(Xg, idx, 5120, 25, gm)] if it % 2 else []
        ops.sage_tail_fwd_bwd(h0, n, s, Ws, Wn, O, Wh, bh, lab, C, False, means, z, y, lo, pr, dl, lr, dz=dz, d_h0=dh0,
                              jobs=jobs, stream=main.handle, sync=sync)
        ops.call("gs_stream_sync", main.handle)
        torch.cuda.synchronize()
        assert ops.tail_sync_error(sync, n) == 0, "launch %d" % it
        outs.append((z.numpy(), dz.numpy(), dh0.numpy(), lr.cpu().numpy()))
        for a, b in zip(outs[0], outs[-1]):
            assert np.array_equal(a, b), "launch %d differs from launch 0" % it
    G = (n + 15) // 16
    st = sync.cpu().numpy().astype(np.int64)
    assert (st[G:2 * G] == 30).all() and st[2 * G] == 0                 # every group's epoch advanced once per launch
    h64 = h0.numpy().astype(np.float64)
    zz, _ = orc.mean_aggregator_fwd(h64[:n], h64[n:].reshape(n, s, D), Ws.numpy().astype(np.float64),
                                    Wn.numpy().astype(np.float64), True, "id")
    np.testing.assert_allclose(outs[0][0], zz, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,d,out,two,act,bias,gathered", [
    (5632, 602, 128, True, ops.ACT_RELU, False, True), (2500, 100, 128, True, ops.ACT_IDENTITY, True, False),
    (3001, 602, 256, False, ops.ACT_RELU, False, False), (2049, 37, 40, True, ops.ACT_RELU, True, True),
    (70, 20, 6, True, ops.ACT_RELU, True, True), (33, 8, 64, False, ops.ACT_IDENTITY, False, False)])
def test_sage_dense_fwd_stream(dev, n, d, out, two, act, bias, gathered):
    """gs_sage_dense_fwd_stream (split-K contraction workgroups; gathered or dense self rows, K tails and K too short
    for some of the four K quarters, ragged rows/columns) + the co-scheduled gather jobs vs NumPy."""
    rng = np.random.default_rng(n + d)
    Nn = 4000
    X = _asym(rng, (Nn + 1, d)); X[Nn] = 0
    self_m, mean = _asym(rng, (n, d)), _asym(rng, (n, d))
    self_ids = rng.integers(0, Nn + 1, size=n).astype(np.int32)
    if gathered:
        self_m = X[self_ids]
    Ws, Wn = _asym(rng, (d, out)) * 0.1, _asym(rng, (d, out)) * 0.1
    b = (_asym(rng, ((2 if two else 1) * out,)) * 0.1) if bias else None
    idx = rng.integers(0, Nn + 1, size=(700, 25)).astype(np.int32)
    ids1 = rng.integers(0, Nn + 1, size=900).astype(np.int32)
    Xd = Mat.from_numpy(X, dev, 32)
    # junk in the pad columns of the dense operands must not leak into the result (ld = round_up(d, 4) only)
    sd, md = Mat.from_numpy(self_m, dev), Mat.from_numpy(mean, dev)
    outm = Mat.zeros(n, (2 if two else 1) * out, dev)
    g_out, c_out = Mat.zeros(700, d, dev), Mat.zeros(900, d, dev)
    idx_d, ids1_d = _i32(idx.reshape(-1), dev), _i32(ids1, dev)        # descriptors hold raw pointers: keep the tensors
    Wsd, Wnd = Mat.from_numpy(Ws, dev), Mat.from_numpy(Wn, dev)
    bd = torch.from_numpy(b).to(dev) if bias else None
    jobs = [ops.gather_job(Xd, idx_d, 700, 25, g_out), ops.gather_job(Xd, ids1_d, 900, 1, c_out)]
    sid_d = _i32(self_ids, dev)
    if gathered:
        ops.sage_dense_fwd_stream(Xd, sid_d, md, n, Wsd, Wnd, out, act, bd, outm, jobs)
    else:
        ops.sage_dense_fwd_stream(sd if two else None, None, md, n, Wsd if two else None, Wnd, out, act, bd, outm, jobs)
    _sync()
    want_n = mean.astype(np.float64) @ Wn
    want = np.concatenate([self_m.astype(np.float64) @ Ws, want_n], axis=1) if two else want_n
    if bias:
        want = want + b
    if act == ops.ACT_RELU:
        want = np.maximum(want, 0)
    assert_close_rownorm(outm.numpy(), want)
    np.testing.assert_allclose(g_out.numpy(), X[idx].mean(axis=1), **TOL)
    assert np.array_equal(c_out.numpy(), X[ids1])


@pytest.mark.parametrize("n,d_self,d_agg,out,gathered", [(5632, 602, 512, 128, True), (2100, 50, 1024, 64, False), (70, 602, 37, 6, True)])
def test_sage_dense_fwd_stream2_terms_of_different_length(dev, n, d_self, d_agg, out, gathered):
    """gs_sage_dense_fwd_stream2: the pooling aggregators' layer -- self rows of d_self features (gathered or dense) . W_self |
    pooled rows of d_agg = hidden_dim . W_neigh, concat + bias + relu -- vs NumPy."""
    rng = np.random.default_rng(n + d_self + d_agg)
    Nn = 3000
    X = _asym(rng, (Nn + 1, d_self)); X[Nn] = 0
    self_ids = rng.integers(0, Nn + 1, size=n).astype(np.int32)
    self_m = X[self_ids] if gathered else _asym(rng, (n, d_self))
    pooled = _asym(rng, (n, d_agg))
    Ws, Wn = _asym(rng, (d_self, out)) * 0.1, _asym(rng, (d_agg, out)) * 0.1
    b = _asym(rng, (2 * out,)) * 0.1
    Xd = Mat.from_numpy(X, dev, 32)
    sd, pd = Mat.from_numpy(self_m, dev), Mat.from_numpy(pooled, dev)
    outm = Mat.zeros(n, 2 * out, dev)
    sid_d = _i32(self_ids, dev)
    Wsd, Wnd, bd = Mat.from_numpy(Ws, dev), Mat.from_numpy(Wn, dev), torch.from_numpy(b).to(dev)
    if gathered:
        ops.sage_dense_fwd_stream2(Xd, sid_d, pd, n, Wsd, Wnd, out, ops.ACT_RELU, bd, outm)
    else:
        ops.sage_dense_fwd_stream2(sd, None, pd, n, Wsd, Wnd, out, ops.ACT_RELU, bd, outm)
    _sync()
    want = np.maximum(np.concatenate([self_m.astype(np.float64) @ Ws, pooled.astype(np.float64) @ Wn], axis=1) + b, 0)
    assert_close_rownorm(outm.numpy(), want)


@pytest.mark.parametrize("entry", ["gs_dense_wgrad_grouped_stream", "gs_dense_wgrad_grouped_tiled3"])
@pytest.mark.parametrize("slices0", [22, 11, 45, 6])
def test_dense_wgrad_grouped_stream(dev, slices0, entry):
    """gs_dense_wgrad_grouped_stream / _tiled3: the weight gradients of a mean step (layer 0: 602x128 x2 over 5632 rows, layer 1:
    256x128 x2 over 512 rows, head 256x41, bias 1x41, odd slice lengths) as split-K slabs + a co-scheduled gather job.
    (tiled3 at 45 slices: 128-row slices, the 45th slab's slice is EMPTY and must come out as zeros.)"""
    if entry.endswith("stream") and slices0 == 6:
        pytest.skip("the stream kernel keeps a gathered slice's <= 512 row offsets in registers")
    import ctypes
    from graphsage_amd import _lib
    rng = np.random.default_rng(77)
    probs = [(5631, 602, 128, 0, slices0), (5631, 602, 128, 128, slices0), (512, 256, 128, 0, 3), (512, 256, 128, 128, 2), (512, 256, 41, 0, 2),
             (512, 1, 41, 0, 2)]
    descs = (_lib.WgradDesc * len(probs))()
    keep, want, slabs = [], [], []
    for i, (n, d, o, col0, ns) in enumerate(probs):
        A = np.ones((n, 1), np.float32) if d == 1 else _asym(rng, (n, d))
        dZ = _asym(rng, (n, col0 + o)) * 0.1
        aidx = None
        if i == 0:          # layer-0 self term: rows gathered from a table through an index vector (cached in registers)
            table = _asym(rng, (3000, d))
            aidx = rng.integers(0, 3000, size=n).astype(np.int32)
            A = table[aidx]
            Ad, idx_dev = Mat.from_numpy(table, dev, 32), _i32(aidx, dev)
            keep.append(idx_dev)
        else:
            Ad = Mat.from_numpy(A, dev)
        Zd = Mat.from_numpy(dZ, dev)
        ld_slab = (o + 3) & ~3
        sl = torch.full((ns * d * ld_slab,), float("nan"), device=dev)      # every slab element inside [d, o] must be WRITTEN
        keep += [Ad, Zd, sl]
        descs[i].A, descs[i].a_idx, descs[i].dZ, descs[i].slabs = Ad.ptr, ops.ptr(idx_dev) if aidx is not None else None, Zd.ptr, sl.data_ptr()
        descs[i].lda, descs[i].ldz, descs[i].ld_slab, descs[i].n = Ad.ld, Zd.ld, ld_slab, n
        descs[i].d, descs[i].col0, descs[i].out_dim, descs[i].n_slabs = d, col0, o, ns
        descs[i].a_rows = 3000 if aidx is not None else 0
        want.append(A.astype(np.float64).T @ dZ[:, col0:].astype(np.float64))
        slabs.append((sl, ns, d, ld_slab, o))
    X = _asym(rng, (1000, 602))
    idx = rng.integers(0, 1000, size=(300, 10)).astype(np.int32)
    g_out = Mat.zeros(300, 602, dev)
    Xd, idx_d = Mat.from_numpy(X, dev, 32), _i32(idx.reshape(-1), dev)   # descriptors hold raw pointers: keep the tensors
    job = ops.gather_job(Xd, idx_d, 300, 10, g_out)
    jarr = (_lib.GatherDesc * 1)(job)
    ops.call(entry, ctypes.addressof(descs), len(probs), ctypes.addressof(jarr), 1, ops.current_stream())
    _sync()
    for (sl, ns, d, ld_slab, o), w in zip(slabs, want):
        got = sl.cpu().numpy().reshape(ns, d, ld_slab)[:, :, :o].astype(np.float64).sum(axis=0)
        assert_close_rownorm(got, w)
    np.testing.assert_allclose(g_out.numpy(), X[idx].mean(axis=1), **TOL)


def test_dense_wgrad_grouped_stream_refuses_tables_beyond_4gb(dev):
    """The stream weight-gradient kernel addresses a row-gathered operand with 32-bit BYTE offsets: a table beyond 4 GB (BASELINE
    configs[4]: 10^7 x 256 fp32 = 10.2 GB) is refused with an error, not wrapped around -- Engine.launch_wgrads sends such a
    pass to the tiled kernel instead (the wide-offset form was measured slower and removed: benchmarks/variants/README.md)."""
    import ctypes
    from graphsage_amd import _lib
    d, o, n = 256, 128, 1536
    small = Mat.zeros(n, d, dev)
    Zd = Mat.zeros(n, o, dev)
    sl = torch.zeros(6 * d * o, device=dev)
    descs = (_lib.WgradDesc * 1)()
    descs[0].A, descs[0].a_idx, descs[0].dZ, descs[0].slabs = small.ptr, ops.ptr(_i32(np.arange(n), dev)), Zd.ptr, sl.data_ptr()
    descs[0].lda, descs[0].ldz, descs[0].ld_slab, descs[0].n = small.ld, Zd.ld, o, n
    descs[0].d, descs[0].col0, descs[0].out_dim, descs[0].n_slabs, descs[0].a_rows = d, 0, o, 6, 4300000      # claims a 4.4 GB table
    jarr = (_lib.GatherDesc * 1)()
    with pytest.raises(_lib.GraphsageAmdError, match="32-bit offsets exceeded"):
        ops.call("gs_dense_wgrad_grouped_stream", ctypes.addressof(descs), 1, ctypes.addressof(jarr), 0, ops.current_stream())


