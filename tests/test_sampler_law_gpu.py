"""-m gpu: the sampling LAWS of the CSR sampler (GS_LAW_REFERENCE = the reference's joint law on a virtual padded table,
minibatch.py:227-245 + neigh_samplers.py:24-29; GS_LAW_DISTINCT = per-row without replacement) -- device kernels
(hop-by-hop, fused fan-out, fan-out riding in the optimizer launch) bit-exact vs oracle/sampler_hash.py and vs the
plain-loop known answers of tests/golden/law_kat.npz."""
import os

import numpy as np
import pytest
import torch

from graphsage_amd import ops
from graphsage_amd.ops import Mat
from oracle import graphsage_oracle as orc
from oracle import sampler_hash

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


def _graph(rng, N, max_deg, frac_zero=0.1):
    deg = rng.integers(1, max_deg + 1, size=N)
    deg[rng.random(N) < frac_zero] = 0
    deg[:3] = [128, 129, 127][: min(3, N)]
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, N, size=int(rowptr[-1])).astype(np.int32)
    return rowptr, col


def test_law_known_answers_on_device(dev):
    k = np.load(os.path.join(GOLD, "law_kat.npz"))
    M, s, seed, step, hop, row_off, pad = [int(v) for v in k["args"]]
    rp, cl, ids = torch.from_numpy(k["rowptr"]).to(dev), _i32(k["col"], dev), _i32(k["ids"], dev)
    for law, cap in ((1, M), (2, M), (2, 0)):
        got = ops.sample_uniform_csr(rp, cl, 5, pad, ids, s, seed, step=step, hop=hop, global_row_offset=row_off,
                                     law=law, max_degree=cap)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().reshape(-1, s), k["picked_law%d_cap%d" % (law, cap)]), (law, cap)


@pytest.mark.parametrize("law,cap", [(1, 128), (1, 25), (2, 0), (2, 128), (2, 10)])
@pytest.mark.parametrize("n,s", [(512, 10), (5120, 25), (65, 1), (37, 7)])
def test_law_csr_bit_exact(dev, law, cap, n, s):
    if law == 1 and s > cap:
        pytest.skip("num_samples > max_degree")
    rng = np.random.default_rng(n * 7 + s + law)
    N = 5000
    rowptr, col = _graph(rng, N, 700)
    ids = rng.integers(0, N + 1, size=n).astype(np.int32)
    ids[:3] = [0, 1, 2]
    for hop, step, off in [(0, 0, 0), (1, 17, 12345)]:
        want = sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids, s, 123, step, hop, off, law=law, max_degree=cap)
        got = ops.sample_uniform_csr(torch.from_numpy(rowptr).to(dev), _i32(col, dev), N, N, _i32(ids, dev), s, 123,
                                     step=step, hop=hop, global_row_offset=off, law=law, max_degree=cap)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().reshape(n, s), want)


def test_reference_law_equals_padded_sampler_on_virtual_table(dev):
    """GS_LAW_REFERENCE on CSR == gs_sample_padded (the exact reference sampler) on the materialised virtual table
    with the call's column permutation."""
    rng = np.random.default_rng(3)
    N, M, s = 3000, 128, 25
    rowptr, col = _graph(rng, N, 400)
    table = np.vstack([sampler_hash.virtual_padded_table(rowptr, col, N, N, 123, M), np.full((1, M), N, np.int32)])
    ids = rng.integers(0, N + 1, size=5120).astype(np.int32)
    cols = sampler_hash.call_columns(123, 9, 1, s, M)
    got = ops.sample_uniform_csr(torch.from_numpy(rowptr).to(dev), _i32(col, dev), N, N, _i32(ids, dev), s, 123, step=9,
                                 hop=1, law=1, max_degree=M).cpu().numpy().reshape(-1, s)
    via_table = ops.sample_padded(_i32(table, dev), _i32(ids, dev), _i32(cols, dev), s).cpu().numpy().reshape(-1, s)
    assert np.array_equal(got, via_table)
    assert np.array_equal(got, orc.uniform_neighbor_sampler(table, ids, s, cols))
    # rows of nodes with deg >= max_degree: s distinct column positions of max_degree distinct entries
    big = np.where(np.diff(rowptr)[np.minimum(ids, N - 1)] >= M)[0]
    big = big[ids[big] < N]
    assert len(big) > 100
    pos_distinct = [len(set(np.searchsorted(np.sort(table[ids[i]]), got[i]).tolist())) for i in big[:50]]
    assert min(pos_distinct) >= 1


@pytest.mark.parametrize("law,cap", [(1, 128), (2, 0), (2, 64)])
@pytest.mark.parametrize("fans,B", [([10, 25], 512), ([3, 2, 4], 37)])
def test_law_fused_fanout_bit_exact(dev, law, cap, fans, B):
    rng = np.random.default_rng(sum(fans) + B + law)
    N, C = 4000, 41
    rowptr, col = _graph(rng, N, 500)
    order = rng.permutation(N).astype(np.int32)
    table = rng.random((N + 1, C)).astype(np.float32)
    sizes = [B]
    for f in fans:
        sizes.append(sizes[-1] * f)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ids_all = torch.full((int(offs[-1]),), -7, dtype=torch.int32, device=dev)
    cur = torch.tensor([N - 20], dtype=torch.int64, device=dev)
    clock = torch.tensor([5], dtype=torch.int64, device=dev)
    lab = Mat.zeros(B, C, dev)
    rp, cl = torch.from_numpy(rowptr).to(dev), _i32(col, dev)
    ops.sample_fanout_csr(rp, cl, N, N, fans, offs.tolist(), ids_all, B, 123, step_dev=clock, hop0=1, root_offset=1000,
                          order=_i32(order, dev), cursor_dev=cur, label_table=Mat.from_numpy(table, dev), labels_out=lab,
                          law=law, max_degree=cap)
    torch.cuda.synchronize()
    got = ids_all.cpu().numpy()
    roots = order[(N - 20 + np.arange(B)) % N]
    assert np.array_equal(got[:B], roots)
    prev, support = roots, 1
    for h, f in enumerate(fans):
        want = sampler_hash.sample_uniform_csr(rowptr, col, N, N, prev, f, 123, 5, 1 + h, global_row_offset=1000 * support,
                                               law=law, max_degree=cap)
        assert np.array_equal(got[offs[h + 1]:offs[h + 2]].reshape(-1, f), want), "hop %d" % h
        prev, support = want.reshape(-1), support * f


def test_reference_law_rejects_bad_arguments(dev):
    rng = np.random.default_rng(0)
    rowptr, col = _graph(rng, 100, 10)
    rp, cl, ids = torch.from_numpy(rowptr).to(dev), _i32(col, dev), _i32(np.arange(10), dev)
    with pytest.raises(ops._lib.GraphsageAmdError, match="max_degree"):
        ops.sample_uniform_csr(rp, cl, 100, 100, ids, 5, 1, law=1, max_degree=0)
    with pytest.raises(ops._lib.GraphsageAmdError, match="num_samples"):
        ops.sample_uniform_csr(rp, cl, 100, 100, ids, 9, 1, law=1, max_degree=8)
    with pytest.raises(ops._lib.GraphsageAmdError, match="unknown law"):
        ops.sample_uniform_csr(rp, cl, 100, 100, ids, 5, 1, law=7)


def test_materialised_padded_table_matches_virtual_table_and_fanout(dev):
    """gs_build_padded_table == oracle/sampler_hash.virtual_padded_table entry for entry (the table the reference builds in
    minibatch.py:227-245 under the keyed law), and the fused fan-out sampler WITH the table (one lookup per draw) draws the
    ids of the table-free path and of the oracle bit for bit -- batch staging from the epoch order included."""
    rng = np.random.default_rng(12)
    N, M, B, fans = 5000, 128, 300, [10, 25]
    rowptr, col = _graph(rng, N, 600)
    rp, cl = torch.from_numpy(rowptr).to(dev), _i32(col, dev)
    table = ops.build_padded_table(rp, cl, N, N, M, 123)
    torch.cuda.synchronize()
    got_t = table.cpu().numpy().reshape(N + 1, M)
    want_t = sampler_hash.virtual_padded_table(rowptr, col, N, N, 123, M)
    assert np.array_equal(got_t[:N], want_t) and (got_t[N] == N).all()
    order = rng.permutation(N).astype(np.int32)
    sizes = [B, B * fans[0], B * fans[0] * fans[1]]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    cur = torch.tensor([17], dtype=torch.int64, device=dev)
    clock = torch.tensor([3], dtype=torch.int64, device=dev)
    outs = []
    for tab in (None, table):
        ids_all = torch.full((int(offs[-1]),), -7, dtype=torch.int32, device=dev)
        desc = ops.fanout_desc(rp, cl, N, N, fans, offs.tolist(), ids_all, B, 123, step_dev=clock, hop0=0, root_offset=64,
                               order=_i32(order, dev), cursor_dev=cur, law=1, max_degree=M, padded_table=tab)
        ops.sample_fanout_desc(desc)
        torch.cuda.synchronize()
        outs.append(ids_all.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    roots = order[(17 + np.arange(B)) % N]
    assert np.array_equal(outs[1][:B], roots)
    prev, support = roots, 1
    for h, f in enumerate(fans):
        want = sampler_hash.sample_uniform_csr(rowptr, col, N, N, prev, f, 123, 3, h, global_row_offset=64 * support, law=1,
                                               max_degree=M)
        assert np.array_equal(outs[1][offs[h + 1]:offs[h + 2]].reshape(-1, f), want), "hop %d" % h
        prev, support = want.reshape(-1), support * f
    with pytest.raises(ops._lib.GraphsageAmdError, match="padded_table"):
        ids_all = torch.zeros(int(offs[-1]), dtype=torch.int32, device=dev)
        ops.sample_fanout_desc(ops.fanout_desc(rp, cl, N, N, fans, offs.tolist(), ids_all, B, 123, law=0, padded_table=table))
