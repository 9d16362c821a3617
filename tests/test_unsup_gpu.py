"""-m gpu: N3 unsupervised objective -- kernels and SampleAndAggregate vs the NumPy oracle (models.py:332-405,
prediction.py:68-110) on identical neighbor sets and identical negatives."""
import os
import re

import numpy as np
import pytest
import torch

from graphsage_amd import engine as eng
from graphsage_amd import inits, ops
from graphsage_amd.minibatch import EdgeMinibatchIterator
from graphsage_amd.models import Placeholder, SAGEInfo, SampleAndAggregate
from graphsage_amd.neigh_samplers import AdjInfo, CSRAdjacency, PaddedAdjacency, UniformNeighborSampler
from graphsage_amd.ops import Mat
from graphsage_amd.utils import run_random_walks, synthetic_graph
from oracle import graphsage_oracle as orc
from oracle import sampler_hash

pytestmark = pytest.mark.gpu


def _sync():
    torch.cuda.synchronize()


def test_unsup_stage_bit_exact(dev):
    rng = np.random.default_rng(0)
    N, B, nn = 5000, 300, 20
    deg = rng.integers(0, 200, size=N)
    cdf = sampler_hash.unigram_cdf_u32(deg)
    pairs = rng.integers(0, N, size=(1000, 2)).astype(np.int32)
    ids = torch.full((2 * B + nn,), -1, dtype=torch.int32, device=dev)
    cur = torch.tensor([900], dtype=torch.int64, device=dev)
    clk = torch.tensor([7], dtype=torch.int64, device=dev)
    cdf_dev = torch.from_numpy(cdf.view(np.int32).copy()).to(dev)
    ops.call("gs_unsup_stage", ops.ptr(torch.from_numpy(pairs).to(dev)), 1000, ops.ptr(cur), B, ops.ptr(cdf_dev), N, nn,
             123, ops.ptr(clk), 0, ops.ptr(ids), ops.current_stream())
    _sync()
    got = ids.cpu().numpy()
    sel = pairs[(900 + np.arange(B)) % 1000]
    assert np.array_equal(got[:B], sel[:, 0]) and np.array_equal(got[B:2 * B], sel[:, 1])
    assert np.array_equal(got[2 * B:], sampler_hash.sample_unigram(cdf, nn, 123, 7))
    # distribution ~ degree^0.75: zero-degree nodes are never drawn, heavy nodes are drawn more often
    big = torch.empty(2 * 0 + 50000, dtype=torch.int32, device=dev)
    ops.call("gs_unsup_stage", None, 0, None, 0, ops.ptr(cdf_dev), N, 50000, 5, ops.ptr(clk), 0, ops.ptr(big), ops.current_stream())
    _sync()
    draws = big.cpu().numpy()
    assert (deg[draws] > 0).all()
    p = deg.astype(np.float64) ** 0.75
    p /= p.sum()
    top = np.argsort(-p)[:50]
    assert abs(np.isin(draws, top).mean() - p[top].sum()) < 0.01


@pytest.mark.parametrize("B,d,nn", [(512, 256, 20), (37, 64, 5), (130, 128, 20)])
def test_linkpred_fwd_bwd_vs_oracle(dev, B, d, nn):
    rng = np.random.default_rng(B + d)
    Y = rng.normal(size=(2 * B + nn, d)).astype(np.float32)
    Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    Y[:B] = 0.7 * Y[:B] + 0.3 * Y[B:2 * B]                     # correlated pairs so ranks are not all the same
    n_slabs = (B + 3) // 4
    loss_rows, rr = torch.zeros(B, device=dev), torch.zeros(B, device=dev)
    aff, dY = Mat.zeros(B, nn + 1, dev), Mat.zeros(2 * B + nn, d, dev)
    slabs = torch.zeros(n_slabs * nn * d, device=dev)
    import ctypes
    ns = ctypes.c_int32()
    Yd = Mat.from_numpy(Y, dev)
    ops.call("gs_linkpred_fwd_bwd", Yd.ptr, Yd.ld, B, d, nn, 1.0, 1.0 / B, ops.ptr(loss_rows), ops.ptr(rr), aff.ptr, aff.ld,
             dY.ptr, dY.ld, ops.ptr(slabs), ctypes.byref(ns), ops.current_stream())
    dneg = dY.rows_slice(2 * B, 2 * B + nn)
    ops.call("gs_reduce_slabs", ops.ptr(slabs), n_slabs, nn * d, nn, d, d, 0.0, None, 0, dneg.ptr, dneg.ld, 0, ops.current_stream())
    _sync()
    Y64 = Y.astype(np.float64)
    want = orc.linkpred_fwd_bwd(Y64[:B], Y64[B:2 * B], Y64[2 * B:])
    np.testing.assert_allclose(loss_rows.cpu().numpy().sum(), want["loss"], rtol=1e-4)
    np.testing.assert_allclose(aff.numpy(), want["aff_all"], rtol=1e-4, atol=1e-5)
    got_rank = np.round(1.0 / rr.cpu().numpy() - 1).astype(np.int64)
    margin = np.abs(want["aff_all"][:, :-1] - want["aff_all"][:, -1:]).min(axis=1) > 1e-5   # ignore float near-ties
    assert np.array_equal(got_rank[margin], want["ranks"][margin])
    g = dY.numpy().astype(np.float64) * B
    np.testing.assert_allclose(g[:B], want["d_o1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g[B:2 * B], want["d_o2"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g[2 * B:], want["d_neg"], rtol=1e-4, atol=1e-4)


def placeholders():
    return {'batch1': Placeholder('batch1'), 'batch2': Placeholder('batch2'), 'neg_samples': Placeholder('neg'),
            'dropout': Placeholder('dropout', 0.), 'batch_size': Placeholder('batch_size')}


def build(agg_type="mean", concat=True, csr=False, wd=0.0, dim=32, nn=6, n_nodes=400, lr=0.01):
    eng.reset_engine()
    inits.set_seed(11)
    np.random.seed(7)            # EdgeMinibatchIterator permutes edges with the global NumPy RNG (minibatch.py:42)
    G = synthetic_graph(n_nodes=n_nodes, feat_dim=50, num_classes=5, avg_degree=6, seed=5)
    ph = placeholders()
    rp, col = None, None
    it = EdgeMinibatchIterator(G, None, ph, context_pairs=None, batch_size=32, max_degree=10)
    e = eng.get_engine()
    adj = CSRAdjacency(it.train_csr[0], it.train_csr[1], G.n_nodes, e.device) if csr else PaddedAdjacency(it.adj, e.device)
    adj_info = AdjInfo(adj)
    sampler = UniformNeighborSampler(adj_info)
    od = 2 * dim if agg_type == "gcn" else dim
    ns = [5, 3]
    layer_infos = [SAGEInfo("node", sampler, ns[0], od), SAGEInfo("node", sampler, ns[1], od)]
    model = SampleAndAggregate(ph, G.padded_features(), adj_info, it.deg, layer_infos, concat=concat,
                               aggregator_type=agg_type, learning_rate=lr, weight_decay=wd, neg_sample_size=nn)
    return G, it, ph, sampler, model, ns


def sample_three_calls(adj, b1, b2, neg, ns, perms):
    """models.py:347-357: sample(batch1), sample(batch2), sample(neg_samples) -- each call of models.py:254-275 shuffles its
    own columns; perms[g * K + k] is the permutation of group g's k-th sampler call.  Returns the per-hop id vectors of the
    concatenated roots [batch1 | batch2 | negatives] (rows are independent, so one pass over them is the same computation)."""
    K = len(ns)
    groups = [orc.sample(adj, r, ns, perms[g * K:(g + 1) * K]) for g, r in enumerate((b1, b2, neg))]
    samples = [np.concatenate([grp[0][h] for grp in groups]) for h in range(K + 1)]
    return samples, groups[0][1]


def oracle_agg_params(model, agg_type):
    out = []
    for a in model.aggregators:
        p = {k: v.numpy().copy() for k, v in a.vars.items()}
        if agg_type in ("maxpool", "meanpool"):
            p["mlp_weights"] = a.mlp_layers[0].vars['weights'].numpy().copy()
            p["mlp_bias"] = a.mlp_layers[0].vars['bias'].numpy().reshape(-1).copy()
        out.append(p)
    return out


@pytest.mark.parametrize("agg_type,concat", [("mean", True), ("gcn", False), ("maxpool", True)])
def test_unsup_train_step_matches_oracle(dev, agg_type, concat):
    wd, nn = 0.01, 6
    G, it, ph, sampler, model, ns = build(agg_type, concat, wd=wd, nn=nn)
    model.use_graphs = False
    rng = np.random.RandomState(3)
    edges = it.train_edges[:29]                                  # ragged batch
    B = len(edges)
    perms = [rng.permutation(it.max_degree) for _ in range(3 * len(ns))]     # six tf.random_shuffle per step
    params = oracle_agg_params(model, agg_type)
    sampler.inject_perms(perms)
    feed = {ph['batch1']: edges[:, 0], ph['batch2']: edges[:, 1], ph['batch_size']: B}
    loss, ranks, aff_all, mrr, outputs1 = model.train_step(feed)
    neg = sampler_hash.sample_unigram(sampler_hash.unigram_cdf_u32(it.deg), nn, 123, 0)
    roots = np.concatenate([edges[:, 0], edges[:, 1], neg]).astype(np.int32)
    assert np.array_equal(model.samples1[0].cpu().numpy(), roots)            # same negatives as the oracle hash
    samples, support = sample_three_calls(it.adj, edges[:, 0], edges[:, 1], neg, ns, perms)
    for got, want in zip(model.samples1, samples):
        assert np.array_equal(got.cpu().numpy(), want)
    res = orc.unsupervised_fwd_bwd(params, G.padded_features(), samples, support, model.dims, ns, B, nn, agg_type, concat,
                                   weight_decay=wd)
    np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(outputs1, res["outputs1"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(aff_all, res["aff_all"], rtol=1e-4, atol=1e-4)
    assert abs(mrr - res["mrr"]) < 0.05                           # float near-ties may move a rank by one
    for li, a in enumerate(model.aggregators):
        for k, v in a.vars.items():
            w = res["grads"][li][k]
            np.testing.assert_allclose(v.grad.numpy().reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()),
                                       err_msg="%d/%s" % (li, k))


def test_unsup_dropout_matches_oracle(dev):
    """dropout > 0 in the unsupervised model (unsupervised_train.py:33,128): masks injected into the oracle."""
    from test_model_gpu import _device_masks
    wd, nn, rate = 0.01, 6, 0.25
    G, it, ph, sampler, model, ns = build("mean", True, wd=wd, nn=nn)
    model.use_graphs = False
    rng = np.random.RandomState(5)
    edges = it.train_edges[:21]
    B = len(edges)
    perms = [rng.permutation(it.max_degree) for _ in range(3 * len(ns))]
    params = oracle_agg_params(model, "mean")
    sampler.inject_perms(perms)
    feed = {ph['batch1']: edges[:, 0], ph['batch2']: edges[:, 1], ph['batch_size']: B, ph['dropout']: rate}
    loss, ranks, aff_all, mrr, outputs1 = model.train_step(feed)
    neg = sampler_hash.sample_unigram(sampler_hash.unigram_cdf_u32(it.deg), nn, 123, 0)
    roots = np.concatenate([edges[:, 0], edges[:, 1], neg]).astype(np.int32)
    samples, support = sample_three_calls(it.adj, edges[:, 0], edges[:, 1], neg, ns, perms)
    masks, _ = _device_masks(model, "mean", rate, 0, ns, len(roots))
    res = orc.unsupervised_fwd_bwd(params, G.padded_features(), samples, support, model.dims, ns, B, nn, "mean", True,
                                   weight_decay=wd, masks=masks)
    np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(outputs1, res["outputs1"], rtol=1e-4, atol=1e-4)
    for li, a in enumerate(model.aggregators):
        for k, v in a.vars.items():
            w = res["grads"][li][k]
            np.testing.assert_allclose(v.grad.numpy().reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()),
                                       err_msg="%d/%s" % (li, k))


def test_unsup_device_pipeline_equals_feed_path(dev):
    """Device-resident pairs + horizontal-fusion pipeline + hipGraph replay == host-fed eager steps (bitwise)."""
    outs = []
    for mode in ("feed", "device"):
        G, it, ph, sampler, model, ns = build("mean", True, csr=True)
        pairs = it.train_edges[:96]
        if mode == "feed":
            model.use_graphs = False
            for i in range(3):
                e = pairs[i * 32:(i + 1) * 32]
                model.train_step({ph['batch1']: e[:, 0], ph['batch2']: e[:, 1], ph['batch_size']: 32})
        else:
            model.attach_device_pairs(pairs)
            for i in range(3):
                model.train_step_device(32)
        eng.get_engine().sync()
        outs.append(eng.get_engine().params.cpu().numpy().copy())
    assert np.array_equal(outs[0], outs[1])


def test_unsup_training_improves_mrr(dev):
    G, it, ph, sampler, model, ns = build("mean", True, csr=True, n_nodes=2000, nn=20, lr=0.005)
    rp, col = it.train_csr
    pairs = run_random_walks(rp, col, np.where(~(G.val_mask | G.test_mask))[0], num_walks=10)
    model.attach_device_pairs(np.random.RandomState(0).permutation(pairs))
    B = 128
    first = model.train_step_device(B, fetch=True)[3]
    for _ in range(300):
        model.train_step_device(B)
    last = np.mean([model.train_step_device(B, fetch=True)[3] for _ in range(5)])
    assert last > first + 0.1 and last > 0.4, (first, last)


def test_unsupervised_train_driver(dev, tmp_path, capsys):
    from graphsage_amd import unsupervised_train as ut
    eng.reset_engine()
    ut.main(["--synthetic", "small", "--model", "graphsage_mean", "--epochs", "1", "--batch_size", "128", "--samples_1", "5",
             "--samples_2", "3", "--dim_1", "32", "--dim_2", "32", "--max_total_steps", "60", "--print_every", "20",
             "--validate_iter", "30", "--learning_rate", "0.001", "--max_walk_pairs", "20000", "--base_log_dir", str(tmp_path)])
    out = capsys.readouterr().out
    assert "Optimization Finished!" in out
    assert re.search(r"Iter: \d{4} train_loss= \d+\.\d{5} train_mrr= \d\.\d{5} train_mrr_ema= \d\.\d{5} val_loss= \d+\.\d{5} "
                     r"val_mrr= \d\.\d{5} val_mrr_ema= \d\.\d{5} time= \d+\.\d{5}", out)
    files = [os.path.join(dp, f) for dp, _, fs in os.walk(str(tmp_path)) for f in fs]
    npy = [p for p in files if p.endswith("val.npy")]
    assert npy and any(p.endswith("val.txt") for p in files)
    emb = np.load(npy[0])
    assert emb.shape == (3000, 64) and np.allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-4)


@pytest.mark.parametrize("B,d,nn", [(512, 256, 20), (37, 64, 5), (130, 128, 20), (9, 512, 3)])
def test_linkpred_norm_fused_vs_oracle(dev, B, d, nn):
    """gs_linkpred_norm_fwd_bwd (l2_normalize + xent link prediction + MRR + the gradient carried back through the
    normalisation, one launch + the negatives' rows) vs the oracle's l2_normalize_fwd / linkpred_fwd_bwd / l2_normalize_bwd
    (models.py:368-405, prediction.py:68-110) in fp64; includes a clamped (all-zero) row in every group."""
    rng = np.random.default_rng(B + d + 1)
    Z = (rng.normal(size=(2 * B + nn, d)) * rng.uniform(0.2, 3.0, size=(2 * B + nn, 1))).astype(np.float32)
    Z[:B] = 0.7 * Z[:B] + 0.3 * Z[B:2 * B]
    Z[1] = 0
    Z[B + 2] = 0
    Z[2 * B + 1] = 0                                                  # clamped rows (sum z^2 < 1e-12)
    Zd = Mat.from_numpy(Z, dev)
    Y, dZ = Mat.zeros(2 * B + nn, d, dev), Mat.zeros(2 * B + nn, d, dev)
    loss_rows, rr = torch.zeros(B, device=dev), torch.zeros(B, device=dev)
    aff = Mat.zeros(B, nn + 1, dev)
    slabs = torch.zeros(((B + 3) // 4) * nn * d, device=dev)
    ops.call("gs_linkpred_norm_fwd_bwd", Zd.ptr, Zd.ld, B, d, nn, 1.0, 1.0 / B, Y.ptr, Y.ld, ops.ptr(loss_rows), ops.ptr(rr),
             aff.ptr, aff.ld, dZ.ptr, dZ.ld, ops.ptr(slabs), ops.current_stream())
    _sync()
    Z64 = Z.astype(np.float64)
    y, cache = orc.l2_normalize_fwd(Z64)
    want = orc.linkpred_fwd_bwd(y[:B], y[B:2 * B], y[2 * B:])
    np.testing.assert_allclose(Y.numpy(), y, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(loss_rows.cpu().numpy().sum(), want["loss"], rtol=1e-4)
    np.testing.assert_allclose(aff.numpy(), want["aff_all"], rtol=1e-4, atol=1e-5)
    got_rank = np.round(1.0 / rr.cpu().numpy() - 1).astype(np.int64)
    margin = np.abs(want["aff_all"][:, :-1] - want["aff_all"][:, -1:]).min(axis=1) > 1e-5
    assert np.array_equal(got_rank[margin], want["ranks"][margin])
    d_y = np.concatenate([want["d_o1"], want["d_o2"], want["d_neg"]], axis=0) / B
    d_z = orc.l2_normalize_bwd(d_y, cache)
    got = dZ.numpy().astype(np.float64)
    live = np.ones(2 * B + nn, bool)
    live[[1, B + 2, 2 * B + 1]] = False                               # clamped rows: 1e6-scaled, compared relatively below
    scale = np.abs(d_z[live]).max()
    np.testing.assert_allclose(got[live], d_z[live], rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(got[~live], d_z[~live], rtol=1e-3, atol=1e-4 * max(1.0, np.abs(d_z[~live]).max()))


def test_fanout_unsup_staging_bit_exact(dev):
    """The fan-out sampler staging its own roots (edge-pair batch + unigram negatives, gs_sample_fanout_desc) == the
    separate gs_unsup_stage launch == the oracle hash, with and without the guide table; then the hops as usual."""
    rng = np.random.default_rng(4)
    N, B, nn, fans = 6000, 300, 20, [10, 25]
    deg = rng.integers(0, 120, size=N)
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, N, size=int(rowptr[-1])).astype(np.int32)
    cdf = sampler_hash.unigram_cdf_u32(deg)
    bits = 12
    thr = (np.arange((1 << bits) + 1, dtype=np.uint64) << np.uint64(32 - bits))
    guide = np.minimum(np.searchsorted(cdf.astype(np.uint64), thr, side="right"), N - 1).astype(np.int32)
    pairs = rng.integers(0, N, size=(1000, 2)).astype(np.int32)
    n_roots = 2 * B + nn
    sizes = [n_roots, n_roots * fans[0], n_roots * fans[0] * fans[1]]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    rp, cl = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
    cdf_dev = torch.from_numpy(cdf.view(np.int32).copy()).to(dev)
    pairs_dev = torch.from_numpy(pairs).to(dev)
    cur = torch.tensor([900], dtype=torch.int64, device=dev)
    clk = torch.tensor([7], dtype=torch.int64, device=dev)
    outs = []
    for g in (None, torch.from_numpy(guide).to(dev)):
        ids_all = torch.full((int(offs[-1]),), -3, dtype=torch.int32, device=dev)
        desc = ops.fanout_desc(rp, cl, N, N, fans, offs.tolist(), ids_all, n_roots, 123, step_dev=clk, cursor_dev=cur,
                               unsup=(pairs_dev, B, cdf_dev, g, bits, nn, 123))
        ops.sample_fanout_desc(desc)
        _sync()
        outs.append(ids_all.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    got = outs[0]
    sel = pairs[(900 + np.arange(B)) % 1000]
    roots = np.concatenate([sel[:, 0], sel[:, 1], sampler_hash.sample_unigram(cdf, nn, 123, 7)]).astype(np.int32)
    assert np.array_equal(got[:n_roots], roots)
    staged = torch.full((n_roots,), -1, dtype=torch.int32, device=dev)
    ops.call("gs_unsup_stage", ops.ptr(pairs_dev), 1000, ops.ptr(cur), B, ops.ptr(cdf_dev), N, nn, 123, ops.ptr(clk), 0,
             ops.ptr(staged), ops.current_stream())
    _sync()
    assert np.array_equal(staged.cpu().numpy(), roots)
    hop1 = sampler_hash.sample_uniform_csr(rowptr, col, N, N, roots, fans[0], 123, 7, 0)
    hop2 = sampler_hash.sample_uniform_csr(rowptr, col, N, N, hop1.reshape(-1), fans[1], 123, 7, 1)
    assert np.array_equal(got[offs[1]:offs[2]], hop1.reshape(-1)) and np.array_equal(got[offs[2]:offs[3]], hop2.reshape(-1))
    # law "reference": batch1 | batch2 | negatives are three sample() calls of the reference (models.py:347-357), each
    # with its own column permutation per hop -> segment call ids g * K + hop; virtual and materialised table agree
    M = 32
    table = ops.build_padded_table(rp, cl, N, N, M, 123)
    hop1 = sampler_hash.sample_uniform_csr_segments(rowptr, col, N, N, roots, fans[0], 123, 7, 0, 2, (B, 2 * B), law=1,
                                                    max_degree=M)
    hop2 = sampler_hash.sample_uniform_csr_segments(rowptr, col, N, N, hop1.reshape(-1), fans[1], 123, 7, 1, 2,
                                                    (B * fans[0], 2 * B * fans[0]), law=1, max_degree=M)
    shared = sampler_hash.sample_uniform_csr(rowptr, col, N, N, roots, fans[0], 123, 7, 0, law=1, max_degree=M)
    assert not np.array_equal(shared, hop1) and np.array_equal(shared[:B], hop1[:B])   # ONE shared permutation is not it
    for tbl in (None, table):
        ids_all = torch.full((int(offs[-1]),), -3, dtype=torch.int32, device=dev)
        desc = ops.fanout_desc(rp, cl, N, N, fans, offs.tolist(), ids_all, n_roots, 123, step_dev=clk, cursor_dev=cur,
                               unsup=(pairs_dev, B, cdf_dev, None, bits, nn, 123), law=1, max_degree=M, padded_table=tbl)
        ops.sample_fanout_desc(desc)
        _sync()
        got = ids_all.cpu().numpy()
        assert np.array_equal(got[:n_roots], roots)
        assert np.array_equal(got[offs[1]:offs[2]], hop1.reshape(-1)) and np.array_equal(got[offs[2]:offs[3]], hop2.reshape(-1))
    # roots already in the buffer (host-fed batches): explicit seg_begin, same ids
    ids_all = torch.full((int(offs[-1]),), -3, dtype=torch.int32, device=dev)
    ids_all[:n_roots] = torch.from_numpy(roots).to(dev)
    desc = ops.fanout_desc(rp, cl, N, N, fans, offs.tolist(), ids_all, n_roots, 123, step_dev=clk, law=1, max_degree=M,
                           padded_table=table, segments=(B, 2 * B))
    ops.sample_fanout_desc(desc)
    _sync()
    got = ids_all.cpu().numpy()
    assert np.array_equal(got[offs[1]:offs[2]], hop1.reshape(-1)) and np.array_equal(got[offs[2]:offs[3]], hop2.reshape(-1))
    # data-parallel: rank r's negatives are keyed by its global root offset -> different ranks, different negatives
    ids_all = torch.full((int(offs[-1]),), -3, dtype=torch.int32, device=dev)
    desc = ops.fanout_desc(rp, cl, N, N, fans, offs.tolist(), ids_all, n_roots, 123, step_dev=clk, cursor_dev=cur,
                           unsup=(pairs_dev, B, cdf_dev, None, bits, nn, 123), root_offset=n_roots)
    ops.sample_fanout_desc(desc)
    _sync()
    neg1 = ids_all.cpu().numpy()[2 * B:n_roots]
    assert np.array_equal(neg1, sampler_hash.sample_unigram(cdf, nn, 123, 7, slot_offset=n_roots))
    assert not np.array_equal(neg1, roots[2 * B:])
    # ... and the stand-alone staging (host-fed / unpipelined steps, eval) keys its negatives the same way
    staged = torch.full((n_roots,), -1, dtype=torch.int32, device=dev)
    ops.call("gs_unsup_stage", None, 0, None, B, ops.ptr(cdf_dev), N, nn, 123, ops.ptr(clk), n_roots, ops.ptr(staged),
             ops.current_stream())
    _sync()
    assert np.array_equal(staged.cpu().numpy()[2 * B:], neg1)


@pytest.mark.parametrize("dim,nn,B", [(64, 6, 29), (128, 20, 64), (64, 32, 8), (128, 5, 3)])
def test_unsup_fused_tail_matches_oracle_and_the_unfused_schedule(dev, dim, nn, B):
    """The fused layer-1 + link-prediction launches (gs_linkpred_tail / gs_linkpred_tail_neg; two-layer mean model at widths
    128 / 256) vs the NumPy oracle -- loss, embeddings, affinities, MRR, every gradient, ragged pair groups (B not a multiple
    of 8), 5 .. 32 negatives -- and vs the per-operator schedule (fuse_tail = False) of the same step; evaluation too."""
    wd = 0.01
    res_dev = []
    for fuse in (True, False):
        G, it, ph, sampler, model, ns = build("mean", True, wd=wd, nn=nn, dim=dim)
        model.use_graphs = False
        model.fuse_tail = fuse
        rng = np.random.RandomState(3)
        edges = it.train_edges[:B]
        perms = [rng.permutation(it.max_degree) for _ in range(3 * len(ns))]
        params = oracle_agg_params(model, "mean")
        sampler.inject_perms(perms)
        feed = {ph['batch1']: edges[:, 0], ph['batch2']: edges[:, 1], ph['batch_size']: B}
        loss, ranks, aff_all, mrr, outputs1 = model.train_step(feed)
        assert bool(getattr(model, "_lp_tail_used", False)) == fuse
        grads = [{k: v.grad.numpy().copy() for k, v in a.vars.items()} for a in model.aggregators]
        full = model.outputs_all.numpy().copy()
        got_samples = [t.cpu().numpy().copy() for t in model.samples1]
        sampler.inject_perms(perms)
        eloss, eranks, emrr, eouts = model.eval_step(feed)
        res_dev.append((loss, aff_all, mrr, outputs1, grads, full, eloss, emrr, eouts))
        if fuse:
            neg = sampler_hash.sample_unigram(sampler_hash.unigram_cdf_u32(it.deg), nn, 123, 0)
            samples, support = sample_three_calls(it.adj, edges[:, 0], edges[:, 1], neg, ns, perms)
            for got, want in zip(got_samples, samples):
                assert np.array_equal(got, want)
            res = orc.unsupervised_fwd_bwd(params, G.padded_features(), samples, support, model.dims, ns, B, nn, "mean", True,
                                           weight_decay=wd)
            np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(outputs1, res["outputs1"], rtol=1e-4, atol=1e-4)
            np.testing.assert_allclose(aff_all, res["aff_all"], rtol=1e-4, atol=1e-4)
            assert abs(mrr - res["mrr"]) < 0.05
            for li, g in enumerate(grads):
                for k, got in g.items():
                    w = res["grads"][li][k]
                    np.testing.assert_allclose(got.reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()),
                                               err_msg="%d/%s" % (li, k))
    a, b = res_dev
    np.testing.assert_allclose(a[0], b[0], rtol=1e-5)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a[5], b[5], rtol=1e-5, atol=1e-6)            # every normalised embedding, negatives included
    for ga, gb in zip(a[4], b[4]):
        for k in ga:
            np.testing.assert_allclose(ga[k], gb[k], rtol=1e-4, atol=1e-5 * max(1e-2, np.abs(gb[k]).max()), err_msg=k)
    np.testing.assert_allclose(a[6], b[6], rtol=1e-5)                        # evaluation: loss, mrr, embeddings
    assert abs(a[7] - b[7]) < 0.05
    np.testing.assert_allclose(a[8], b[8], rtol=1e-5, atol=1e-6)


def test_unsup_fused_tail_pipeline_equals_feed_path(dev):
    """At widths the fused tail takes: device-resident pairs + rider pipeline + multi-step hipGraph replay == host-fed eager
    steps, bit for bit (the hand-over counters survive replays; 12 steps on one buffer)."""
    outs = []
    for mode in ("feed", "device", "device_multi"):
        G, it, ph, sampler, model, ns = build("mean", True, csr=True, dim=64, nn=20, n_nodes=1500)
        pairs = it.train_edges[:12 * 32]
        if mode == "feed":
            model.use_graphs = False
            for i in range(12):
                e = pairs[i * 32:(i + 1) * 32]
                model.train_step({ph['batch1']: e[:, 0], ph['batch2']: e[:, 1], ph['batch_size']: 32})
        else:
            model.attach_device_pairs(pairs)
            if mode == "device":
                for i in range(12):
                    model.train_step_device(32)
            else:
                model.train_steps_device(32, 12, steps_per_launch=4)
        assert model._lp_tail_used
        loss = model._fetch_unsup(32)[0]                                # reads the hand-over error word
        outs.append((loss, eng.get_engine().params.cpu().numpy().copy()))
    assert outs[0][0] == outs[1][0] == outs[2][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][1], outs[2][1])
