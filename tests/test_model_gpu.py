"""-m gpu end-to-end parity: SupervisedGraphsage (HIP kernels through the C ABI) vs the NumPy oracle on
IDENTICAL sampled neighbor sets (north_star: within 1e-4 fp32), for every aggregator of SURVEY §8a."""
import numpy as np
import pytest

from graphsage_amd import engine as eng
from graphsage_amd import inits
from graphsage_amd.minibatch import NodeMinibatchIterator
from graphsage_amd.models import Placeholder, SAGEInfo
from graphsage_amd.neigh_samplers import AdjInfo, CSRAdjacency, PaddedAdjacency, UniformNeighborSampler
from graphsage_amd.supervised_models import SupervisedGraphsage
from graphsage_amd.utils import synthetic_graph
from oracle import graphsage_oracle as orc
from oracle import sampler_hash

pytestmark = pytest.mark.gpu


def placeholders():
    return {'labels': Placeholder('labels'), 'batch': Placeholder('batch1'),
            'dropout': Placeholder('dropout', 0.), 'batch_size': Placeholder('batch_size')}


def oracle_params(model, agg_type):
    agg = []
    for a in model.aggregators:
        p = {k: v.numpy().copy() for k, v in a.vars.items()}
        if agg_type in ("maxpool", "meanpool"):
            p["mlp_weights"] = a.mlp_layers[0].vars['weights'].numpy().copy()
            p["mlp_bias"] = a.mlp_layers[0].vars['bias'].numpy().reshape(-1).copy()
        agg.append(p)
    return {"agg": agg, "node_pred": {"weights": model.node_pred.vars['weights'].numpy().copy(),
                                      "bias": model.node_pred.vars['bias'].numpy().reshape(-1).copy()}}


def device_grads(model, agg_type):
    agg = []
    for a in model.aggregators:
        g = {k: v.grad.numpy().copy() for k, v in a.vars.items()}
        if agg_type in ("maxpool", "meanpool"):
            g["mlp_weights"] = a.mlp_layers[0].vars['weights'].grad.numpy().copy()
            g["mlp_bias"] = a.mlp_layers[0].vars['bias'].grad.numpy().reshape(-1).copy()
        agg.append(g)
    return {"agg": agg, "node_pred": {"weights": model.node_pred.vars['weights'].grad.numpy().copy(),
                                      "bias": model.node_pred.vars['bias'].grad.numpy().reshape(-1).copy()}}


def build(dev, agg_type, concat, sigmoid, K=2, csr=False, wd=0.0, feat_dim=50, dim=32, max_degree=10, n_nodes=400,
          fuse=True, identity_dim=0, use_features=True):
    eng.reset_engine()
    inits.set_seed(7)
    G = synthetic_graph(n_nodes=n_nodes, feat_dim=feat_dim, num_classes=7, avg_degree=6, seed=5, multilabel=sigmoid)
    ph = placeholders()
    it = NodeMinibatchIterator(G, None, ph, None, G.num_classes, batch_size=32, max_degree=max_degree)
    e = eng.get_engine()
    if csr:
        adj_train = CSRAdjacency(it.train_csr[0], it.train_csr[1], G.n_nodes, e.device)
    else:
        adj_train = PaddedAdjacency(it.adj, e.device)
    adj_info = AdjInfo(adj_train)
    sampler = UniformNeighborSampler(adj_info)
    ns = [5, 3, 2][:K]
    od = (2 * dim if agg_type == "gcn" else dim)
    layer_infos = [SAGEInfo("node", sampler, ns[i], od) for i in range(K)]
    model = SupervisedGraphsage(G.num_classes, ph, G.padded_features() if use_features else None, adj_info, it.deg,
                                layer_infos, concat=concat, aggregator_type=agg_type, sigmoid_loss=sigmoid,
                                learning_rate=0.01, weight_decay=wd, identity_dim=identity_dim)
    model.fuse_head = model.fuse_sampler = model.fuse_tail = fuse
    return G, it, ph, sampler, model, ns


CASES = [("mean", True, False), ("mean", False, True), ("gcn", False, False), ("maxpool", True, False),
         ("meanpool", True, True), ("mean", True, True)]


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("agg_type,concat,sigmoid", CASES)
def test_train_step_matches_oracle(dev, agg_type, concat, sigmoid, fuse):
    wd = 0.01
    G, it, ph, sampler, model, ns = build(dev, agg_type, concat, sigmoid, wd=wd, fuse=fuse)
    model.use_graphs = False
    rng = np.random.RandomState(3)
    batch = rng.choice(it.train_nodes, size=37, replace=False).astype(np.int32)   # ragged batch
    batch[0] = G.n_nodes - 1 if (G.val_mask | G.test_mask)[G.n_nodes - 1] else batch[0]
    perms = [rng.permutation(it.max_degree) for _ in ns]
    labels = it.label_matrix[batch]
    params = oracle_params(model, agg_type)
    sampler.inject_perms(perms)
    feed = {ph['batch']: batch, ph['labels']: labels, ph['batch_size']: len(batch)}
    loss, preds = model.train_step(feed)
    # ---- oracle on the identical neighbor sets
    samples, support = orc.sample(it.adj, batch, ns, perms)
    for got, want in zip(model.samples1, samples):
        assert np.array_equal(got.cpu().numpy(), want)          # S1/S2: bit exact
    feats = G.padded_features()
    res = orc.supervised_fwd_bwd(params, feats, samples, support, labels, model.dims, ns, len(batch), agg_type,
                                 concat, sigmoid, weight_decay=wd)
    np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(preds, res["preds"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(model.outputs1.numpy(), res["outputs1"], rtol=1e-4, atol=1e-4)
    got = device_grads(model, agg_type)
    for (name, g), (_, w) in zip(orc.flat_param_items(got, agg_type), orc.flat_param_items(res["grads"], agg_type)):
        np.testing.assert_allclose(g.reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()),
                                   err_msg=name)
    # ---- optimizer: clip +-5 and TF Adam, t = 1
    after = oracle_params(model, agg_type)
    for (name, p0), (_, g), (_, p1) in zip(orc.flat_param_items(params, agg_type),
                                           orc.flat_param_items(res["grads"], agg_type),
                                           orc.flat_param_items(after, agg_type)):
        p = p0.copy()
        orc.adam_tf_update(p, orc.clip_by_value(g).reshape(p.shape), np.zeros_like(p), np.zeros_like(p), 1, 0.01)
        np.testing.assert_allclose(p1.reshape(p.shape), p, rtol=1e-4, atol=2e-5, err_msg=name)


@pytest.mark.parametrize("agg_type,concat,sigmoid,idim,use_features",
                         [("mean", True, False, 6, True), ("gcn", False, False, 8, True), ("maxpool", True, True, 5, True),
                          ("meanpool", True, False, 4, True), ("mean", True, False, 12, False)])
def test_identity_features_match_oracle(dev, agg_type, concat, sigmoid, idim, use_features):
    """identity_dim > 0 (supervised_models.py:49-60): trainable node_embeddings concatenated in front of the features;
    gradient = per-id sum of the layer-0 input gradients, dense clip + Adam, no weight decay.  Two steps: the second
    checks that the gathered table was refreshed and the gradient accumulator cleared."""
    wd = 0.01
    G, it, ph, sampler, model, ns = build(dev, agg_type, concat, sigmoid, wd=wd, identity_dim=idim,
                                          use_features=use_features)
    model.use_graphs = False
    assert model.dims[0] == idim + (G.feats.shape[1] if use_features else 0)
    rng = np.random.RandomState(4)
    batch = rng.choice(it.train_nodes, size=29, replace=False).astype(np.int32)
    labels = it.label_matrix[batch]
    feed = {ph['batch']: batch, ph['labels']: labels, ph['batch_size']: len(batch)}
    emb_m = emb_v = None
    for step in (1, 2):
        perms = [rng.permutation(it.max_degree) for _ in ns]
        sampler.inject_perms(perms)
        params = oracle_params(model, agg_type)
        emb0 = model.embeds.numpy().copy()
        assert emb0.shape == (G.n_nodes + 1, idim)
        np.testing.assert_array_equal(model.features.numpy()[:, :idim], emb0)   # materialised concat is current
        feats = np.concatenate([emb0, G.padded_features()], axis=1) if use_features else emb0
        loss, preds = model.train_step(feed)
        samples, support = orc.sample(it.adj, batch, ns, perms)
        res = orc.supervised_fwd_bwd(params, feats, samples, support, labels, model.dims, ns, len(batch), agg_type,
                                     concat, sigmoid, weight_decay=wd, identity_dim=idim)
        np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(preds, res["preds"], rtol=1e-4, atol=1e-4)
        got = device_grads(model, agg_type)
        for (name, g), (_, w) in zip(orc.flat_param_items(got, agg_type), orc.flat_param_items(res["grads"], agg_type)):
            np.testing.assert_allclose(g.reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()),
                                       err_msg=name)
        w = res["grads"]["embeds"]
        assert np.count_nonzero(w) > 0
        np.testing.assert_allclose(model.embeds.grad.numpy(), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()))
        if emb_m is None:
            emb_m, emb_v = np.zeros_like(emb0), np.zeros_like(emb0)
        want = emb0.copy()
        orc.adam_tf_update(want, orc.clip_by_value(w), emb_m, emb_v, step, 0.01)     # dense Adam: untouched rows move too
        np.testing.assert_allclose(model.embeds.numpy(), want, rtol=1e-4, atol=2e-5)
    assert float(model.embeds.slabs.abs().max().item()) == 0.0                       # accumulator consumed


def test_identity_features_maxpool_distinct_id_mlp_reads_the_fresh_table(dev):
    """Round 5's review: the default pooling MLP (two fp16 pieces per operand) reads a copy of the feature table cut ONCE
    (Engine.table16_of); with identity_dim > 0 the table's leading columns are the trainable node_embeddings, rewritten behind
    every optimizer launch, so from the second step on that copy is stale.  Such a table must take the kernel that cuts the
    rows it reads (three bf16 pieces).  The distinct-id path is forced (dedup_min_rows = 0; the other identity tests gather
    < 2048 rows and never reach it); three steps against the oracle on the CURRENT embeddings."""
    wd, idim = 0.01, 5
    G, it, ph, sampler, model, ns = build(dev, "maxpool", True, True, wd=wd, identity_dim=idim)
    model.use_graphs = False
    e = eng.get_engine()
    assert e.pool_f16 and e.split_pool and not e.is_constant_table(model.features)
    with pytest.raises(Exception):
        e.table16_of(model.features)
    model.aggregators[0].dedup_min_rows = 0
    rng = np.random.RandomState(4)
    batch = rng.choice(it.train_nodes, size=29, replace=False).astype(np.int32)
    labels = it.label_matrix[batch]
    feed = {ph['batch']: batch, ph['labels']: labels, ph['batch_size']: len(batch)}
    for step in (1, 2, 3):
        perms = [rng.permutation(it.max_degree) for _ in ns]
        sampler.inject_perms(perms)
        params = oracle_params(model, "maxpool")
        emb0 = model.embeds.numpy().copy()
        feats = np.concatenate([emb0, G.padded_features()], axis=1)
        loss, preds = model.train_step(feed)
        assert model.aggregators[0].last_pool_kernel == "split_bf16x3", model.aggregators[0].last_pool_kernel
        cnt, rows_total = model.aggregators[0].last_unique
        assert 0 < int(cnt.item()) <= rows_total
        samples, support = orc.sample(it.adj, batch, ns, perms)
        res = orc.supervised_fwd_bwd(params, feats, samples, support, labels, model.dims, ns, len(batch), "maxpool",
                                     True, True, weight_decay=wd, identity_dim=idim)
        np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(preds, res["preds"], rtol=1e-4, atol=1e-4)
        got = device_grads(model, "maxpool")
        for (name, g), (_, w) in zip(orc.flat_param_items(got, "maxpool"), orc.flat_param_items(res["grads"], "maxpool")):
            np.testing.assert_allclose(g.reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()),
                                       err_msg="step %d %s" % (step, name))


def test_identity_features_graph_replay(dev):
    """The scatter + refresh launches are part of the captured step: replayed steps == eager steps (up to the
    summation order of the fp32 atomics)."""
    outs = []
    for use_graphs in (False, True):
        G, it, ph, sampler, model, ns = build(dev, "mean", True, False, csr=True, identity_dim=8)
        model.use_graphs = use_graphs
        batch = np.random.RandomState(1).choice(it.train_nodes, size=32, replace=False).astype(np.int32)
        feed = {ph['batch']: batch, ph['labels']: it.label_matrix[batch], ph['batch_size']: 32}
        losses = [model.train_step(feed)[0] for _ in range(4)]
        outs.append((losses, model.embeds.numpy().copy(), model.features.numpy()[:, :8].copy()))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(outs[1][1], outs[1][2])


def _device_masks(model, agg_type, rate, clock, ns, n_batch):
    """The masks the device draws this step (site / global-row conventions of aggregators.py + layers.py), computed
    with the CPU restatement of the counter hash, for injection into the oracle."""
    from graphsage_amd.layers import SITE_DENSE, SITE_MLP, SITE_NEIGH, SITE_SELF
    seed = model.engine.dropout_seed
    K = len(ns)
    support = [1]
    for k in range(K):
        support.append(support[-1] * ns[K - k - 1])
    pooling = agg_type in ("maxpool", "meanpool")

    def masks(layer, hop, role, n_rows, d):
        agg = model.aggregators[layer]
        if role == "self":
            if pooling:
                return None
            row0 = sum(n_batch * support[h] for h in range(hop))
            site = agg.site + SITE_SELF
        else:
            row0 = sum(n_batch * support[h + 1] for h in range(hop))
            site = agg.site + (SITE_MLP if pooling else SITE_NEIGH)
        return sampler_hash.dropout_mask(seed, clock, site, row0, n_rows, d, rate)

    def head_mask(n, d):
        return sampler_hash.dropout_mask(seed, clock, model.node_pred.site + SITE_DENSE, 0, n, d, rate)
    return masks, head_mask


@pytest.mark.parametrize("agg_type,concat,sigmoid,K", [("mean", True, False, 2), ("mean", False, True, 3), ("gcn", False, False, 2),
                                                       ("maxpool", True, True, 2), ("meanpool", True, False, 2)])
def test_dropout_matches_oracle(dev, agg_type, concat, sigmoid, K):
    """dropout > 0 (supervised_train.py:38,117): tf.nn.dropout on the aggregator inputs / the pooling MLP input / the
    prediction layer input.  The device masks are a counter hash; the same masks are injected into the oracle, so
    forward, loss and every gradient must agree as in the dropout-free case.  Step 2 uses the next clock value."""
    wd, rate = 0.01, 0.3
    G, it, ph, sampler, model, ns = build(dev, agg_type, concat, sigmoid, wd=wd, K=K)
    model.use_graphs = False
    rng = np.random.RandomState(6)
    batch = rng.choice(it.train_nodes, size=27, replace=False).astype(np.int32)
    labels = it.label_matrix[batch]
    feed = {ph['batch']: batch, ph['labels']: labels, ph['batch_size']: len(batch), ph['dropout']: rate}
    feats = G.padded_features()
    losses = []
    for step in range(2):
        perms = [rng.permutation(it.max_degree) for _ in ns]
        sampler.inject_perms(perms)
        params = oracle_params(model, agg_type)
        clock = int(model.engine.sample_clock_dev.item())
        assert clock == step
        loss, preds = model.train_step(feed)
        samples, support = orc.sample(it.adj, batch, ns, perms)
        masks, head_mask = _device_masks(model, agg_type, rate, clock, ns, len(batch))
        res = orc.supervised_fwd_bwd(params, feats, samples, support, labels, model.dims, ns, len(batch), agg_type,
                                     concat, sigmoid, weight_decay=wd, masks=masks,
                                     head_mask=head_mask(len(batch), model.agg_out.d))
        np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(preds, res["preds"], rtol=1e-4, atol=1e-4)
        got = device_grads(model, agg_type)
        for (name, g), (_, w) in zip(orc.flat_param_items(got, agg_type), orc.flat_param_items(res["grads"], agg_type)):
            np.testing.assert_allclose(g.reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()),
                                       err_msg="step %d %s" % (step, name))
        losses.append(loss)
    # a validation feed (no dropout key -> 0, minibatch.py:269) runs the dropout-free forward
    perms = [rng.permutation(it.max_degree) for _ in ns]
    sampler.inject_perms(perms)
    params = oracle_params(model, agg_type)
    loss, preds = model.eval_step({ph['batch']: batch, ph['labels']: labels, ph['batch_size']: len(batch)})
    samples, support = orc.sample(it.adj, batch, ns, perms)
    res = orc.supervised_fwd_bwd(params, feats, samples, support, labels, model.dims, ns, len(batch), agg_type, concat,
                                 sigmoid, weight_decay=wd, want_grads=False)
    np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("agg_type,concat", [("mean", True), ("gcn", False), ("maxpool", True)])
def test_identity_with_dropout_matches_oracle(dev, agg_type, concat):
    """identity_dim > 0 AND dropout > 0: the embedding gradient goes through the per-sampled-row dropout masks."""
    wd, rate, idim = 0.0, 0.4, 6
    G, it, ph, sampler, model, ns = build(dev, agg_type, concat, False, wd=wd, identity_dim=idim)
    model.use_graphs = False
    rng = np.random.RandomState(8)
    batch = rng.choice(it.train_nodes, size=25, replace=False).astype(np.int32)
    labels = it.label_matrix[batch]
    perms = [rng.permutation(it.max_degree) for _ in ns]
    sampler.inject_perms(perms)
    params = oracle_params(model, agg_type)
    emb0 = model.embeds.numpy().copy()
    feats = np.concatenate([emb0, G.padded_features()], axis=1)
    loss, preds = model.train_step({ph['batch']: batch, ph['labels']: labels, ph['batch_size']: len(batch),
                                    ph['dropout']: rate})
    samples, support = orc.sample(it.adj, batch, ns, perms)
    masks, head_mask = _device_masks(model, agg_type, rate, 0, ns, len(batch))
    res = orc.supervised_fwd_bwd(params, feats, samples, support, labels, model.dims, ns, len(batch), agg_type, concat,
                                 False, weight_decay=wd, identity_dim=idim, masks=masks,
                                 head_mask=head_mask(len(batch), model.agg_out.d))
    np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5)
    w = res["grads"]["embeds"]
    assert np.count_nonzero(w) > 0
    np.testing.assert_allclose(model.embeds.grad.numpy(), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()))
    got = device_grads(model, agg_type)
    for (name, g), (_, w) in zip(orc.flat_param_items(got, agg_type), orc.flat_param_items(res["grads"], agg_type)):
        np.testing.assert_allclose(g.reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()), err_msg=name)


def test_dropout_graph_replay_and_device_epoch(dev):
    """Masks are keyed by the DEVICE step clock, so replayed hipGraphs draw new masks every step and equal the eager
    run; the device-epoch path falls back to the sequential schedule under dropout."""
    outs = []
    for use_graphs in (False, True):
        G, it, ph, sampler, model, ns = build(dev, "mean", True, False, csr=True)
        model.use_graphs = use_graphs
        batch = np.random.RandomState(1).choice(it.train_nodes, size=32, replace=False).astype(np.int32)
        feed = {ph['batch']: batch, ph['labels']: it.label_matrix[batch], ph['batch_size']: 32, ph['dropout']: 0.5}
        losses = [model.train_step(feed)[0] for _ in range(5)]
        outs.append((losses, model.aggregators[0].vars['self_weights'].numpy().copy()))
    assert len(set(np.round(outs[0][0], 6))) > 1
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-6)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-7)
    # device-resident epoch with the dropout placeholder set: sequential schedule, still trains
    order = np.random.RandomState(2).permutation(it.train_nodes).astype(np.int32)
    model.attach_device_epoch(order, np.vstack([it.label_matrix, np.zeros((1, G.num_classes), np.float32)]))
    ph['dropout'].value = 0.5
    model.train_steps_device(32, 4)
    assert np.isfinite(model._fetch(32)[0])


def test_three_layer_mean(dev):
    """samples_3 != 0 (supervised_train.py:153-156): a hidden tensor is both a self and a neighbor input."""
    G, it, ph, sampler, model, ns = build(dev, "mean", True, False, K=3)
    model.use_graphs = False
    rng = np.random.RandomState(4)
    batch = rng.choice(it.train_nodes, size=16, replace=False).astype(np.int32)
    perms = [rng.permutation(it.max_degree) for _ in ns]
    labels = it.label_matrix[batch]
    params = oracle_params(model, "mean")
    sampler.inject_perms(perms)
    loss, preds = model.train_step({ph['batch']: batch, ph['labels']: labels, ph['batch_size']: len(batch)})
    samples, support = orc.sample(it.adj, batch, ns, perms)
    res = orc.supervised_fwd_bwd(params, G.padded_features(), samples, support, labels, model.dims, ns, len(batch),
                                 "mean", True, False)
    np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5)
    got = device_grads(model, "mean")
    for (name, g), (_, w) in zip(orc.flat_param_items(got, "mean"), orc.flat_param_items(res["grads"], "mean")):
        np.testing.assert_allclose(g.reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()), err_msg=name)


def test_eval_swaps_adjacency(dev):
    """tf.assign(adj_info, test_adj) semantics: the sampler follows the handle (supervised_train.py:280,285)."""
    G, it, ph, sampler, model, ns = build(dev, "mean", True, False)
    model.use_graphs = False
    e = eng.get_engine()
    val = it.val_nodes[:20].astype(np.int32)
    labels = it.label_matrix[val]
    feed = {ph['batch']: val, ph['labels']: labels, ph['batch_size']: len(val)}
    perms = [np.arange(it.max_degree), np.arange(it.max_degree)]
    sampler.inject_perms(perms)
    model.eval_step(feed)
    # under the TRAIN adjacency every val node row is all-pad (minibatch.py:232-233)
    assert (model.samples1[1].cpu().numpy() == G.n_nodes).all()
    model.adj_info.assign(PaddedAdjacency(it.test_adj, e.device))
    sampler.inject_perms(perms)
    loss, preds = model.eval_step(feed)
    want, _ = orc.sample(it.test_adj, val, ns, perms)
    assert np.array_equal(model.samples1[2].cpu().numpy(), want[2])
    assert np.isfinite(loss) and preds.shape == (20, G.num_classes)


def test_csr_training_graph_replay_equals_eager(dev):
    """CSR sampler + hipGraph replay: three steps replayed == three steps eager (deterministic kernels)."""
    outs = []
    # (hipGraph?, fused kernels?, cross-step pipeline?)
    for use_graphs, fuse, pipe in ((False, True, False), (True, True, True), (False, False, False), (True, True, False),
                                   (False, True, True)):
        G, it, ph, sampler, model, ns = build(dev, "mean", True, False, csr=True, fuse=fuse)
        model.use_graphs = use_graphs
        model.pipeline = pipe
        order = it.train_nodes[:96]
        model.attach_device_epoch(order, it.label_matrix)
        losses = [model.train_step_device(32, fetch=True)[0] for _ in range(3)]
        # the sampled ids must be what the oracle hash predicts for (seed, step=2, hop)
        want1 = sampler_hash.sample_uniform_csr(it.train_csr[0], it.train_csr[1], G.n_nodes, G.n_nodes,
                                                order[64:96], ns[1], 123, 2, 0)
        assert np.array_equal(model.samples1[1].cpu().numpy().reshape(32, ns[1]), want1)
        outs.append((losses, eng.get_engine().params.cpu().numpy().copy()))
    # several steps per graph launch: same schedule, same bits
    G, it, ph, sampler, model, ns = build(dev, "mean", True, False, csr=True)
    model.attach_device_epoch(it.train_nodes[:96], it.label_matrix)
    for _ in range(3):                                    # eager, capture, replay of the 2-step graph (+ priming step)
        model.set_epoch_order(it.train_nodes[:96])
        model.train_steps_device(32, 3, steps_per_launch=2)
    eng.get_engine().sync()
    multi = eng.get_engine().params.cpu().numpy().copy()
    G, it, ph, sampler, model, ns = build(dev, "mean", True, False, csr=True)
    model.use_graphs = model.pipeline = False
    model.attach_device_epoch(it.train_nodes[:96], it.label_matrix)
    for _ in range(3):
        model.set_epoch_order(it.train_nodes[:96])
        for _ in range(3):
            model.train_step_device(32)
    eng.get_engine().sync()
    assert np.array_equal(multi, eng.get_engine().params.cpu().numpy())
    for other in (1, 3, 4):           # eager == hipGraph replay == fused pipeline, bitwise
        assert outs[0][0] == outs[other][0]
        assert np.array_equal(outs[0][1], outs[other][1])
    # fused sampler/head kernels vs the per-hop / unfused kernels: same maths, different summation order
    np.testing.assert_allclose(outs[0][0], outs[2][0], rtol=1e-5)
    np.testing.assert_allclose(outs[0][1], outs[2][1], rtol=1e-4, atol=1e-6)
    assert outs[0][0][2] < outs[0][0][0] + 0.5  # training is not diverging


@pytest.mark.parametrize("agg_type,concat", [("gcn", False), ("meanpool", True)])
def test_pipelined_equals_sequential_other_aggregators(dev, agg_type, concat):
    """The prefetch pipeline (GCN: single-term co-gather launch + split into the weight-gradient launch; pooling:
    nothing to prefetch) gives the bits of the sequential eager schedule."""
    outs = []
    for use_graphs, pipe in ((False, False), (True, True)):
        G, it, ph, sampler, model, ns = build(dev, agg_type, concat, False, csr=True)
        model.use_graphs, model.pipeline = use_graphs, pipe
        model.attach_device_epoch(it.train_nodes[:160], it.label_matrix)
        if pipe:
            model.train_steps_device(32, 5, steps_per_launch=2)
        else:
            for _ in range(5):
                model.train_step_device(32)
        eng.get_engine().sync()
        outs.append(eng.get_engine().params.cpu().numpy().copy())
    assert np.array_equal(outs[0], outs[1])


def test_split_weight_pieces_follow_every_update_under_graph_replay(dev):
    """The max-pool MLP on the step's distinct ids reads a three-piece bf16 copy of its weights (Engine.split_of).  The copy
    must be re-cut behind EVERY optimizer step also when the step is a replayed hipGraph: train (eager) / eval (eager) /
    train (captured) / train (replay) / eval == the same calls without graphs, bit for bit.  (Before round 5 the re-cut was
    decided by a host flag read at capture time: a train graph captured with the flag clean replayed on stale pieces.)"""
    outs = []
    for use_graphs in (False, True):
        G, it, ph, sampler, model, ns = build(dev, "maxpool", True, False, csr=True, n_nodes=1500)
        assert eng.get_engine().split_pool
        model.use_graphs = use_graphs
        rng = np.random.RandomState(11)
        B = 128                                         # 128 * (3 + 15) = 2304 gathered rows > 2048: the dedup + split path
        feeds = []
        for _ in range(6):
            b = rng.choice(it.train_nodes, size=B, replace=False).astype(np.int32)
            feeds.append({ph['batch']: b, ph['labels']: it.label_matrix[b], ph['batch_size']: B})
        seq = [("t", 0), ("e", 1), ("t", 2), ("t", 3), ("e", 4), ("t", 5), ("e", 1)]
        res = []
        for kind, i in seq:
            loss, preds = (model.train_step if kind == "t" else model.eval_step)(feeds[i])
            res.append((loss, preds.copy()))
        assert model.aggregators[0].mlp_layers[0].vars['weights'].split3 is not None, "the split-MFMA path did not run"
        outs.append((res, eng.get_engine().params.cpu().numpy().copy()))
    for (l0, p0), (l1, p1) in zip(outs[0][0], outs[1][0]):
        assert l0 == l1 and np.array_equal(p0, p1)
    assert np.array_equal(outs[0][1], outs[1][1])


def test_training_learns(dev):
    """A few epochs on a planted-community graph reach a high val micro-F1 (the metric's quality half)."""
    G, it, ph, sampler, model, ns = build(dev, "mean", True, False, csr=True, n_nodes=3000, dim=32)
    e = eng.get_engine()
    model.attach_device_epoch(it.train_nodes, it.label_matrix)
    B = 128
    for epoch in range(4):
        model.set_epoch_order(np.random.RandomState(epoch).permutation(it.train_nodes))
        for _ in range(len(it.train_nodes) // B):
            model.train_step_device(B)
    model.adj_info.assign(CSRAdjacency(it.test_csr[0], it.test_csr[1], G.n_nodes, e.device))
    val = it.val_nodes.astype(np.int32)
    loss, preds = model.eval_step({ph['batch']: val, ph['labels']: it.label_matrix[val], ph['batch_size']: len(val)})
    f1 = orc.calc_f1_micro(it.label_matrix[val], preds, False)
    assert f1 > 0.8, f1


def test_eval_between_device_steps_keeps_the_prefetched_batch(dev):
    """A host-fed step (eval_step / train_step(feed)) of the SAME batch size between two device-epoch steps must not
    touch the batch the pipeline has already prefetched (ids, labels, samples, layer-0 means live in parity buffers;
    host-fed batches own theirs): the next device step equals the one of an uninterleaved run, bitwise."""
    outs = []
    for interleave in (False, True):
        G, it, ph, sampler, model, ns = build(dev, "mean", True, False, csr=True)
        model.attach_device_epoch(it.train_nodes[:160], it.label_matrix)
        for _ in range(2):
            model.train_step_device(32)
        if interleave:
            val = it.val_nodes[:32].astype(np.int32)
            loss_v, preds_v = model.eval_step({ph['batch']: val, ph['labels']: it.label_matrix[val], ph['batch_size']: 32})
            assert np.isfinite(loss_v) and preds_v.shape == (32, G.num_classes)
        loss, preds = model.train_step_device(32, fetch=True)
        assert np.array_equal(model.samples1[0].cpu().numpy(), it.train_nodes[64:96])
        outs.append((loss, preds.copy(), eng.get_engine().params.cpu().numpy().copy()))
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
