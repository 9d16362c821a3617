"""-m gpu: oracle parity of every TIMED BASELINE configuration at its own shapes, through the path bench.py times.

  * configs[3] (`aux.unsupervised`): unsupervised graphsage_mean at B = 512, fan-out 25x10, F = 602, dims 128/128,
    20 negatives through SampleAndAggregate.train_step(s)_device -> _pipelined_steps_unsup (device-resident pairs,
    hipGraph capture/replay, next step's gather riding in this step's launches): pairs, negatives and every sampled id
    bit-exact vs oracle/sampler_hash.py, then orc.unsupervised_fwd_bwd (models.py:332-405, prediction.py:68-110
    restated) on exactly those ids: loss, MRR, affinities, every gradient, parameters after clip + Adam at 1e-4.
  * configs[4] (`aux.rmat`): the RMAT configuration's shapes (F = 256, fan-out 15x10, C = 64, dims 128/128, B = 512)
    through bench.build_rmat at a reduced node count: same checks as tests/test_bench_parity_gpu.py.
  * configs[1] at FULL size (N = 232,965, average degree 492): training steps of the headline model on the graph
    bench.py times vs the oracle (the table is copied to the host once; the oracle touches only the sampled rows).
The graphs of the first two are smaller than the benched ones so that the NumPy oracle finishes in seconds per step."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from graphsage_amd import engine as eng
from graphsage_amd import inits
from graphsage_amd.minibatch import NodeMinibatchIterator
from graphsage_amd.models import Placeholder, SAGEInfo, SampleAndAggregate
from graphsage_amd.neigh_samplers import AdjInfo, CSRAdjacency, UniformNeighborSampler
from graphsage_amd.supervised_models import SupervisedGraphsage
from graphsage_amd.utils import reddit_shaped, reddit_shaped_device, run_random_walks
from oracle import graphsage_oracle as orc
from oracle import sampler_hash

pytestmark = pytest.mark.gpu
B, S1, S2, F, DIM, NEG = 512, 25, 10, 602, 128, 20


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _layer0_ties(model, n_roots, s2):
    """relu'(x) at pre-activations that cancel to ~1e-8 follows the device's sign pattern (injected like the sampler
    draws, see tests/test_bench_parity_gpu.py)."""
    h_dev = model._tape[0][4].numpy()
    pieces = [h_dev[:n_roots] > 0, h_dev[n_roots:n_roots + n_roots * s2] > 0]

    def ties(shape, _it=iter(pieces)):
        return next(_it, None)
    return ties


LAW, MAXDEG = sampler_hash.LAW_REFERENCE, 128          # bench.py's default sampling law (--sampler_law reference)


def _check_sampled_ids(got, roots, rowptr, col, N, fans, seed, t, segments=None):
    """`segments` = (start of batch2, start of the negatives) in the roots: the unsupervised pass, whose three root groups
    are three sample() calls of the reference (models.py:347-357) with their own column permutations."""
    assert np.array_equal(got[0], roots)
    prev, hop = roots, 0
    for f, g in zip(fans, got[1:]):
        rows_per_root = len(prev) // len(roots)
        seg = None if segments is None else tuple(b * rows_per_root for b in segments)
        want = sampler_hash.sample_uniform_csr_segments(rowptr, col, N, N, prev, f, seed, t, hop, len(fans), seg, law=LAW,
                                                        max_degree=MAXDEG)
        assert np.array_equal(g, want.reshape(-1)), "hop %d ids differ from the hash restatement" % (hop + 1)
        prev, hop = want.reshape(-1), hop + 1


# ------------------------------------------------------------------------------------------------ configs[3]
def test_unsupervised_benched_shapes_match_oracle(dev):
    steps, lr = 4, 0.001
    G = reddit_shaped(avg_degree=60, seed=123, n_nodes=60000, feat_dim=F, num_classes=41)
    it = NodeMinibatchIterator(G, None, {}, None, G.num_classes, batch_size=B, max_degree=128, build_padded=False)
    rowptr, col = it.train_csr
    eng.reset_engine()
    inits.set_seed(11)
    e = eng.get_engine()
    ph = {'batch1': Placeholder('batch1'), 'batch2': Placeholder('batch2'), 'neg_samples': Placeholder('neg'),
          'dropout': Placeholder('dropout', 0.), 'batch_size': Placeholder('batch_size')}
    adj_info = AdjInfo(CSRAdjacency(rowptr, col, G.n_nodes, e.device))
    sampler = UniformNeighborSampler(adj_info, seed=123, law="reference", max_degree=MAXDEG)
    layer_infos = [SAGEInfo("node", sampler, S1, DIM), SAGEInfo("node", sampler, S2, DIM)]
    model = SampleAndAggregate(ph, G.padded_features(), adj_info, it.deg, layer_infos, concat=True, aggregator_type="mean",
                               learning_rate=lr, weight_decay=0.0, neg_sample_size=NEG)
    pairs = run_random_walks(rowptr, col, it.train_nodes[:4000], num_walks=2)
    pairs = np.random.RandomState(0).permutation(pairs)[: 8 * B].astype(np.int32)
    model.attach_device_pairs(pairs)
    assert model.use_graphs and model.pipeline                       # the schedule bench.py's aux.unsupervised runs
    feats = G.padded_features()
    cdf = sampler_hash.unigram_cdf_u32(it.deg)
    n_roots = 2 * B + NEG
    adam_state = None
    for t in range(steps):                                           # eager, eager, capture, replay
        before = [{k: v.numpy().copy() for k, v in a.vars.items()} for a in model.aggregators]
        loss, ranks, aff_all, mrr, outputs1 = model.train_step_device(B, fetch=True)
        # ---- the batch of pairs, the negatives and the sampled ids: bit exact
        sel = pairs[(t * B + np.arange(B)) % len(pairs)]
        neg = sampler_hash.sample_unigram(cdf, NEG, 123, t)
        roots = np.concatenate([sel[:, 0], sel[:, 1], neg]).astype(np.int32)
        got = [s.cpu().numpy() for s in model.samples1]
        _check_sampled_ids(got, roots, rowptr, col, G.n_nodes, [S2, S1], 123, t, segments=(B, 2 * B))
        # batch1, batch2 and the negatives do NOT share their column permutations (six tf.random_shuffle per step)
        for h, f in ((1, S2), (2, S1)):
            assert len({tuple(sampler_hash.call_columns(123, t, g * 2 + h - 1, f, MAXDEG).tolist()) for g in range(3)}) == 3
        # ---- the oracle on exactly these ids
        with orc.relu_ties_from(_layer0_ties(model, n_roots, S2)):
            res = orc.unsupervised_fwd_bwd(before, feats, got, [1, S2, S2 * S1], model.dims, [S1, S2], B, NEG, "mean", True,
                                           weight_decay=0.0)
        np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5, err_msg="step %d" % t)
        np.testing.assert_allclose(outputs1, res["outputs1"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(aff_all, res["aff_all"], rtol=1e-4, atol=1e-4)
        margin = np.abs(res["aff_all"][:, :-1] - res["aff_all"][:, -1:]).min(axis=1) > 1e-4     # float near-ties aside
        assert np.array_equal(ranks[margin], res["ranks"][margin])
        assert abs(mrr - res["mrr"]) < 2e-3
        dev_g = [{k: v.grad.numpy().copy() for k, v in a.vars.items()} for a in model.aggregators]
        for li in range(2):
            for k, w in res["grads"][li].items():
                g = dev_g[li][k].reshape(w.shape)
                assert np.abs(w).max() > 0, (li, k)
                np.testing.assert_allclose(g, w, rtol=1e-4, atol=1e-4 * np.abs(w).max(), err_msg="step %d %d/%s" % (t, li, k))
        # ---- clip +-5 and TF Adam (models.py:379-383), moments carried across steps
        after = [{k: v.numpy().copy() for k, v in a.vars.items()} for a in model.aggregators]
        if adam_state is None:
            adam_state = [{k: (np.zeros_like(p), np.zeros_like(p)) for k, p in d.items()} for d in before]
        for li in range(2):
            for k, p0 in before[li].items():
                want = p0.copy()
                m, v = adam_state[li][k]
                orc.adam_tf_update(want, orc.clip_by_value(dev_g[li][k]).reshape(want.shape), m, v, t + 1, lr)
                np.testing.assert_allclose(after[li][k], want, rtol=1e-5, atol=2e-7, err_msg="step %d %d/%s" % (t, li, k))
                assert np.abs(after[li][k] - p0).max() > 1e-4            # the update is visible at this lr
    # the multi-step graph schedule (8 steps per launch, what bench.py times) continues from here with the same bits as
    # single-step launches
    params_a = None
    for mode in ("multi", "single"):
        eng.reset_engine()
        inits.set_seed(11)
        e = eng.get_engine()
        adj_info = AdjInfo(CSRAdjacency(rowptr, col, G.n_nodes, e.device))
        sampler = UniformNeighborSampler(adj_info, seed=123, law="reference", max_degree=MAXDEG)
        layer_infos = [SAGEInfo("node", sampler, S1, DIM), SAGEInfo("node", sampler, S2, DIM)]
        m2 = SampleAndAggregate(ph, feats, adj_info, it.deg, layer_infos, concat=True, aggregator_type="mean",
                                learning_rate=lr, weight_decay=0.0, neg_sample_size=NEG)
        m2.attach_device_pairs(pairs)
        if mode == "multi":
            m2.train_steps_device(B, 25, steps_per_launch=8)
        else:
            for _ in range(25):
                m2.train_step_device(B)
        e.sync()
        p = e.params.cpu().numpy().copy()
        if params_a is None:
            params_a = p
        else:
            assert np.array_equal(params_a, p)


# ------------------------------------------------------------------------------------------------ configs[4]
def test_rmat_benched_shapes_match_oracle(dev):
    bench = _bench()
    args = bench.parse_args(["--workload", "rmat", "--nodes", "200000", "--rmat-edges", "4000000"])
    assert (args.samples_1, args.samples_2, args.feat_dim, args.classes, args.dim_1, args.batch_size) == (15, 10, 256, 64, 128, 512)
    inits.set_seed(11)
    e, model, ph, order, labels, n_edges = bench.build_rmat(args, 1, 0)
    N, s1, s2, C = args.nodes, 15, 10, 64
    model.attach_device_epoch(order, labels)
    adj = model.layer_infos[0].neigh_sampler.adj_info.current
    rowptr, col = adj.rowptr.cpu().numpy(), adj.col.cpu().numpy()
    feats = model.features.numpy()
    label_h = labels.numpy()
    assert feats.shape == (N + 1, 256) and not feats[N].any() and np.abs(feats[:N]).max() <= 1.0
    adam_state = None
    for t in range(4):
        before = {"agg": [{k: v.numpy().copy() for k, v in a.vars.items()} for a in model.aggregators],
                  "node_pred": {"weights": model.node_pred.vars['weights'].numpy().copy(),
                                "bias": model.node_pred.vars['bias'].numpy().reshape(-1).copy()}}
        loss, preds = model.train_step_device(B, fetch=True)
        batch = order[t * B:(t + 1) * B]
        got = [s.cpu().numpy() for s in model.samples1]
        _check_sampled_ids(got, batch, rowptr, col, N, [s2, s1], 123, t)
        with orc.relu_ties_from(_layer0_ties(model, B, s2)):
            res = orc.supervised_fwd_bwd(before, feats, got, [1, s2, s2 * s1], label_h[batch], model.dims, [s1, s2], B, "mean",
                                         True, False, weight_decay=0.0)
        np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5, err_msg="step %d" % t)
        np.testing.assert_allclose(preds, res["preds"], rtol=1e-4, atol=1e-4)
        dev_g = {"agg": [{k: v.grad.numpy().copy() for k, v in a.vars.items()} for a in model.aggregators],
                 "node_pred": {"weights": model.node_pred.vars['weights'].grad.numpy().copy(),
                               "bias": model.node_pred.vars['bias'].grad.numpy().reshape(-1).copy()}}
        for (name, g), (_, w) in zip(orc.flat_param_items(dev_g, "mean"), orc.flat_param_items(res["grads"], "mean")):
            np.testing.assert_allclose(g.reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()),
                                       err_msg="step %d %s" % (t, name))
        after = {"agg": [{k: v.numpy().copy() for k, v in a.vars.items()} for a in model.aggregators],
                 "node_pred": {"weights": model.node_pred.vars['weights'].numpy().copy(),
                               "bias": model.node_pred.vars['bias'].numpy().reshape(-1).copy()}}
        if adam_state is None:
            adam_state = [(np.zeros_like(p), np.zeros_like(p)) for _, p in orc.flat_param_items(before, "mean")]
        for (name, p0), (_, g), (_, p1), (m, v) in zip(orc.flat_param_items(before, "mean"), orc.flat_param_items(dev_g, "mean"),
                                                       orc.flat_param_items(after, "mean"), adam_state):
            want = p0.copy()
            orc.adam_tf_update(want, orc.clip_by_value(g).reshape(want.shape), m, v, t + 1, 0.01)
            np.testing.assert_allclose(p1.reshape(want.shape), want, rtol=1e-5, atol=2e-6, err_msg="step %d %s" % (t, name))


# ------------------------------------------------------------------------------------------------ configs[1], full size
def test_fullsize_training_steps_match_oracle(dev):
    """N = 232,965, average degree 492, F = 602, C = 41, B = 512, 25x10: the graph and the schedule bench.py times."""
    DG = reddit_shaped_device(dev, avg_degree=492, feat_signal=0.02)
    eng.reset_engine()
    inits.set_seed(5)
    ph = {'labels': Placeholder('labels'), 'batch': Placeholder('batch1'), 'dropout': Placeholder('dropout', 0.),
          'batch_size': Placeholder('batch_size')}
    adj_info = AdjInfo(CSRAdjacency.from_device(DG.train_csr[0], DG.train_csr[1], DG.n_nodes))
    sampler = UniformNeighborSampler(adj_info, seed=123, law="reference", max_degree=MAXDEG)
    layer_infos = [SAGEInfo("node", sampler, S1, DIM), SAGEInfo("node", sampler, S2, DIM)]
    model = SupervisedGraphsage(DG.num_classes, ph, DG.feats, adj_info, DG.deg, layer_infos, concat=True,
                                aggregator_type="mean", sigmoid_loss=False, learning_rate=0.01, weight_decay=0.0)
    order = np.random.RandomState(123).permutation(DG.train_nodes).astype(np.int32)
    model.attach_device_epoch(order, DG.label_table)
    N = DG.n_nodes
    assert N == 232965 and DG.train_csr[1].numel() > 40_000_000
    rowptr, col = DG.train_csr[0].cpu().numpy(), DG.train_csr[1].cpu().numpy()
    feats = DG.feats.numpy()                                         # 561 MB once; the oracle reads the sampled rows
    label_h = DG.label_table.numpy()
    for t in range(3):                                               # eager, eager (prefetched), captured graph
        before = {"agg": [{k: v.numpy().copy() for k, v in a.vars.items()} for a in model.aggregators],
                  "node_pred": {"weights": model.node_pred.vars['weights'].numpy().copy(),
                                "bias": model.node_pred.vars['bias'].numpy().reshape(-1).copy()}}
        loss, preds = model.train_step_device(B, fetch=True)
        batch = order[t * B:(t + 1) * B]
        got = [s.cpu().numpy() for s in model.samples1]
        _check_sampled_ids(got, batch, rowptr, col, N, [S2, S1], 123, t)
        assert (got[2] != N).mean() > 0.95
        with orc.relu_ties_from(_layer0_ties(model, B, S2)):
            res = orc.supervised_fwd_bwd(before, feats, got, [1, S2, S2 * S1], label_h[batch], model.dims, [S1, S2], B, "mean",
                                         True, False, weight_decay=0.0)
        np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5, err_msg="step %d" % t)
        np.testing.assert_allclose(preds, res["preds"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(model.outputs1.numpy(), res["outputs1"], rtol=1e-4, atol=1e-4)
        dev_g = {"agg": [{k: v.grad.numpy().copy() for k, v in a.vars.items()} for a in model.aggregators],
                 "node_pred": {"weights": model.node_pred.vars['weights'].grad.numpy().copy(),
                               "bias": model.node_pred.vars['bias'].grad.numpy().reshape(-1).copy()}}
        for (name, g), (_, w) in zip(orc.flat_param_items(dev_g, "mean"), orc.flat_param_items(res["grads"], "mean")):
            np.testing.assert_allclose(g.reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()),
                                       err_msg="step %d %s" % (t, name))


# ------------------------------------------------------------------------------------------------ configs[0]-shaped (PPI)
def test_ppi_shaped_sigmoid_121_classes_takes_the_fused_tail(dev):
    """The reference's own example (example_supervised.sh: graphsage_mean, --sigmoid, PPI: F = 50, C = 121 multi-label,
    fan-out 25x10, dims 128/128, B = 512) on a PPI-sized synthetic graph through the device-epoch path: 121 classes = two
    64-class groups of the fused tail launch, so the step is the 4-launch schedule; sampled ids bit-exact, then loss /
    sigmoid preds / every gradient vs the oracle at 1e-4 for 4 steps (eager, eager, capture, replay)."""
    Fp, Cp = 50, 121
    G = reddit_shaped(avg_degree=28, seed=7, n_nodes=15000, feat_dim=Fp, num_classes=Cp)
    it = NodeMinibatchIterator(G, None, {}, None, G.num_classes, batch_size=B, max_degree=128, build_padded=False)
    rowptr, col = it.train_csr
    rng = np.random.RandomState(3)
    labels = (rng.rand(G.n_nodes + 1, Cp) < 0.3).astype(np.float32)         # multi-hot (PPI: 121 gene-ontology sets)
    eng.reset_engine()
    inits.set_seed(11)
    e = eng.get_engine()
    ph = {'labels': Placeholder('labels'), 'batch': Placeholder('batch1'), 'dropout': Placeholder('dropout', 0.),
          'batch_size': Placeholder('batch_size')}
    adj_info = AdjInfo(CSRAdjacency(rowptr, col, G.n_nodes, e.device))
    sampler = UniformNeighborSampler(adj_info, seed=123, law="reference", max_degree=MAXDEG)
    layer_infos = [SAGEInfo("node", sampler, S1, DIM), SAGEInfo("node", sampler, S2, DIM)]
    model = SupervisedGraphsage(Cp, ph, G.padded_features(), adj_info, it.deg, layer_infos, concat=True,
                                aggregator_type="mean", sigmoid_loss=True, learning_rate=0.01, weight_decay=0.0)
    order = np.random.RandomState(123).permutation(it.train_nodes).astype(np.int32)
    model.attach_device_epoch(order, labels)
    feats = G.padded_features()
    for t in range(4):
        before = {"agg": [{k: v.numpy().copy() for k, v in a.vars.items()} for a in model.aggregators],
                  "node_pred": {"weights": model.node_pred.vars['weights'].numpy().copy(),
                                "bias": model.node_pred.vars['bias'].numpy().reshape(-1).copy()}}
        loss, preds = model.train_step_device(B, fetch=True)
        assert model._tail_used, "C = 121 must take the fused tail launch"
        batch = order[t * B:(t + 1) * B]
        got = [s.cpu().numpy() for s in model.samples1]
        _check_sampled_ids(got, batch, rowptr, col, G.n_nodes, [S2, S1], 123, t)
        with orc.relu_ties_from(_layer0_ties(model, B, S2)):
            res = orc.supervised_fwd_bwd(before, feats, got, [1, S2, S2 * S1], labels[batch], model.dims, [S1, S2], B, "mean",
                                         True, True, weight_decay=0.0)
        np.testing.assert_allclose(loss, res["loss"], rtol=1e-4, atol=1e-5, err_msg="step %d" % t)
        assert preds.shape == (B, Cp)
        np.testing.assert_allclose(preds, res["preds"], rtol=1e-4, atol=1e-4)
        dev_g = {"agg": [{k: v.grad.numpy().copy() for k, v in a.vars.items()} for a in model.aggregators],
                 "node_pred": {"weights": model.node_pred.vars['weights'].grad.numpy().copy(),
                               "bias": model.node_pred.vars['bias'].grad.numpy().reshape(-1).copy()}}
        for (name, g), (_, w) in zip(orc.flat_param_items(dev_g, "mean"), orc.flat_param_items(res["grads"], "mean")):
            np.testing.assert_allclose(g.reshape(w.shape), w, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(w).max()),
                                       err_msg="step %d %s" % (t, name))
