"""-m gpu: the committed hand-computed golden fixtures (tests/golden/*.npz, plain-loop arithmetic written
independently of the oracle) pushed through the HIP path: reference-API calls (UniformNeighborSampler((ids, n)),
SampleAndAggregate.sample / .aggregate, aggregator((self_vecs, neigh_vecs))) -> C ABI -> gfx950 kernels.
Integer-valued fixtures are compared EXACTLY."""
import os

import numpy as np
import pytest
import torch

from graphsage_amd import engine as eng
from graphsage_amd.aggregators import GCNAggregator, MaxPoolingAggregator
from graphsage_amd.layers import Rows
from graphsage_amd.models import Placeholder, SAGEInfo
from graphsage_amd.neigh_samplers import AdjInfo, PaddedAdjacency, UniformNeighborSampler
from graphsage_amd.ops import Mat
from graphsage_amd.supervised_models import SupervisedGraphsage

pytestmark = pytest.mark.gpu
GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ph():
    return {'labels': Placeholder('labels'), 'batch': Placeholder('batch1'), 'dropout': Placeholder('dropout', 0.),
            'batch_size': Placeholder('batch_size')}


@pytest.mark.parametrize("fuse", [True, False])
def test_golden_tiny_mean_through_hip(dev, fuse):
    """tiny_mean.npz: sampler ids, both layer-0 hop outputs and the final two-layer mean output, exact."""
    g = np.load(os.path.join(GOLD_DIR, "tiny_mean.npz"))
    eng.reset_engine()
    e = eng.get_engine()
    adj_info = AdjInfo(PaddedAdjacency(g["adj"], e.device))
    sampler = UniformNeighborSampler(adj_info)
    ns = [int(v) for v in g["num_samples"]]
    dims = [int(v) for v in g["dims"]]
    # the concat epilogue of the gfx950 kernels needs out_dim % 4 == 0: the fixture's 2-wide layers are embedded in
    # 4-wide ones with zero weight columns (exact: the extra outputs are 0 and meet zero weight rows downstream)
    o, OP = dims[1], 4
    assert dims[1] == dims[2] == 2
    layer_infos = [SAGEInfo("node", sampler, ns[0], OP), SAGEInfo("node", sampler, ns[1], OP)]
    model = SupervisedGraphsage(3, _ph(), g["feats"], adj_info, None, layer_infos, concat=True, aggregator_type="mean")
    model.fuse_sampler = model.fuse_head = model.fuse_tail = fuse

    def _place(W, rows, n_rows=None):
        out = np.zeros((n_rows or (max(rows) + 1), OP), np.float32)
        for src, dst in enumerate(rows):
            out[dst, :o] = W[src]
        return out

    F = g["feats"].shape[1]
    l1_rows = [0, 1, OP, OP + 1]         # fixture concat [fs0, fs1, fn0, fn1] -> padded [fs0, fs1, 0, 0, fn0, fn1, 0, 0]
    model.aggregators[0].vars['self_weights'].assign(_place(g["W0_self"], list(range(F)), F))
    model.aggregators[0].vars['neigh_weights'].assign(_place(g["W0_neigh"], list(range(F)), F))
    model.aggregators[1].vars['self_weights'].assign(_place(g["W1_self"], l1_rows, 2 * OP))
    model.aggregators[1].vars['neigh_weights'].assign(_place(g["W1_neigh"], l1_rows, 2 * OP))
    keep = [0, 1, OP, OP + 1]            # the fixture's columns inside a padded concat output
    n = len(g["batch"])
    batch_dev = model.ids_buffer(n)[0][:n]
    batch_dev.copy_(torch.from_numpy(g["batch"]))
    torch.cuda.synchronize()
    sampler.new_step()
    sampler.inject_perms([g["perm0"], g["perm1"]])
    samples, support = model.sample(batch_dev, layer_infos, n)                      # models.py:254-275
    e.sync()
    assert support == [1, 2, 4]
    assert np.array_equal(samples[1].cpu().numpy(), g["samples1"])
    assert np.array_equal(samples[2].cpu().numpy(), g["samples2"])
    out, _ = model.aggregate(samples, [model.features], model.dims, model.num_samples, support, batch_size=n,
                             aggregators=model.aggregators, concat=True)            # models.py:278-330
    e.sync()
    assert np.array_equal(out.numpy()[:n][:, keep], g["out"])
    assert not out.numpy()[:n][:, [2, 3, OP + 2, OP + 3]].any()
    h_all = model._tape[0][4].numpy()[:, keep]                                      # layer-0 outputs of both hops
    assert np.array_equal(h_all[:n], g["l0_hop0"]) and np.array_equal(h_all[n:n + len(g["samples1"])], g["l0_hop1"])


def test_golden_gcn_and_maxpool_calls_through_hip(dev):
    """tiny_gcn_maxpool.npz: ONE GCNAggregator call and ONE MaxPoolingAggregator call, aggregator((self, neigh))."""
    g = np.load(os.path.join(GOLD_DIR, "tiny_mean.npz"))
    m = np.load(os.path.join(GOLD_DIR, "tiny_gcn_maxpool.npz"))
    eng.reset_engine()
    e = eng.get_engine()
    X = Mat.from_numpy(g["feats"], e.device, ld_multiple=32)
    n, s, d = len(g["batch"]), 2, g["feats"].shape[1]
    ids_self = torch.from_numpy(g["batch"]).to(e.device)
    ids_neigh = torch.from_numpy(g["samples1"]).to(e.device)
    gcn = GCNAggregator(d, m["W_gcn"].shape[1], dropout=0., concat=False)
    hid = m["W_mlp"].shape[1]
    o, OP = m["W_self"].shape[1], 4          # concat epilogue needs out_dim % 4 == 0: 2 -> 4 with zero weight columns
    mp = MaxPoolingAggregator(d, OP, model_size="small", dropout=0., concat=True)
    e.finalize()
    gcn.vars['weights'].assign(m["W_gcn"])
    # the reference's pooling MLP is 512 wide ("small", aggregators.py:139-140): the fixture's 3 hidden units are
    # embedded in the first columns; the zero columns give relu(0) = 0 pooled activations that meet zero weights
    W_mlp = np.zeros((d, mp.hidden_dim), np.float32); W_mlp[:, :hid] = m["W_mlp"]
    b_mlp = np.zeros((mp.hidden_dim,), np.float32); b_mlp[:hid] = m["b_mlp"]
    W_neigh = np.zeros((mp.hidden_dim, OP), np.float32); W_neigh[:hid, :o] = m["W_neigh"]
    W_self = np.zeros((d, OP), np.float32); W_self[:, :o] = m["W_self"]
    mp.mlp_layers[0].vars['weights'].assign(W_mlp)
    mp.mlp_layers[0].vars['bias'].assign(b_mlp)
    mp.vars['neigh_weights'].assign(W_neigh)
    mp.vars['self_weights'].assign(W_self)
    torch.cuda.synchronize()
    self_vecs = Rows(X, ids_self, requires_grad=False)
    neigh_vecs = Rows(X, ids_neigh, requires_grad=False).reshape((n, s, d))
    y = gcn((self_vecs, neigh_vecs))
    e.sync()
    np.testing.assert_allclose(y.numpy(), m["gcn_out"], rtol=1e-6, atol=1e-6)        # thirds: not exact in fp32
    y = mp((self_vecs, neigh_vecs))
    e.sync()
    assert np.array_equal(y.numpy()[:, [0, 1, OP, OP + 1]], m["maxpool_out"])         # integers: exact
