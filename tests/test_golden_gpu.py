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
    layer_infos = [SAGEInfo("node", sampler, ns[0], dims[1]), SAGEInfo("node", sampler, ns[1], dims[2])]
    model = SupervisedGraphsage(3, _ph(), g["feats"], adj_info, None, layer_infos, concat=True, aggregator_type="mean")
    model.fuse_sampler = model.fuse_head = fuse
    for li, a in enumerate(model.aggregators):
        a.vars['self_weights'].assign(g["W%d_self" % li])
        a.vars['neigh_weights'].assign(g["W%d_neigh" % li])
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
    assert np.array_equal(out.numpy()[:n], g["out"])
    h_all = model._tape[0][4].numpy()                                               # layer-0 outputs of both hops
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
    mp = MaxPoolingAggregator(d, m["W_self"].shape[1], model_size="small", dropout=0., concat=True)
    e.finalize()
    gcn.vars['weights'].assign(m["W_gcn"])
    # the reference's pooling MLP is 512 wide ("small", aggregators.py:139-140): the fixture's 3 hidden units are
    # embedded in the first columns; the zero columns give relu(0) = 0 pooled activations that meet zero weights
    W_mlp = np.zeros((d, mp.hidden_dim), np.float32); W_mlp[:, :hid] = m["W_mlp"]
    b_mlp = np.zeros((mp.hidden_dim,), np.float32); b_mlp[:hid] = m["b_mlp"]
    W_neigh = np.zeros((mp.hidden_dim, m["W_neigh"].shape[1]), np.float32); W_neigh[:hid] = m["W_neigh"]
    mp.mlp_layers[0].vars['weights'].assign(W_mlp)
    mp.mlp_layers[0].vars['bias'].assign(b_mlp)
    mp.vars['neigh_weights'].assign(W_neigh)
    mp.vars['self_weights'].assign(m["W_self"])
    torch.cuda.synchronize()
    self_vecs = Rows(X, ids_self, requires_grad=False)
    neigh_vecs = Rows(X, ids_neigh, requires_grad=False).reshape((n, s, d))
    y = gcn((self_vecs, neigh_vecs))
    e.sync()
    np.testing.assert_allclose(y.numpy(), m["gcn_out"], rtol=1e-6, atol=1e-6)        # thirds: not exact in fp32
    y = mp((self_vecs, neigh_vecs))
    e.sync()
    assert np.array_equal(y.numpy(), m["maxpool_out"])                                # integers: exact
