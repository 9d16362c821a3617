"""-m gpu: the hot path AT BASELINE.json configs[1]'s FULL size (N = 232,965, average degree 492 = 55 M undirected
edges, F = 602, C = 41, B = 512, fan-out 25x10) -- the graph bench.py times, built on the device -- checked through
what stays cheap at that size:
  * S1/S2: the ids drawn by the fused fan-out sampler inside the 8-steps-per-launch hipGraph schedule, bit for bit
    against the CPU restatement of the counter hash (the oracle needs only the CSR, which is copied to the host);
    every sampled id is a neighbor of its parent (or the pad id for a degree-0 parent);
  * A0/A2 (K2): gather+mean of the hop-2 rows (5120 x 25 rows of 602 floats out of a 567 MB table) against a plain
    torch fp32 reference of the same op, plus its linearity in the table;
  * the schedule: 8 steps per graph launch == one step per launch, bitwise, and the loss falls."""
import numpy as np
import pytest
import torch

from graphsage_amd import engine as eng
from graphsage_amd import inits, ops
from graphsage_amd.models import Placeholder, SAGEInfo
from graphsage_amd.neigh_samplers import AdjInfo, CSRAdjacency, UniformNeighborSampler
from graphsage_amd.ops import Mat
from graphsage_amd.supervised_models import SupervisedGraphsage
from graphsage_amd.utils import reddit_shaped_device
from oracle import sampler_hash

pytestmark = pytest.mark.gpu
B, S1, S2, DIM = 512, 25, 10, 128
_cache = {}


def graph(dev):
    if "DG" not in _cache:
        _cache["DG"] = reddit_shaped_device(dev, avg_degree=492, feat_signal=0.02)
    return _cache["DG"]


def build(dev):
    DG = graph(dev)
    eng.reset_engine()
    inits.set_seed(5)
    ph = {'labels': Placeholder('labels'), 'batch': Placeholder('batch1'), 'dropout': Placeholder('dropout', 0.),
          'batch_size': Placeholder('batch_size')}
    adj_info = AdjInfo(CSRAdjacency.from_device(DG.train_csr[0], DG.train_csr[1], DG.n_nodes))
    sampler = UniformNeighborSampler(adj_info, seed=123)
    layer_infos = [SAGEInfo("node", sampler, S1, DIM), SAGEInfo("node", sampler, S2, DIM)]
    model = SupervisedGraphsage(DG.num_classes, ph, DG.feats, adj_info, DG.deg, layer_infos, concat=True,
                                aggregator_type="mean", sigmoid_loss=False, learning_rate=0.01, weight_decay=0.0)
    order = np.random.RandomState(123).permutation(DG.train_nodes).astype(np.int32)
    model.attach_device_epoch(order, DG.label_table)
    return DG, model, order


def test_fullsize_sampler_and_schedule(dev):
    DG, model, order = build(dev)
    N = DG.n_nodes
    assert N == 232965 and DG.train_csr[1].numel() > 40_000_000          # Reddit-sized train view
    steps = 17                                                           # priming + eager + capture + replay of 8-step graphs
    model.train_steps_device(B, steps, steps_per_launch=8)
    loss, _ = model._fetch(B)
    got = [s.cpu().numpy() for s in model.samples1]
    t = steps - 1
    batch = order[t * B:(t + 1) * B]
    assert np.array_equal(got[0], batch)
    rowptr, col = DG.train_csr[0].cpu().numpy(), DG.train_csr[1].cpu().numpy()
    hop1 = sampler_hash.sample_uniform_csr(rowptr, col, N, N, batch, S2, 123, t, 0)
    hop2 = sampler_hash.sample_uniform_csr(rowptr, col, N, N, hop1.reshape(-1), S1, 123, t, 1)
    assert np.array_equal(got[1], hop1.reshape(-1)) and np.array_equal(got[2], hop2.reshape(-1))      # bit exact
    # membership (independent of the hash restatement): (parent, child) is an edge of the train view, or child == pad
    for parents, kids, s in ((got[0], got[1], S2), (got[1], got[2], S1)):
        par = np.repeat(parents.astype(np.int64), s)
        kid = kids.astype(np.int64)
        real = kid != N
        lo, hi = rowptr[par[real]], rowptr[par[real] + 1]
        pos = np.array([np.searchsorted(col[a:b], k) for a, b, k in zip(lo[:4000], hi[:4000], kid[real][:4000])])
        assert np.array_equal(col[lo[:4000] + pos], kid[real][:4000])
        deg_par = np.where(par < N, rowptr[np.minimum(par, N - 1) + 1] - rowptr[np.minimum(par, N - 1)], 0)
        assert not (deg_par[~real] > 0).any()                            # pad only for degree-0 / pad parents
    assert real.mean() > 0.95
    params_multi = eng.get_engine().params.cpu().numpy().copy()
    # same number of steps, one hipGraph launch per step
    DG, model1, _ = build(dev)
    first = None
    for i in range(steps):
        out = model1.train_step_device(B, fetch=(i == 0))
        if i == 0:
            first = out[0]
    loss1, _ = model1._fetch(B)
    assert loss == loss1
    assert np.array_equal(params_multi, eng.get_engine().params.cpu().numpy())
    assert np.isfinite(loss) and loss < first


def test_fullsize_gather_mean_vs_torch_and_linear(dev):
    DG = graph(dev)
    X = DG.feats
    F, N = DG.feat_dim, DG.n_nodes
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    n = B * S2
    idx = torch.randint(0, N + 1, (n * S1,), device=dev, generator=gen, dtype=torch.int64).to(torch.int32)
    out = Mat.zeros(n, F, dev)
    ops.gather_mean_fwd(X, idx, n, S1, out=out)
    torch.cuda.synchronize()
    rows = X.buf[idx.long()][:, :F].view(n, S1, F)
    want = rows.to(torch.float64).mean(dim=1)
    got = out.buf[:, :F].to(torch.float64)
    assert float((got - want).abs().max()) < 1e-5
    assert not out.buf[:, F:].any()                                       # pad columns stay zero
    # linearity: mean over the rows of (2 X + 1) == 2 mean + 1 up to rounding (pad row excluded: it is not scaled)
    sub = torch.randint(0, N, (64 * S1,), device=dev, generator=gen, dtype=torch.int64).to(torch.int32)
    X2 = Mat(X.buf * 2.0 + 1.0, F)
    a, b = Mat.zeros(64, F, dev), Mat.zeros(64, F, dev)
    ops.gather_mean_fwd(X, sub, 64, S1, out=a)
    ops.gather_mean_fwd(X2, sub, 64, S1, out=b)
    torch.cuda.synchronize()
    assert float((b.buf[:, :F] - (2.0 * a.buf[:, :F] + 1.0)).abs().max()) < 1e-5
