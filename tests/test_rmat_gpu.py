"""-m gpu: BASELINE configs[4] (RMAT graph generated in HBM, supervised graphsage_mean fan-out 15x10) at a small scale:
CSR invariants, sampler membership, and a few device-epoch training steps through bench.py's own builder."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_rmat_csr_invariants(dev):
    from graphsage_amd.utils import rmat_csr_device
    N, E = 5000, 120000
    rowptr, col = rmat_csr_device(N, E, dev, seed=3)
    rp, c = rowptr.cpu().numpy(), col.cpu().numpy()
    assert rp.shape == (N + 1,) and rp[0] == 0 and rp[-1] == E and np.all(np.diff(rp) >= 0)
    assert c.shape == (E,) and c.min() >= 0 and c.max() < N
    deg = np.diff(rp)
    assert deg.max() > 20 * deg.mean()            # R-MAT (0.57, 0.19, 0.19, 0.05) is heavy-tailed
    rowptr2, col2 = rmat_csr_device(N, E, dev, seed=3)
    assert torch.equal(rowptr, rowptr2) and torch.equal(col, col2)    # deterministic in the seed


def test_rmat_training_steps(dev):
    bench = _bench()
    args = bench.parse_args(["--workload", "rmat", "--nodes", "20000", "--rmat-edges", "400000", "--feat_dim", "64",
                             "--classes", "16", "--batch_size", "128", "--dim_1", "32", "--dim_2", "32"])
    assert (args.samples_1, args.samples_2) == (15, 10)
    e, model, ph, order, labels, n_edges = bench.build_rmat(args, 1, 0)
    assert n_edges == 400000 and order.shape == (20000,) and sorted(order.tolist()) == list(range(20000))
    model.attach_device_epoch(order, labels)
    model.train_steps_device(128, 20, steps_per_launch=8)
    loss, preds = model._fetch(128)
    assert np.isfinite(loss) and preds.shape == (128, 16)
    # every sampled hop-1 id is a neighbor of its root (or the pad id for degree-0 roots)
    adj = model.layer_infos[0].neigh_sampler.adj_info.current
    rp, col = adj.rowptr.cpu().numpy(), adj.col.cpu().numpy()
    roots = model.samples1[0].cpu().numpy()
    hop1 = model.samples1[1].cpu().numpy().reshape(len(roots), 10)
    for r, row in zip(roots[:32], hop1[:32]):
        nb = set(col[rp[r]:rp[r + 1]].tolist())
        assert all((v in nb) if nb else (v == 20000) for v in row.tolist())
