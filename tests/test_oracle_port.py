"""CPU tests: the NumPy oracle (hand-written backward) against the independent torch-autograd port
(oracle/torch_port.py) for every aggregator of SURVEY §8a and for the unsupervised objective -- the second pin of
VERDICT r01 "missing #1".  fp64 comparisons are tight (1e-9); an fp32 case checks the 1e-4 budget the device
tests use."""
import numpy as np
import pytest

from oracle import graphsage_oracle as orc
from oracle import torch_port


def _setup(rng, agg, concat, dtype, N=60, F=7, C=5, B=6, hidden=9, ns=(4, 3), dims_out=(6, 5)):
    feat = np.vstack([rng.normal(size=(N, F)), np.zeros((1, F))]).astype(dtype)
    neigh = [list(rng.choice(N, size=rng.integers(0, 10), replace=False)) for _ in range(N)]
    adj, _ = orc.construct_adj(neigh, 8, rng)
    ns = list(ns)
    dims = [F] + list(dims_out)
    if agg == "gcn":
        dims = [F] + [2 * d for d in dims_out]
    params = orc.make_supervised_params(agg, dims, C, concat, rng, dtype=dtype)
    if agg in ("maxpool", "meanpool"):
        for p in params["agg"]:
            p["mlp_weights"] = p["mlp_weights"][:, :hidden].copy()
            p["mlp_bias"] = (rng.normal(size=hidden) * 0.1).astype(dtype)
            p["neigh_weights"] = orc.glorot((hidden, p["neigh_weights"].shape[1]), rng, dtype)
    params["node_pred"]["bias"] = (rng.normal(size=C) * 0.1).astype(dtype)
    return feat, adj, ns, dims, params


CASES = [("mean", True), ("mean", False), ("gcn", False), ("maxpool", True), ("maxpool", False), ("meanpool", True)]


@pytest.mark.parametrize("agg,concat", CASES)
@pytest.mark.parametrize("sig", [False, True])
def test_supervised_oracle_equals_autograd_port_fp64(agg, concat, sig):
    import torch
    rng = np.random.default_rng(11)
    N, C, B = 60, 5, 6
    feat, adj, ns, dims, params = _setup(rng, agg, concat, np.float64, N=N, C=C, B=B)
    perms = [rng.permutation(8), rng.permutation(8)]
    batch = rng.choice(N, B, replace=False)
    samples, ss = orc.sample(adj, batch, ns, perms)
    labels = (rng.random((B, C)) > 0.5).astype(np.float64) if sig else np.eye(C)[rng.integers(0, C, B)]
    wd = 0.013
    a = orc.supervised_fwd_bwd(params, feat, samples, ss, labels, dims, ns, B, agg, concat, sig, weight_decay=wd)
    b = torch_port.supervised(params, feat, samples, ss, labels, dims, ns, B, agg, concat, sig, weight_decay=wd,
                              dtype=torch.float64)
    assert abs(a["loss"] - b["loss"]) < 1e-10
    np.testing.assert_allclose(a["preds"], b["preds"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a["outputs1"], b["outputs1"], rtol=1e-9, atol=1e-12)
    for (name, ga), (_, gb) in zip(orc.flat_param_items(a["grads"], agg), orc.flat_param_items(b["grads"], agg)):
        assert np.abs(gb).max() > 0 or name.endswith("mlp_bias"), name
        np.testing.assert_allclose(ga.reshape(gb.shape), gb, rtol=1e-8, atol=1e-11, err_msg=name)


@pytest.mark.parametrize("agg,concat", [("mean", True), ("gcn", False), ("maxpool", True)])
def test_supervised_oracle_equals_autograd_port_fp32(agg, concat):
    """The same comparison in fp32 at a wider shape: both sides inside the 1e-4 budget of the device tests."""
    rng = np.random.default_rng(12)
    N, C, B = 300, 9, 24
    feat, adj, ns, dims, params = _setup(rng, agg, concat, np.float32, N=N, F=40, C=C, B=B, hidden=32, ns=(6, 4),
                                         dims_out=(16, 16))
    perms = [rng.permutation(8), rng.permutation(8)]
    batch = rng.choice(N, B, replace=False)
    samples, ss = orc.sample(adj, batch, ns, perms)
    labels = np.eye(C, dtype=np.float32)[rng.integers(0, C, B)]
    a = orc.supervised_fwd_bwd(params, feat, samples, ss, labels, dims, ns, B, agg, concat, False, weight_decay=0.01)
    b = torch_port.supervised(params, feat, samples, ss, labels, dims, ns, B, agg, concat, False, weight_decay=0.01)
    assert abs(a["loss"] - b["loss"]) < 1e-5
    np.testing.assert_allclose(a["preds"], b["preds"], rtol=1e-4, atol=1e-5)
    for (name, ga), (_, gb) in zip(orc.flat_param_items(a["grads"], agg), orc.flat_param_items(b["grads"], agg)):
        np.testing.assert_allclose(ga.reshape(gb.shape), gb, rtol=1e-4, atol=1e-4 * max(1e-2, np.abs(gb).max()),
                                   err_msg=name)


@pytest.mark.parametrize("agg,concat", [("mean", True), ("gcn", False), ("maxpool", True), ("meanpool", True)])
def test_unsupervised_oracle_equals_autograd_port(agg, concat):
    """models.py:332-405 + prediction.py:102-110: loss / MRR / ranks / aff_all / every aggregator gradient."""
    import torch
    rng = np.random.default_rng(13)
    N, B, n_neg = 80, 7, 5
    feat, adj, ns, dims, params = _setup(rng, agg, concat, np.float64, N=N, B=B)
    roots = np.concatenate([rng.choice(N, B), rng.choice(N, B), rng.choice(N, n_neg)])
    perms = [rng.permutation(8), rng.permutation(8)]
    samples, ss = orc.sample(adj, roots, ns, perms)
    wd = 0.02
    a = orc.unsupervised_fwd_bwd(params["agg"], feat, samples, ss, dims, ns, B, n_neg, agg, concat, weight_decay=wd)
    b = torch_port.unsupervised(params["agg"], feat, samples, ss, dims, ns, B, n_neg, agg, concat, weight_decay=wd,
                                dtype=torch.float64)
    assert abs(a["loss"] - b["loss"]) < 1e-10
    assert abs(a["mrr"] - b["mrr"]) < 1e-12
    assert np.array_equal(a["ranks"], b["ranks"])
    np.testing.assert_allclose(a["aff_all"], b["aff_all"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a["outputs1"], b["outputs1"], rtol=1e-9, atol=1e-12)
    for li in range(len(ns)):
        for k in a["grads"][li]:
            np.testing.assert_allclose(a["grads"][li][k], b["grads"][li][k].reshape(a["grads"][li][k].shape), rtol=1e-8,
                                       atol=1e-11, err_msg="layer %d %s" % (li, k))
