"""Second, independent pin of the NumPy oracle: the reference's TF graph restated with torch-CPU ops and
differentiated by torch.autograd (the analogue of tf.gradients / optimizer.compute_gradients).

TEST INFRASTRUCTURE ONLY (see oracle/graphsage_oracle.py header): imported by tests/ only.

The NumPy oracle has hand-written backward functions; this port has NONE -- every gradient comes from
autograd over a forward written directly from the reference lines, so an error in the oracle's backward
(or a mis-read of the forward that both finite differences and the oracle share) shows up as a mismatch:

  MeanAggregator._call            aggregators.py:43-64
  GCNAggregator._call             aggregators.py:101-116
  MaxPoolingAggregator._call      aggregators.py:168-195   (+ Dense, layers.py:104-116)
  MeanPoolingAggregator._call     aggregators.py:246-273
  SampleAndAggregate.aggregate    models.py:278-330
  SupervisedGraphsage.build/_loss supervised_models.py:78-126
  SampleAndAggregate._build/_loss models.py:332-391, BipartiteEdgePredLayer._xent_loss prediction.py:102-110
  _accuracy (MRR)                 models.py:393-405

PARITY UNPINNED still applies: TF 1.x cannot run here, so neither this port nor the NumPy oracle has been
compared with the reference's own outputs; they pin each other.
"""
import numpy as np
import torch


def _t(a, requires_grad=False, dtype=torch.float32):
    t = torch.as_tensor(np.ascontiguousarray(a), dtype=dtype)
    return t.clone().requires_grad_(requires_grad)


def params_to_torch(params, dtype=torch.float32):
    """NumPy parameter dicts (graphsage_oracle.make_supervised_params layout) -> leaf tensors with requires_grad."""
    out = {"agg": [{k: _t(v, True, dtype) for k, v in p.items()} for p in params["agg"]]}
    if "node_pred" in params:
        out["node_pred"] = {k: _t(v, True, dtype) for k, v in params["node_pred"].items()}
    return out


def _aggregator(agg_type, p, self_vecs, neigh_vecs, concat, last):
    act = (lambda x: x) if last else torch.relu                      # models.py:307-310: identity on the last layer
    if agg_type == "mean":
        neigh_means = neigh_vecs.mean(dim=1)                          # :48
        from_neighs = neigh_means @ p["neigh_weights"]                # :51
        from_self = self_vecs @ p["self_weights"]                     # :53
        out = torch.cat([from_self, from_neighs], dim=1) if concat else from_self + from_neighs   # :55-58
        return act(out)
    if agg_type == "gcn":
        means = torch.cat([neigh_vecs, self_vecs.unsqueeze(1)], dim=1).mean(dim=1)                # :106-107
        return act(means @ p["weights"])                              # :110-116 (concat ignored, :79)
    n, s, d = neigh_vecs.shape
    h = neigh_vecs.reshape(n * s, d)                                  # :176
    h = torch.relu(h @ p["mlp_weights"] + p["mlp_bias"])              # Dense, layers.py:104-116
    h = h.reshape(n, s, -1)                                           # :180
    pooled = h.max(dim=1).values if agg_type == "maxpool" else h.mean(dim=1)   # :181 / :259
    from_neighs = pooled @ p["neigh_weights"]                         # :183
    from_self = self_vecs @ p["self_weights"]                         # :184
    out = torch.cat([from_self, from_neighs], dim=1) if concat else from_self + from_neighs
    return act(out)


def aggregate(tp, feats, samples, support_sizes, dims, num_samples, batch_size, agg_type, concat):
    """models.py:278-330: hidden[h] = embedding_lookup(features, samples[h]); per layer, per hop."""
    K = len(num_samples)
    hidden = [feats.index_select(0, torch.as_tensor(np.asarray(s), dtype=torch.int64)) for s in samples]   # :299
    for layer in range(K):
        dim_mult = 2 if (concat and layer != 0) else 1                # :305
        nxt = []
        for hop in range(K - layer):                                  # :321
            neigh_dims = (batch_size * support_sizes[hop], num_samples[K - hop - 1], dim_mult * dims[layer])   # :323-325
            nxt.append(_aggregator(agg_type, tp["agg"][layer], hidden[hop], hidden[hop + 1].reshape(neigh_dims),
                                   concat, layer == K - 1))
        hidden = nxt
    return hidden[0]


def _l2_normalize(x):
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim=1, keepdim=True), min=1e-12))   # tf.nn.l2_normalize


def _decayed(agg_type):
    return ("weights",) if agg_type == "gcn" else ("neigh_weights", "self_weights")     # aggregator.vars only


def supervised(params, features, samples, support_sizes, labels, dims, num_samples, batch_size, agg_type="mean",
               concat=True, sigmoid_loss=False, weight_decay=0.0, dtype=torch.float32):
    """supervised_models.py:78-126 on injected samples.  Returns dict(loss, preds, outputs1, node_preds, grads) with
    the gradient layout of graphsage_oracle.supervised_fwd_bwd (NumPy arrays)."""
    tp = params_to_torch(params, dtype)
    feats = _t(features, False, dtype)
    out = aggregate(tp, feats, samples, support_sizes, dims, num_samples, batch_size, agg_type, concat)
    out_n = _l2_normalize(out)                                                          # :85
    logits = out_n @ tp["node_pred"]["weights"] + tp["node_pred"]["bias"]               # :88-92
    y = _t(labels, False, dtype)
    loss = torch.zeros((), dtype=dtype)
    for p in tp["agg"]:                                                                 # :104-106
        for k in _decayed(agg_type):
            loss = loss + weight_decay * (p[k] * p[k]).sum() / 2
    for v in tp["node_pred"].values():                                                  # :107-108
        loss = loss + weight_decay * (v * v).sum() / 2
    if sigmoid_loss:                                                                    # :112-114
        loss = loss + torch.nn.functional.binary_cross_entropy_with_logits(logits, y, reduction="mean")
        preds = torch.sigmoid(logits)
    else:                                                                               # :116-118
        loss = loss + (-(y * torch.log_softmax(logits, dim=1)).sum(dim=1)).mean()
        preds = torch.softmax(logits, dim=1)
    leaves = [(("agg", li, k), v) for li, p in enumerate(tp["agg"]) for k, v in p.items()] + \
             [(("node_pred", k), v) for k, v in tp["node_pred"].items()]
    gs = torch.autograd.grad(loss, [v for _, v in leaves], allow_unused=True)
    grads = {"agg": [dict() for _ in tp["agg"]], "node_pred": {}}
    for (key, v), g in zip(leaves, gs):
        g = (torch.zeros_like(v) if g is None else g).numpy()
        if key[0] == "agg":
            grads["agg"][key[1]][key[2]] = g
        else:
            grads["node_pred"][key[1]] = g
    return {"loss": float(loss.detach()), "preds": preds.detach().numpy(), "outputs1": out_n.detach().numpy(),
            "node_preds": logits.detach().numpy(), "grads": grads}


def unsupervised(params_agg, features, samples, support_sizes, dims, num_samples, batch_size, n_neg, agg_type="mean",
                 concat=True, weight_decay=0.0, neg_sample_weights=1.0, dtype=torch.float32):
    """models.py:332-405 + prediction.py:68-110 (bilinear_weights=False, xent loss) on injected samples whose roots
    are [batch1 | batch2 | negatives].  Returns dict(loss, mrr, ranks, aff_all, outputs1, grads)."""
    tp = params_to_torch({"agg": params_agg}, dtype)
    feats = _t(features, False, dtype)
    B = batch_size
    n_roots = 2 * B + n_neg
    out = aggregate(tp, feats, samples, support_sizes, dims, num_samples, n_roots, agg_type, concat)
    out_n = _l2_normalize(out)                                                          # :368-370
    o1, o2, neg = out_n[:B], out_n[B:2 * B], out_n[2 * B:]
    aff = (o1 * o2).sum(dim=1)                                                          # prediction.py:79
    neg_aff = o1 @ neg.t()                                                              # prediction.py:91
    sp = torch.nn.functional.softplus
    true_xent = sp(-aff)                                # sigmoid_cross_entropy_with_logits(labels=1)   :105-106
    negative_xent = sp(neg_aff)                         # sigmoid_cross_entropy_with_logits(labels=0)   :107-108
    loss = true_xent.sum() + neg_sample_weights * negative_xent.sum()                   # :109
    for p in tp["agg"]:                                                                 # models.py:386-388
        for k in _decayed(agg_type):
            loss = loss + weight_decay * (p[k] * p[k]).sum() / 2
    loss = loss / B                                                                     # models.py:378
    leaves = [((li, k), v) for li, p in enumerate(tp["agg"]) for k, v in p.items()]
    gs = torch.autograd.grad(loss, [v for _, v in leaves], allow_unused=True)
    grads = [dict() for _ in tp["agg"]]
    for ((li, k), v), g in zip(leaves, gs):
        grads[li][k] = (torch.zeros_like(v) if g is None else g).numpy()
    # _accuracy (models.py:393-405): rank of the true pair among [negatives | true], ties towards the negatives
    aff_all = torch.cat([neg_aff, aff.unsqueeze(1)], dim=1).detach()
    ranks = (aff_all[:, :-1] >= aff_all[:, -1:]).sum(dim=1)
    mrr = (1.0 / (ranks.to(torch.float64) + 1.0)).mean()
    return {"loss": float(loss.detach()), "mrr": float(mrr), "ranks": ranks.numpy(), "aff_all": aff_all.numpy(),
            "outputs1": o1.detach().numpy(), "grads": grads}
