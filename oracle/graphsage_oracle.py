"""CPU oracle for the GraphSAGE sample -> gather -> aggregate hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under graphsage_amd/ may import this module.
Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.

PARITY PINNED (round 4) to the reference's OWN source: tests/golden/make_ref_fixtures.py puts /root/reference on sys.path and
executes graphsage/{minibatch,neigh_samplers,models,aggregators,layers,supervised_models,prediction}.py UNMODIFIED on
tests/tf1_shim (a torch-backed stand-in for the ~70 TensorFlow 1.x entry points they use; TF 1.8 itself is not installable
here) and commits what they computed as tests/golden/ref_*.npz: padded tables, sampled ids, loss, predictions, embeddings,
MRR, every gradient and the parameters after clip + Adam for 14 runs (mean / GCN / max-pool / mean-pool, softmax and sigmoid,
2 and 3 layers, identity features, dropout, unsupervised, evaluation on the test adjacency, short last batch,
num_samples == max_degree, degree-0 and val/test nodes).  tests/test_ref_pin.py: THIS FILE == those fixtures, float64 twin
at 1e-9, float32 at 1e-4, integer outputs bit-exact.  Every function still cites the reference file:line it follows
(paths relative to /root/reference/graphsage/).  Further pins:
 (a) hand-computed integer fixtures in tests/golden/ (see tests/golden/make_golden.py, make_golden_more.py) and the
     big-int known answers of the sampler / dropout hashes (hash_kat.npz, law_kat.npz),
 (b) finite-difference checks of every backward function (tests/test_oracle.py),
 (c) an independent second restatement in torch whose gradients come from autograd (oracle/torch_port.py,
     tests/test_oracle_port.py: forward values and hand-derived gradients must agree, fp64 1e-9 / fp32 1e-4),
 (d) sklearn for micro/macro-F1.

All arithmetic is done in the dtype of the inputs (float32 for parity runs,
float64 for finite-difference checks).  Summation order inside TF's
reduce_mean/matmul is TF/Eigen-internal, so comparisons against this oracle use
rtol/atol 1e-4, never bitwise, except for integer (index) outputs.
"""
import numpy as np

# --------------------------------------------------------------------------
# S0  padded adjacency table            minibatch.py:227-259
# --------------------------------------------------------------------------

def construct_adj(neigh_lists, max_degree, rng, skip_mask=None):
    """Padded adjacency [N+1, max_degree] + degree vector.

    minibatch.py:227-245 (train table, skip_mask = val|test flags so those rows
    stay all-pad, :232-233) and :247-259 (test table, skip_mask=None).
    `neigh_lists[i]` must already exclude train_removed edges when building the
    train table (:234-236).  Pad id is N (:228).  Rows are down-sampled without
    replacement (:240-241) or up-sampled with replacement (:242-243) ONCE.
    """
    n = len(neigh_lists)
    adj = np.full((n + 1, max_degree), n, dtype=np.int32)
    deg = np.zeros((n,), dtype=np.int64)
    for i in range(n):
        if skip_mask is not None and skip_mask[i]:
            continue
        nb = np.asarray(neigh_lists[i], dtype=np.int32)
        deg[i] = len(nb)
        if len(nb) == 0:
            continue
        if len(nb) > max_degree:
            nb = rng.choice(nb, max_degree, replace=False)
        elif len(nb) < max_degree:
            nb = rng.choice(nb, max_degree, replace=True)
        adj[i, :] = nb
    return adj, deg


# --------------------------------------------------------------------------
# S1  UniformNeighborSampler._call      neigh_samplers.py:24-29
# --------------------------------------------------------------------------

def uniform_neighbor_sampler(adj, ids, num_samples, col_perm):
    """rows = adj[ids] (:26); shuffle the COLUMN axis with one permutation shared
    by every row of the call (:27 transpose/random_shuffle/transpose); keep the
    first num_samples columns (:28).

    TF's random_shuffle stream cannot be reproduced, so the permutation is an
    input: `col_perm` is a permutation of range(max_degree) (or its first
    num_samples entries).  out[i, j] = adj[ids[i], col_perm[j]].
    """
    ids = np.asarray(ids, dtype=np.int64)
    cols = np.asarray(col_perm[:num_samples], dtype=np.int64)
    return adj[ids][:, cols].astype(np.int32)


# --------------------------------------------------------------------------
# S2  SampleAndAggregate.sample         models.py:254-275
# --------------------------------------------------------------------------

def sample(adj, inputs, num_samples_per_layer, col_perms):
    """num_samples_per_layer = [layer_infos[i].num_samples]; loop k uses
    t = K-1-k (:269), support_size *= num_samples[t] (:270), result flattened
    row-major (:273).  col_perms[k] is the permutation for the k-th sampler call.
    Returns (samples, support_sizes)."""
    K = len(num_samples_per_layer)
    samples = [np.asarray(inputs, dtype=np.int32)]
    support = 1
    support_sizes = [1]
    for k in range(K):
        t = K - k - 1
        support *= num_samples_per_layer[t]
        node = uniform_neighbor_sampler(adj, samples[k], num_samples_per_layer[t], col_perms[k])
        samples.append(node.reshape(-1))
        support_sizes.append(support)
    return samples, support_sizes


# --------------------------------------------------------------------------
# inits                                 inits.py:15-20, layers.py:94-99
# --------------------------------------------------------------------------

def glorot(shape, rng, dtype=np.float32):
    """U(-r, r), r = sqrt(6/(in+out))  (inits.py:15-20).  tf xavier_initializer
    (uniform=True default, layers.py:96) has the same range."""
    r = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-r, r, size=shape).astype(dtype)


# --------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------

# relu'(x) at a pre-activation that is zero up to fp32 summation noise is decided by that noise: TF/Eigen, this
# oracle and the device sum in different orders, so such entries can come out > 0 on one side and == 0 on the other.
# Like the sampler's permutations and the dropout masks, the tie is INJECTED: _RELU_TIES (set through
# `relu_ties_from`) maps an activation to the sign pattern to follow for entries with |pre-activation| <= _TIE_EPS.
_TIE_EPS = 2e-6
_relu_ties = None


class relu_ties_from(object):
    """with relu_ties_from(fn): fn(shape) -> bool array ("positive on the other side") or None, consulted by every
    relu in call order; only entries whose pre-activation is within _TIE_EPS of zero follow it."""

    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global _relu_ties
        _relu_ties = self.fn

    def __exit__(self, *exc):
        global _relu_ties
        _relu_ties = None


# reduce_max over neighbors whose two largest entries agree to summation noise: WHICH row is the arg-max is decided by
# the summation order of the MLP contraction (TF/Eigen, this oracle and the device all differ), and the backward pass
# routes d_pooled * x_row to that row's features -- a different row is a different (equally valid) gradient.  Like the
# relu ties, the choice is INJECTED: `argmax_ties_from(fn)`; fn(shape [n, hidden]) -> the other side's arg-max, consulted
# by every max-pool call in call order; an entry follows it only where that row's activation is within _ARGMAX_EPS of the
# maximum (so a genuinely wrong arg-max on the other side still shows up as a mismatch).
_ARGMAX_EPS = 1e-5
_argmax_ties = None


class argmax_ties_from(object):
    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global _argmax_ties
        _argmax_ties = self.fn

    def __exit__(self, *exc):
        global _argmax_ties
        _argmax_ties = None


def _act(x, act):
    if act == "relu":
        y = np.maximum(x, 0)
        if _relu_ties is not None:
            other = _relu_ties(x.shape)
            if other is not None:
                ties = np.abs(x) <= _TIE_EPS
                if ties.any():
                    # follow the other side on ties: a strictly positive marker where it is positive, exact 0 elsewhere
                    y = np.where(ties, np.where(other, np.asarray(np.finfo(x.dtype).tiny, x.dtype), 0), y).astype(x.dtype)
        return y
    return x


def _act_bwd(y, dy, act):
    if act == "relu":  # tf relu grad: dy * (y > 0)
        return dy * (y > 0)
    return dy


# --------------------------------------------------------------------------
# A2  MeanAggregator                    aggregators.py:43-64
# --------------------------------------------------------------------------

def mean_aggregator_fwd(self_vecs, neigh_vecs, W_self, W_neigh, concat, act, bias=None):
    """neigh_means = mean over axis 1 (:48); from_neighs = means @ W_neigh (:51);
    from_self = self @ W_self (:53); add (:56) or concat [self, neigh] (:58);
    optional bias (:61-62); act (:64).  Dropout is identity at p=0 (:46-47)."""
    means = neigh_vecs.mean(axis=1, dtype=neigh_vecs.dtype)
    from_neighs = means @ W_neigh
    from_self = self_vecs @ W_self
    if concat:
        out = np.concatenate([from_self, from_neighs], axis=1)
    else:
        out = from_self + from_neighs
    if bias is not None:
        out = out + bias
    y = _act(out, act)
    cache = (self_vecs, means, neigh_vecs.shape, y)
    return y, cache


def mean_aggregator_bwd(dy, cache, W_self, W_neigh, concat, act, has_bias=False):
    self_vecs, means, nshape, y = cache
    dout = _act_bwd(y, dy, act)
    o = W_self.shape[1]
    if concat:
        d_self_part, d_neigh_part = dout[:, :o], dout[:, o:]
    else:
        d_self_part = d_neigh_part = dout
    gW_self = self_vecs.T @ d_self_part
    gW_neigh = means.T @ d_neigh_part
    d_self = d_self_part @ W_self.T
    d_means = d_neigh_part @ W_neigh.T
    s = nshape[1]
    d_neigh = np.broadcast_to((d_means / s)[:, None, :], nshape).copy()
    grads = {"self_weights": gW_self, "neigh_weights": gW_neigh}
    if has_bias:
        grads["bias"] = dout.sum(axis=0)
    return d_self, d_neigh, grads


# --------------------------------------------------------------------------
# A3  GCNAggregator                     aggregators.py:101-116
# --------------------------------------------------------------------------

def gcn_aggregator_fwd(self_vecs, neigh_vecs, W, act, bias=None):
    """means = mean(concat([neigh, self[:,None,:]], axis=1), axis=1) (:106-107)
    = (sum_j neigh_j + self)/(s+1); out = means @ W (:110); act (:116).
    The `concat` ctor kwarg is stored but ignored (:79)."""
    s = neigh_vecs.shape[1]
    means = (neigh_vecs.sum(axis=1, dtype=neigh_vecs.dtype) + self_vecs) / np.asarray(s + 1, dtype=self_vecs.dtype)
    out = means @ W
    if bias is not None:
        out = out + bias
    y = _act(out, act)
    return y, (means, neigh_vecs.shape, y)


def gcn_aggregator_bwd(dy, cache, W, act, has_bias=False):
    means, nshape, y = cache
    dout = _act_bwd(y, dy, act)
    gW = means.T @ dout
    d_means = dout @ W.T
    s = nshape[1]
    d_self = d_means / (s + 1)
    d_neigh = np.broadcast_to(d_self[:, None, :], nshape).copy()
    grads = {"weights": gW}
    if has_bias:
        grads["bias"] = dout.sum(axis=0)
    return d_self, d_neigh, grads


# --------------------------------------------------------------------------
# A4  MaxPoolingAggregator              aggregators.py:168-195 + layers.py:104-116
# --------------------------------------------------------------------------

def maxpool_aggregator_fwd(self_vecs, neigh_vecs, W_mlp, b_mlp, W_self, W_neigh, concat, act,
                           bias=None, pool="max"):
    """h = reshape(neigh, [n*s, d]) (:176); h = relu(h @ W_mlp + b_mlp) (Dense,
    layers.py:104-116, act relu :147); reshape [n, s, hidden] (:180); reduce_max
    over axis 1 (:181) [reduce_mean for the meanpool variant, :259];
    from_neighs = h @ W_neigh (:183); from_self = self @ W_self (:184);
    add/concat [self, neigh] (:186-189); act (:195)."""
    n, s, d = neigh_vecs.shape
    h = neigh_vecs.reshape(n * s, d) @ W_mlp + b_mlp
    h = np.maximum(h, 0).reshape(n, s, -1)
    if pool == "max":
        arg = h.argmax(axis=1)
        if _argmax_ties is not None:
            other = _argmax_ties(arg.shape)
            if other is not None:
                other = np.asarray(other, dtype=np.int64).reshape(arg.shape)
                mx = np.take_along_axis(h, arg[:, None, :], axis=1)[:, 0, :]
                ho = np.take_along_axis(h, np.clip(other, 0, s - 1)[:, None, :], axis=1)[:, 0, :]
                near = (other >= 0) & (other < s) & (mx - ho <= _ARGMAX_EPS * np.maximum(1.0, np.abs(mx)))
                arg = np.where(near, other, arg)
        pooled = np.take_along_axis(h, arg[:, None, :], axis=1)[:, 0, :]
    else:
        arg = None
        pooled = h.mean(axis=1, dtype=h.dtype)
    from_neighs = pooled @ W_neigh
    from_self = self_vecs @ W_self
    if concat:
        out = np.concatenate([from_self, from_neighs], axis=1)
    else:
        out = from_self + from_neighs
    if bias is not None:
        out = out + bias
    y = _act(out, act)
    return y, (self_vecs, neigh_vecs, h, arg, pooled, y)


def maxpool_aggregator_bwd(dy, cache, W_mlp, W_self, W_neigh, concat, act, pool="max"):
    self_vecs, neigh_vecs, h, arg, pooled, y = cache
    n, s, d = neigh_vecs.shape
    dout = _act_bwd(y, dy, act)
    o = W_self.shape[1]
    if concat:
        d_self_part, d_neigh_part = dout[:, :o], dout[:, o:]
    else:
        d_self_part = d_neigh_part = dout
    gW_self = self_vecs.T @ d_self_part
    gW_neigh = pooled.T @ d_neigh_part
    d_self = d_self_part @ W_self.T
    d_pooled = d_neigh_part @ W_neigh.T
    dh = np.zeros_like(h)
    if pool == "max":
        np.put_along_axis(dh, arg[:, None, :], d_pooled[:, None, :], axis=1)
    else:
        dh[:] = (d_pooled / s)[:, None, :]
    dh = dh * (h > 0)
    dh2 = dh.reshape(n * s, -1)
    x2 = neigh_vecs.reshape(n * s, d)
    gW_mlp = x2.T @ dh2
    gb_mlp = dh2.sum(axis=0)
    d_neigh = (dh2 @ W_mlp.T).reshape(n, s, d)
    grads = {"self_weights": gW_self, "neigh_weights": gW_neigh,
             "mlp_weights": gW_mlp, "mlp_bias": gb_mlp}
    return d_self, d_neigh, grads


# --------------------------------------------------------------------------
# A0 + A1  SampleAndAggregate.aggregate  models.py:278-330
# --------------------------------------------------------------------------

def make_aggregator_params(aggregator_type, dims, concat, rng, model_size="small", dtype=np.float32):
    """One parameter dict per layer, shapes per models.py:303-315:
    input dim = dim_mult*dims[layer] with dim_mult = 2 if concat and layer != 0 (:305)."""
    K = len(dims) - 1
    params = []
    for layer in range(K):
        dim_mult = 2 if (concat and layer != 0) else 1
        din, dout = dim_mult * dims[layer], dims[layer + 1]
        if aggregator_type == "mean":
            p = {"neigh_weights": glorot((din, dout), rng, dtype),
                 "self_weights": glorot((din, dout), rng, dtype)}
        elif aggregator_type == "gcn":
            p = {"weights": glorot((din, dout), rng, dtype)}
        elif aggregator_type in ("maxpool", "meanpool"):
            hidden = 512 if model_size == "small" else 1024  # aggregators.py:139-142
            p = {"mlp_weights": glorot((din, hidden), rng, dtype),
                 "mlp_bias": np.zeros((hidden,), dtype),
                 "neigh_weights": glorot((hidden, dout), rng, dtype),
                 "self_weights": glorot((din, dout), rng, dtype)}
        else:
            raise Exception("Unknown aggregator: ", aggregator_type)
        params.append(p)
    return params


def _agg_fwd(aggregator_type, p, self_vecs, neigh_vecs, concat, act):
    if aggregator_type == "mean":
        return mean_aggregator_fwd(self_vecs, neigh_vecs, p["self_weights"], p["neigh_weights"], concat, act)
    if aggregator_type == "gcn":
        return gcn_aggregator_fwd(self_vecs, neigh_vecs, p["weights"], act)
    pool = "max" if aggregator_type == "maxpool" else "mean"
    return maxpool_aggregator_fwd(self_vecs, neigh_vecs, p["mlp_weights"], p["mlp_bias"],
                                  p["self_weights"], p["neigh_weights"], concat, act, pool=pool)


def _agg_bwd(aggregator_type, p, dy, cache, concat, act):
    if aggregator_type == "mean":
        return mean_aggregator_bwd(dy, cache, p["self_weights"], p["neigh_weights"], concat, act)
    if aggregator_type == "gcn":
        return gcn_aggregator_bwd(dy, cache, p["weights"], act)
    pool = "max" if aggregator_type == "maxpool" else "mean"
    return maxpool_aggregator_bwd(dy, cache, p["mlp_weights"], p["self_weights"], p["neigh_weights"],
                                  concat, act, pool=pool)


def aggregate_fwd(samples, features, dims, num_samples, support_sizes, batch_size, params,
                  aggregator_type="mean", concat=True, masks=None):
    """hidden[h] = features[samples[h]] for every hop (:299); for each layer, for
    each hop < K-layer, h = aggregator((hidden[hop], reshape(hidden[hop+1],
    [batch*support[hop], num_samples[K-1-hop], dim_mult*dims[layer]]))) (:321-328);
    the last layer has identity activation (:307-310).  Returns (out, tape).

    Dropout (tf.nn.dropout(x, 1 - p) = x * mask / (1 - p)) is INJECTED like the sampler's
    permutations: masks(layer, hop, role, n_rows, d) returns the already scaled mask of
    role "self" ([n, d]) / "neigh" ([n*s, d]) or None.  Mean/GCN drop both inputs
    (aggregators.py:46-47, :104-105); the pooling aggregators drop only the neighbor rows,
    inside their Dense MLP (layers.py:107), so their "self" mask is None."""
    K = len(num_samples)
    hidden = [features[np.asarray(s, dtype=np.int64)] for s in samples]
    tape = []
    mask_table = []
    for layer in range(K):
        act = "id" if layer == K - 1 else "relu"
        dim_mult = 2 if (concat and layer != 0) else 1
        next_hidden = []
        layer_tape = []
        layer_masks = []
        for hop in range(K - layer):
            neigh_dims = (batch_size * support_sizes[hop], num_samples[K - hop - 1], dim_mult * dims[layer])
            self_in, neigh_in = hidden[hop], hidden[hop + 1].reshape(neigh_dims)
            ms = mn = None
            if masks is not None:
                ms = masks(layer, hop, "self", neigh_dims[0], neigh_dims[2])
                mn = masks(layer, hop, "neigh", neigh_dims[0] * neigh_dims[1], neigh_dims[2])
                if ms is not None:
                    self_in = self_in * ms
                if mn is not None:
                    mn = mn.reshape(neigh_dims)
                    neigh_in = neigh_in * mn
            y, cache = _agg_fwd(aggregator_type, params[layer], self_in, neigh_in, concat, act)
            next_hidden.append(y)
            layer_tape.append(cache)
            layer_masks.append((ms, mn))
        tape.append(layer_tape)
        mask_table.append(layer_masks)
        hidden = next_hidden
    tape.append(mask_table)      # tape[K]: the injected dropout masks, tape[layer][hop] stays the aggregator cache
    return hidden[0], tape


def aggregate_bwd(d_out, tape, params, num_samples, aggregator_type="mean", concat=True, input_grads=False):
    """Reverse of aggregate_fwd.  No gradient flows into the fixed `features`
    (non-trainable Variable, models.py:238), so layer-0 input grads are dropped --
    unless input_grads=True (identity features, models.py:229-240: the leading
    columns of the table are the trainable `node_embeddings`), in which case
    (grads, [dL/d hidden[h] of layer 0 for every hop h]) is returned."""
    K = len(num_samples)
    grads = [None] * K
    d_hidden = [d_out]
    for layer in range(K - 1, -1, -1):
        act = "id" if layer == K - 1 else "relu"
        d_prev = [None] * (K - layer + 1)
        g_layer = None
        for hop in range(K - layer):
            ms, mn = tape[K][layer][hop] if len(tape) > K else (None, None)
            d_self, d_neigh, g = _agg_bwd(aggregator_type, params[layer], d_hidden[hop], tape[layer][hop], concat, act)
            if ms is not None:
                d_self = d_self * ms
            if mn is not None:
                d_neigh = d_neigh.reshape(mn.shape) * mn
            if g_layer is None:
                g_layer = {k: v.copy() for k, v in g.items()}
            else:
                for k in g:
                    g_layer[k] += g[k]
            if layer > 0 or input_grads:
                dn = d_neigh.reshape(-1, d_neigh.shape[-1])
                d_prev[hop] = d_self if d_prev[hop] is None else d_prev[hop] + d_self
                d_prev[hop + 1] = dn if d_prev[hop + 1] is None else d_prev[hop + 1] + dn
        grads[layer] = g_layer
        d_hidden = d_prev
    if input_grads:
        return grads, d_hidden
    return grads


def embedding_grad(d_hidden0, samples, n_rows, identity_dim):
    """Gradient w.r.t. `node_embeddings` [N+1, identity_dim] of
    features = concat([embeds, fixed], axis=1); hidden[h] = embedding_lookup(features, samples[h])
    (models.py:240, :299): the first identity_dim columns of every hop's input
    gradient, summed per looked-up id (IndexedSlices semantics)."""
    g = np.zeros((n_rows, identity_dim), dtype=d_hidden0[0].dtype)
    for dh, ids in zip(d_hidden0, samples):
        np.add.at(g, np.asarray(ids, dtype=np.int64), dh[:, :identity_dim])
    return g


# --------------------------------------------------------------------------
# H0  SupervisedGraphsage head / loss / optimizer   supervised_models.py:78-126
# --------------------------------------------------------------------------

def l2_normalize_fwd(x, eps=1e-12):
    """tf.nn.l2_normalize(x, 1) = x * rsqrt(max(sum(x^2), eps))  (:85)."""
    ss = (x * x).sum(axis=1, keepdims=True, dtype=x.dtype)
    inv = 1.0 / np.sqrt(np.maximum(ss, np.asarray(eps, dtype=x.dtype)))
    return x * inv, (x, inv, ss, eps)


def l2_normalize_bwd(dy, cache):
    x, inv, ss, eps = cache
    y = x * inv
    dot = (dy * y).sum(axis=1, keepdims=True, dtype=x.dtype)
    dx = inv * (dy - y * dot)
    # rows clamped by eps: y = x / sqrt(eps), derivative is dy * inv
    clamped = ss < eps
    return np.where(clamped, dy * inv, dx)


def softmax(x):
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=1, keepdims=True, dtype=x.dtype)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def classification_loss(logits, labels, sigmoid_loss):
    """reduce_mean(sigmoid_cross_entropy_with_logits) over all B*C entries (:112-114)
    or reduce_mean(softmax_cross_entropy_with_logits) over B rows (:116-118).
    Returns (loss, dlogits)."""
    B = logits.shape[0]
    if sigmoid_loss:
        x, z = logits, labels
        per = np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))
        loss = per.mean(dtype=logits.dtype)
        dlogits = (sigmoid(x) - z) / np.asarray(x.size, dtype=logits.dtype)
    else:
        m = logits.max(axis=1, keepdims=True)
        lse = m + np.log(np.exp(logits - m).sum(axis=1, keepdims=True, dtype=logits.dtype))
        logp = logits - lse
        per = -(labels * logp).sum(axis=1, dtype=logits.dtype)
        loss = per.mean(dtype=logits.dtype)
        # d/dlogits of -sum_c z_c log p_c = p * sum(z) - z
        zs = labels.sum(axis=1, keepdims=True, dtype=logits.dtype)
        dlogits = (np.exp(logp) * zs - labels) / np.asarray(B, dtype=logits.dtype)
    return loss, dlogits


def make_supervised_params(aggregator_type, dims, num_classes, concat, rng, model_size="small",
                           dtype=np.float32):
    params = {"agg": make_aggregator_params(aggregator_type, dims, concat, rng, model_size, dtype)}
    dim_mult = 2 if concat else 1
    params["node_pred"] = {"weights": glorot((dim_mult * dims[-1], num_classes), rng, dtype),  # layers.py:94-96
                           "bias": np.zeros((num_classes,), dtype)}                            # layers.py:98-99
    return params


def supervised_fwd_bwd(params, features, samples, support_sizes, labels, dims, num_samples, batch_size,
                       aggregator_type="mean", concat=True, sigmoid_loss=False, weight_decay=0.0,
                       want_grads=True, identity_dim=0, masks=None, head_mask=None):
    """SupervisedGraphsage.build/_loss/predict on INJECTED `samples`
    (supervised_models.py:78-126).  Returns dict with loss, preds, outputs1,
    node_preds and (optionally) clipped-free raw grads.  With identity_dim > 0,
    `features` is concat([node_embeddings, fixed features], axis=1)
    (supervised_models.py:49-60) and grads["embeds"] is returned (no weight decay:
    the embedding is in neither aggregator.vars nor node_pred.vars).  `masks` as in
    aggregate_fwd; `head_mask` [batch, dim] is the scaled dropout mask of the
    prediction Dense's input (layers.py:107)."""
    out, tape = aggregate_fwd(samples, features, dims, num_samples, support_sizes, batch_size,
                              params["agg"], aggregator_type, concat, masks=masks)
    out_n, ncache = l2_normalize_fwd(out)                                 # :85
    W, b = params["node_pred"]["weights"], params["node_pred"]["bias"]
    head_in = out_n if head_mask is None else out_n * head_mask
    logits = head_in @ W + b                                              # :88-92 (Dense, identity act)
    loss_c, dlogits = classification_loss(logits, labels, sigmoid_loss)   # :111-118
    dt = features.dtype
    wd = np.asarray(weight_decay, dtype=dt)
    reg = np.asarray(0, dtype=dt)
    for li, p in enumerate(params["agg"]):                                 # :104-106 (aggregator.vars only)
        for k in _decayed_keys(aggregator_type):
            reg = reg + wd * (p[k] * p[k]).sum(dtype=dt) / 2
    for k in ("weights", "bias"):                                         # :107-108
        v = params["node_pred"][k]
        reg = reg + wd * (v * v).sum(dtype=dt) / 2
    loss = loss_c + reg
    preds = sigmoid(logits) if sigmoid_loss else softmax(logits)          # :122-126
    res = {"loss": loss, "preds": preds, "outputs1": out_n, "node_preds": logits, "agg_out": out}
    if not want_grads:
        return res
    gW = head_in.T @ dlogits + wd * W
    gb = dlogits.sum(axis=0, dtype=dt) + wd * b
    d_out_n = dlogits @ W.T
    if head_mask is not None:
        d_out_n = d_out_n * head_mask
    d_out = l2_normalize_bwd(d_out_n, ncache)
    g_emb = None
    if identity_dim > 0:
        g_agg, d_hidden0 = aggregate_bwd(d_out, tape, params["agg"], num_samples, aggregator_type, concat,
                                         input_grads=True)
        g_emb = embedding_grad(d_hidden0, samples, features.shape[0], identity_dim)
    else:
        g_agg = aggregate_bwd(d_out, tape, params["agg"], num_samples, aggregator_type, concat)
    for li, p in enumerate(params["agg"]):
        for k in _decayed_keys(aggregator_type):
            g_agg[li][k] = g_agg[li][k] + wd * p[k]
    res["grads"] = {"agg": g_agg, "node_pred": {"weights": gW, "bias": gb}}
    if g_emb is not None:
        res["grads"]["embeds"] = g_emb
    return res


def _decayed_keys(aggregator_type):
    """Keys of aggregator.vars (what the weight-decay loop iterates).  MaxPool's
    MLP Dense weights live in mlp_layers[0].vars, not aggregator.vars
    (aggregators.py:144-159), so they get no weight decay."""
    if aggregator_type == "gcn":
        return ("weights",)
    return ("neigh_weights", "self_weights")


def clip_by_value(g, lo=-5.0, hi=5.0):
    """supervised_models.py:96 / models.py:380."""
    return np.clip(g, lo, hi)


def adam_tf_update(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (supervised_models.py:73, :99): epsilon is added
    OUTSIDE the bias-corrected sqrt:  lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    p -= lr_t * m / (sqrt(v) + eps).  t starts at 1.  Updates in place."""
    dt = p.dtype
    m *= dt.type(beta1); m += dt.type(1 - beta1) * g
    v *= dt.type(beta2); v += dt.type(1 - beta2) * (g * g)
    lr_t = dt.type(lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t))
    p -= lr_t * m / (np.sqrt(v) + dt.type(eps))


def flat_param_items(params, aggregator_type):
    """Deterministic (name, array) order shared with the device engine's flat
    parameter buffer: per layer, sorted keys; then node_pred weights, bias."""
    items = []
    for li, p in enumerate(params["agg"]):
        for k in sorted(p.keys()):
            items.append(("agg%d/%s" % (li, k), p[k]))
    items.append(("node_pred/weights", params["node_pred"]["weights"]))
    items.append(("node_pred/bias", params["node_pred"]["bias"]))
    return items


def calc_f1_micro(y_true, y_pred, sigmoid_loss):
    """supervised_train.py:63-70 (micro average only; implemented without sklearn
    so it can be checked against sklearn in tests)."""
    if not sigmoid_loss:
        t = np.argmax(y_true, axis=1)
        p = np.argmax(y_pred, axis=1)
        return float((t == p).mean())  # single-label micro-F1 == accuracy
    p = (y_pred > 0.5).astype(np.int64)
    t = (y_true > 0.5).astype(np.int64)
    tp = float((p & t).sum()); fp = float((p & (1 - t)).sum()); fn = float(((1 - p) & t).sum())
    return 0.0 if tp == 0 else 2 * tp / (2 * tp + fp + fn)


# --------------------------------------------------------------------------
# N3  unsupervised objective            models.py:332-405, prediction.py:68-110
# --------------------------------------------------------------------------

def _softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def linkpred_fwd_bwd(o1, o2, neg, neg_sample_weights=1.0):
    """BipartiteEdgePredLayer._xent_loss (prediction.py:102-110) with bilinear_weights=False:
    aff = sum(o1*o2, axis=1) (:79); neg_aff = o1 @ neg.T (:91); loss = sum(xent(1, aff)) + w*sum(xent(0, neg_aff)).
    MRR per models.py:393-405: aff_all = [neg_aff | aff]; the rank of the true pair (0-based, ties broken towards
    the negatives as tf.nn.top_k does: lower index first) is #{j: neg_aff_j >= aff}; mrr = mean(1/(rank+1)).
    Returns dict(loss (summed, NOT yet divided by batch_size), mrr, ranks, aff_all, d_o1, d_o2, d_neg)."""
    aff = (o1 * o2).sum(axis=1, dtype=o1.dtype)
    neg_aff = o1 @ neg.T
    loss = _softplus(-aff).sum(dtype=o1.dtype) + neg_sample_weights * _softplus(neg_aff).sum(dtype=o1.dtype)
    ranks = (neg_aff >= aff[:, None]).sum(axis=1)
    mrr = (1.0 / (ranks + 1)).mean()
    d_aff = sigmoid(aff) - 1.0
    d_neg_aff = neg_sample_weights * sigmoid(neg_aff)
    d_o1 = d_aff[:, None] * o2 + d_neg_aff @ neg
    d_o2 = d_aff[:, None] * o1
    d_neg = d_neg_aff.T @ o1
    return {"loss": loss, "mrr": mrr, "ranks": ranks, "aff_all": np.concatenate([neg_aff, aff[:, None]], axis=1),
            "d_o1": d_o1, "d_o2": d_o2, "d_neg": d_neg}


def unsupervised_fwd_bwd(params_agg, features, samples, support_sizes, dims, num_samples, batch_size, n_neg,
                         aggregator_type="mean", concat=True, weight_decay=0.0, neg_sample_weights=1.0, want_grads=True,
                         masks=None):
    """SampleAndAggregate._build/_loss/build (models.py:332-391) on INJECTED samples whose roots are
    [batch1 (B) | batch2 (B) | neg_samples (n_neg)]: the three aggregate() passes of :350-360 share the aggregators,
    and rows are independent, so one pass over the concatenated roots is the same computation.
    loss = (sum wd*l2_loss(aggregator vars) + xent) / batch_size   (:386-390, :378)."""
    n_roots = 2 * batch_size + n_neg
    out, tape = aggregate_fwd(samples, features, dims, num_samples, support_sizes, n_roots, params_agg,
                              aggregator_type, concat, masks=masks)       # masks: injected dropout, see aggregate_fwd
    out_n, ncache = l2_normalize_fwd(out)                                  # :368-370
    B = batch_size
    lp = linkpred_fwd_bwd(out_n[:B], out_n[B:2 * B], out_n[2 * B:], neg_sample_weights)
    dt = features.dtype
    wd = np.asarray(weight_decay, dtype=dt)
    reg = np.asarray(0, dtype=dt)
    for p in params_agg:
        for k in _decayed_keys(aggregator_type):
            reg = reg + wd * (p[k] * p[k]).sum(dtype=dt) / 2
    loss = (reg + lp["loss"]) / np.asarray(B, dtype=dt)
    res = {"loss": loss, "mrr": lp["mrr"], "ranks": lp["ranks"], "aff_all": lp["aff_all"], "outputs1": out_n[:B],
           "outputs_all": out_n}
    if not want_grads:
        return res
    d_out_n = np.concatenate([lp["d_o1"], lp["d_o2"], lp["d_neg"]], axis=0) / np.asarray(B, dtype=dt)
    d_out = l2_normalize_bwd(d_out_n, ncache)
    g_agg = aggregate_bwd(d_out, tape, params_agg, num_samples, aggregator_type, concat)
    for li, p in enumerate(params_agg):
        for k in _decayed_keys(aggregator_type):
            g_agg[li][k] = g_agg[li][k] + wd * p[k] / np.asarray(B, dtype=dt)
    res["grads"] = g_agg
    return res
