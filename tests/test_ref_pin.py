"""CPU suite: the oracle == the REFERENCE'S OWN CODE.

tests/golden/ref_*.npz hold what /root/reference/graphsage/{minibatch,neigh_samplers,models,aggregators,layers,
supervised_models,prediction}.py computed when executed unmodified on the TF1 stand-in (tests/golden/make_ref_fixtures.py).
Here every row of SURVEY 8a is checked against them:
  S0 construct_adj / construct_test_adj   bit-exact (same NumPy stream)
  S1/S2 sampler + sample schedule          bit-exact (permutations injected)
  A0-A4, H0, N3                            float64 twin at 1e-9, float32 at 1e-4: loss, predictions, embeddings, every
                                           gradient, parameters after clip + Adam for every step of the run.
"""
import numpy as np
import pytest

from oracle import graphsage_oracle as orc
from ref_fixtures import SUP, SUP_CPU, SUP_DROPOUT, UNSUP, UNSUP_CPU, Fixture, flat_items

TOL = {"32": dict(rtol=1e-4, atol=2e-6), "64": dict(rtol=1e-9, atol=1e-12)}
DT = {"32": np.float32, "64": np.float64}


def close(got, want, prec, msg=""):
    want = np.asarray(want)
    scale = max(1.0, float(np.abs(want).max())) if want.size else 1.0
    np.testing.assert_allclose(np.asarray(got).reshape(want.shape), want, rtol=TOL[prec]["rtol"],
                               atol=TOL[prec]["atol"] * scale, err_msg=msg)


# ---------------------------------------------------------------------------------------------------------------
# S0
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", SUP + SUP_CPU + SUP_DROPOUT + UNSUP + UNSUP_CPU)
def test_padded_adjacency_tables_equal_the_reference_iterators(name):
    """minibatch.py:227-259 (Node iterator) / :76-108 (Edge iterator): same global NumPy stream, same node order."""
    fx = Fixture(name)
    rng = np.random.RandomState(fx.cfg["np_seed"])
    if fx.cfg["kind"] == "unsupervised":
        rng.permutation(fx.n_nodes)                  # EdgeMinibatchIterator.__init__ shuffles G.nodes() first (:36)
    skip = fx["graph/val"] | fx["graph/test"]
    adj, deg = orc.construct_adj(fx.lists("train"), fx.cfg["max_degree"], rng, skip_mask=skip)
    test_adj, _ = orc.construct_adj(fx.lists("full"), fx.cfg["max_degree"], rng)
    assert np.array_equal(adj, fx["graph/adj_train"])
    assert np.array_equal(deg, fx["graph/deg"])
    assert np.array_equal(test_adj, fx["graph/adj_test"])
    # the edge cases the fixture graph was built to hold
    assert (deg[~skip] == 0).any() and (deg > fx.cfg["max_degree"]).any()
    assert (adj[:-1][skip] == fx.n_nodes).all() and (adj[-1] == fx.n_nodes).all()


# ---------------------------------------------------------------------------------------------------------------
# S1 + operators
# ---------------------------------------------------------------------------------------------------------------
def test_sampler_operator_equals_reference_call():
    fx = Fixture("operators")
    adj, ids = fx["op/sampler/adj"], fx["op/sampler/ids"]
    for num in (1, 4, 6):                           # 6 == max_degree
        got = orc.uniform_neighbor_sampler(adj, ids, num, fx["op/sampler/perm_%d" % num])
        assert np.array_equal(got, fx["op/sampler/out_%d" % num])


AGG_OPS = [("mean_concat", "mean", True, "relu"), ("mean_add", "mean", False, "relu"), ("mean_id", "mean", True, "id"),
           ("gcn", "gcn", False, "relu"), ("maxpool", "maxpool", True, "relu"), ("meanpool", "meanpool", False, "relu")]


@pytest.mark.parametrize("prec", ["32", "64"])
@pytest.mark.parametrize("tag,agg,concat,act", AGG_OPS)
def test_aggregator_operators_equal_reference_calls(tag, agg, concat, act, prec):
    """aggregator((self_vecs, neigh_vecs)) for the four aggregators (aggregators.py:43-64, 101-116, 168-195, 246-273),
    forward value and the weight gradients of sum(y * dy)."""
    fx, dt = Fixture("operators"), DT[prec]
    sv, nv = fx["op/self_vecs"].astype(dt), fx["op/neigh_vecs"].astype(dt)
    p = {k.split("/")[-1]: fx[k].astype(dt) for k in fx.z.files if k.startswith("op/%s/var/" % tag)}
    dy = fx["op/%s/dy" % tag].astype(dt)
    y, cache = orc._agg_fwd(agg, p, sv, nv, concat, act)
    close(y, fx["op/%s/%s/y" % (tag, prec)], prec)
    _, _, g = orc._agg_bwd(agg, p, dy, cache, concat, act)
    for k in p:
        close(g[k], fx["op/%s/%s/grad/%s" % (tag, prec, k)], prec, k)


# ---------------------------------------------------------------------------------------------------------------
# supervised model: S1/S2 + A0-A4 + H0, every step of the reference run
# ---------------------------------------------------------------------------------------------------------------
def _adam_chain_check(fx, p, prec, names, before, m, v, t, lr):
    """clip (supervised_models.py:96 / models.py:380) + tf.train.AdamOptimizer on the REFERENCE's gradients must give the
    reference's parameters; m, v are carried over the steps of the run."""
    dt = DT[prec]
    for k in names:
        g = orc.clip_by_value(fx[p + prec + "/grad/" + k].astype(dt))
        w = before[k].copy()
        orc.adam_tf_update(w, g.reshape(w.shape), m[k], v[k], t, lr)
        close(w, fx[p + prec + "/after/" + k], prec, "after/" + k)


def _dropout_masks(fx, p, dt):
    """The masks the reference run drew, in the order its graph applies tf.nn.dropout: per layer, per hop: neigh_vecs then
    self_vecs (aggregators.py:46-47, :104-105) or only the MLP input for the pooling aggregators (layers.py:107), and
    last the prediction Dense (supervised_models.py:88-92)."""
    order = []
    for layer in range(fx.K):
        for hop in range(fx.K - layer):
            order.append((layer, hop, "neigh"))
            if fx.agg in ("mean", "gcn"):
                order.append((layer, hop, "self"))
    keep = dt(1.0 - fx.cfg["dropout"])                  # tf.nn.dropout(x, 1 - dropout): x / keep * bits
    table = {key: fx[p + "mask%d" % j].astype(dt) / keep for j, key in enumerate(order)}
    head = fx[p + "mask%d" % len(order)].astype(dt) / keep
    assert not fx.has(p + "mask%d" % (len(order) + 1))

    def masks(layer, hop, role, n, d):
        mk = table.get((layer, hop, role))
        return None if mk is None else mk.reshape(n, d)
    return masks, head


@pytest.mark.parametrize("prec", ["32", "64"])
@pytest.mark.parametrize("name", SUP + SUP_CPU + SUP_DROPOUT)
def test_supervised_steps_equal_reference_run(name, prec):
    fx, dt = Fixture(name), DT[prec]
    c = fx.cfg
    ns, K = c["num_samples"], fx.K
    feats = fx["graph/feats"].astype(dt)
    adj = fx["graph/adj_train"]
    params = fx.params("init/", dt)
    names = [k for k, _ in flat_items(params)]
    m = {k: np.zeros_like(a) for k, a in flat_items(params)}
    v = {k: np.zeros_like(a) for k, a in flat_items(params)}
    for s in range(fx.n_steps):
        p = "s%d/" % s
        batch, labels = fx[p + "batch"], fx[p + "labels"].astype(dt)
        samples, support = orc.sample(adj, batch, ns, fx.perms(p, K))
        for k in range(K):                                          # S1/S2 bit-exact
            assert np.array_equal(samples[k + 1], fx[p + "sampled%d" % k].reshape(-1)), (s, k)
        features = np.concatenate([params["embeds"], feats], axis=1) if fx.identity_dim else feats
        masks, head_mask = _dropout_masks(fx, p, dt) if c.get("dropout") else (None, None)
        res = orc.supervised_fwd_bwd(params, features, samples, support, labels, fx.dims, ns, len(batch), fx.agg,
                                     c["concat"], c["sigmoid"], weight_decay=c["weight_decay"],
                                     identity_dim=fx.identity_dim, masks=masks, head_mask=head_mask)
        close(res["loss"], fx[p + prec + "/loss"], prec, "loss")
        close(res["preds"], fx[p + prec + "/preds"], prec, "preds")
        close(res["outputs1"], fx[p + prec + "/outputs1"], prec, "outputs1")
        close(res["node_preds"], fx[p + prec + "/node_preds"], prec, "node_preds")
        for k, g in flat_items(res["grads"]):
            close(g, fx[p + prec + "/grad/" + k], prec, "grad/" + k)
        before = dict(flat_items(params))
        _adam_chain_check(fx, p, prec, names, before, m, v, s + 1, c["learning_rate"])
        params = fx.params(p + prec + "/after/", dt)                # next step starts from the reference's parameters
    assert fx.n_steps >= 1


def test_reference_epoch_has_a_short_last_batch():
    fx = Fixture("sup_mean")
    sizes = [len(fx["s%d/batch" % s]) for s in range(fx.n_steps)]
    assert sizes[-1] < sizes[0] == fx.cfg["batch_size"] and sum(sizes) == len(fx["graph/train_nodes"])


@pytest.mark.parametrize("prec", ["32", "64"])
@pytest.mark.parametrize("name", ["sup_mean", "sup_gcn"])
def test_evaluation_on_the_test_adjacency_equals_reference(name, prec):
    """supervised_train.py:280-285: tf.assign(adj_info, test_adj), forward only, parameters = after the last step."""
    fx, dt = Fixture(name), DT[prec]
    c = fx.cfg
    params = fx.params("s%d/%s/after/" % (fx.n_steps - 1, prec), dt)
    batch, labels = fx["eval/batch"], fx["eval/labels"].astype(dt)
    samples, support = orc.sample(fx["graph/adj_test"], batch, c["num_samples"], fx.perms("eval/", fx.K))
    for k in range(fx.K):
        assert np.array_equal(samples[k + 1], fx["eval/sampled%d" % k].reshape(-1))
    assert (fx["graph/val"] | fx["graph/test"])[batch].any()          # val/test roots have neighbors only here
    res = orc.supervised_fwd_bwd(params, fx["graph/feats"].astype(dt), samples, support, labels, fx.dims,
                                 c["num_samples"], len(batch), fx.agg, c["concat"], c["sigmoid"],
                                 weight_decay=c["weight_decay"], want_grads=False)
    close(res["loss"], fx["eval/%s/loss" % prec], prec)
    close(res["preds"], fx["eval/%s/preds" % prec], prec)


# ---------------------------------------------------------------------------------------------------------------
# unsupervised model (models.py:332-405, prediction.py:68-110)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["32", "64"])
@pytest.mark.parametrize("name", UNSUP + UNSUP_CPU)
def test_unsupervised_steps_equal_reference_run(name, prec):
    fx, dt = Fixture(name), DT[prec]
    c = fx.cfg
    ns, K, n_neg = c["num_samples"], fx.K, c["neg_sample_size"]
    feats, adj = fx["graph/feats"].astype(dt), fx["graph/adj_train"]
    params = fx.params("init/", dt, supervised=False)
    names = [k for k, _ in flat_items(params)]
    m = {k: np.zeros_like(a) for k, a in flat_items(params)}
    v = {k: np.zeros_like(a) for k, a in flat_items(params)}
    for s in range(fx.n_steps):
        p = "s%d/" % s
        roots = [fx[p + "batch1"], fx[p + "batch2"], fx[p + "neg_samples"]]
        B = len(roots[0])
        per_group = []
        for gi, r in enumerate(roots):                # three sample() calls, each with its OWN permutations (:347-357)
            perms = [fx[p + "perm%d" % (gi * K + k)] for k in range(K)]
            smp, support = orc.sample(adj, r, ns, perms)
            for k in range(K):
                assert np.array_equal(smp[k + 1], fx[p + "sampled%d" % (gi * K + k)].reshape(-1)), (s, gi, k)
            per_group.append(smp)
        samples = [np.concatenate([g[h] for g in per_group]) for h in range(K + 1)]
        res = orc.unsupervised_fwd_bwd(params["agg"], feats, samples, support, fx.dims, ns, B, n_neg, fx.agg, c["concat"],
                                       weight_decay=c["weight_decay"])
        close(res["loss"], fx[p + prec + "/loss"], prec, "loss")
        close(res["mrr"], fx[p + prec + "/mrr"], prec, "mrr")
        close(res["aff_all"], fx[p + prec + "/aff_all"], prec, "aff_all")
        close(res["outputs_all"][:B], fx[p + prec + "/outputs1"], prec)
        close(res["outputs_all"][B:2 * B], fx[p + prec + "/outputs2"], prec)
        close(res["outputs_all"][2 * B:], fx[p + prec + "/neg_outputs"], prec)
        # models.py:402-403: ranks[:, -1] is the rank of the true pair; the oracle reports exactly that column
        assert np.array_equal(res["ranks"], fx[p + prec + "/ranks"][:, -1]), "rank of the true pair"
        for k, g in flat_items({"agg": res["grads"]}):
            close(g, fx[p + prec + "/grad/" + k], prec, "grad/" + k)
        before = dict(flat_items(params))
        _adam_chain_check(fx, p, prec, names, before, m, v, s + 1, c["learning_rate"])
        params = fx.params(p + prec + "/after/", dt, supervised=False)
    assert fx.n_steps >= 2
    assert len(fx["s%d/batch1" % (fx.n_steps - 1)]) < c["batch_size"]      # short last batch


def test_negatives_follow_the_degree_law_support():
    """fixed_unigram_candidate_sampler(unigrams=degrees): a degree-0 node is never drawn (models.py:336-343)."""
    fx = Fixture("unsup_mean")
    deg = fx["graph/deg"]
    for s in range(fx.n_steps):
        assert (deg[fx["s%d/neg_samples" % s]] > 0).all()


# ---------------------------------------------------------------------------------------------------------------
# the TIMED CPU baseline (oracle/cpu_baseline.py, bench.py's cpu_baseline leg) is the same computation
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["sup_mean", "sup_mean_3layer", "sup_mean_add_sigmoid", "sup_mean_tail"])
def test_cpu_baseline_port_steps_equal_reference_run(name):
    """The torch-CPU port whose step time bench.py reports as `cpu_baseline` trains exactly like the reference run:
    sampled ids, loss, logits and the parameters after every clip + Adam step (its own Adam, moments carried)."""
    import torch
    from oracle.cpu_baseline import CpuSupervisedMean
    fx = Fixture(name)
    c = fx.cfg
    port = CpuSupervisedMean(fx["graph/feats"], fx["graph/adj_train"], fx.dims, fx["graph/labels"].shape[1], c["num_samples"],
                             concat=c["concat"], sigmoid_loss=c["sigmoid"], lr=c["learning_rate"],
                             weight_decay=c["weight_decay"], threads=1)
    port.set_params_from_oracle(fx.params("init/", np.float32))
    for s in range(fx.n_steps):
        p = "s%d/" % s
        batch, labels = fx[p + "batch"], fx[p + "labels"]
        samples, _ = port.sample(batch, fx.perms(p, fx.K))
        for k in range(fx.K):
            assert np.array_equal(samples[k + 1].numpy(), fx[p + "sampled%d" % k].reshape(-1))
        loss, logits, _ = port.train_step(batch, labels, fx.perms(p, fx.K))
        close(loss, fx[p + "32/loss"], "32", "loss")
        close(logits, fx[p + "32/node_preds"], "32", "logits")
        after = fx.params(p + "32/after/", np.float32)
        for (wn, ws), a in zip(port.params, after["agg"]):
            np.testing.assert_allclose(wn.detach().numpy(), a["neigh_weights"], rtol=1e-4, atol=2e-5)
            np.testing.assert_allclose(ws.detach().numpy(), a["self_weights"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(port.W.detach().numpy(), after["node_pred"]["weights"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(port.b.detach().numpy(), after["node_pred"]["bias"], rtol=1e-4, atol=2e-5)
