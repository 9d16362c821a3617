"""-m gpu: the HIP path (through the C ABI) == the REFERENCE'S OWN CODE.

tests/golden/ref_*.npz were computed by /root/reference/graphsage/*.py executed unmodified on the TF1 stand-in
(tests/golden/make_ref_fixtures.py); the reference's padded adjacency tables, initial weights, batches, labels, column
permutations and negatives are fed to the device model, and every step of the reference run is compared: sampled ids
bit-exact; loss, predictions, embeddings, every gradient and the parameters after clip + Adam within north_star's 1e-4
(fp32 run of the reference; gradients relative to the tensor's largest entry)."""
import numpy as np
import pytest
import torch

from graphsage_amd import engine as eng
from graphsage_amd import inits
from graphsage_amd.models import Placeholder, SAGEInfo, SampleAndAggregate
from graphsage_amd.neigh_samplers import AdjInfo, PaddedAdjacency, UniformNeighborSampler
from graphsage_amd.supervised_models import SupervisedGraphsage
from ref_fixtures import SUP, SUP_DROPOUT, UNSUP, Fixture, flat_items

pytestmark = pytest.mark.gpu
RTOL = 1e-4
# TF's Adam step is lr * g / (|g| + eps_hat) on the first step, eps_hat = 1e-8 / sqrt(1 - beta2) = 3.2e-7: inside that knee
# d(step)/dg = lr * eps_hat / (|g| + eps_hat)^2 reaches 1e4, so a gradient element of ~1e-7 that agrees with the reference to
# 2e-9 (far inside the gradient tolerance) moves its parameter by 4e-5.  The parameters-after-Adam comparison therefore takes
# the elements with |g| > 30 eps_hat; the gradients themselves are compared element by element without this exclusion.
ADAM_KNEE = 1e-5


def close(got, want, msg, atol_rel=1e-4):
    want = np.asarray(want)
    np.testing.assert_allclose(np.asarray(got).reshape(want.shape), want, rtol=RTOL,
                               atol=atol_rel * max(1e-2, float(np.abs(want).max())), err_msg=msg)


def model_variables(model, supervised=True):
    """name -> engine Variable in the fixture's naming."""
    out = {}
    for i, a in enumerate(model.aggregators):
        for k, v in a.vars.items():
            out["agg%d/%s" % (i, k)] = v
        for l in getattr(a, "mlp_layers", []):
            out["agg%d/mlp_weights" % i] = l.vars['weights']
            out["agg%d/mlp_bias" % i] = l.vars['bias']
    if supervised:
        out["node_pred/weights"] = model.node_pred.vars['weights']
        out["node_pred/bias"] = model.node_pred.vars['bias']
    if model.embeds is not None:
        out["embeds"] = model.embeds
    return out


def load_weights(model, fx, prefix, supervised=True):
    mv = model_variables(model, supervised)
    assert sorted(mv) == sorted(k[len(prefix):] for k in fx.z.files if k.startswith(prefix))
    for k, v in mv.items():
        v.assign(fx[prefix + k].astype(np.float32).reshape(v.numpy().shape))
    if model.embeds is not None:
        model._refresh_embeds()
    eng.get_engine().sync()
    return mv


def build_supervised(fx):
    c = fx.cfg
    eng.reset_engine()
    inits.set_seed(1)
    e = eng.get_engine()
    ph = {'labels': Placeholder('labels'), 'batch': Placeholder('batch1'), 'dropout': Placeholder('dropout', 0.),
          'batch_size': Placeholder('batch_size')}
    adj_info = AdjInfo(PaddedAdjacency(fx["graph/adj_train"], e.device))
    sampler = UniformNeighborSampler(adj_info)
    layer_infos = [SAGEInfo("node", sampler, s, fx.out_dim) for s in c["num_samples"]]
    model = SupervisedGraphsage(fx["graph/labels"].shape[1], ph, fx["graph/feats"], adj_info, fx["graph/deg"], layer_infos,
                                concat=c["concat"], aggregator_type=fx.agg, sigmoid_loss=c["sigmoid"],
                                learning_rate=c["learning_rate"], weight_decay=c["weight_decay"],
                                identity_dim=fx.identity_dim, model_size=c.get("model_size", "small"))
    model.use_graphs = False                      # the padded sampler takes a host permutation per call
    return e, ph, adj_info, sampler, model


def reference_dropout_masks(model, fx, p):
    """{device dropout site: uint8 keep bits [rows, d]} from the masks the reference run drew at step `p`, in the order its
    graph applies tf.nn.dropout (tests/test_ref_pin.py::_dropout_masks): per layer, per hop: neigh_vecs then self_vecs
    (aggregators.py:46-47, :104-105) or only the MLP input for the pooling aggregators (layers.py:107), last the
    prediction Dense (supervised_models.py:88-92).  The device runs all hops of a layer as ONE call, so a site's rows are
    the hops' rows in hop order (Engine.inject_dropout_masks)."""
    from graphsage_amd.layers import SITE_DENSE, SITE_MLP, SITE_NEIGH, SITE_SELF
    pooling = fx.agg in ("maxpool", "meanpool")
    j, sites = 0, {}
    for layer in range(fx.K):
        agg = model.aggregators[layer]
        neigh, selfs = [], []
        for hop in range(fx.K - layer):
            neigh.append(fx[p + "mask%d" % j]); j += 1
            if not pooling:
                selfs.append(fx[p + "mask%d" % j]); j += 1
        d = neigh[0].shape[-1]
        sites[agg.site + (SITE_MLP if pooling else SITE_NEIGH)] = np.concatenate([m.reshape(-1, d) for m in neigh])
        if selfs:
            sites[agg.site + SITE_SELF] = np.concatenate([m.reshape(-1, d) for m in selfs])
    head = fx[p + "mask%d" % j]
    assert not fx.has(p + "mask%d" % (j + 1))
    sites[model.node_pred.site + SITE_DENSE] = head.reshape(-1, head.shape[-1])
    return sites


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("name", SUP + SUP_DROPOUT)
def test_supervised_steps_equal_reference_run(dev, name, fuse):
    _supervised_steps(name, fuse)


def _pool_through_distinct_ids(e, model, form):
    """Send the max-pool MLP of layer 0 through the path the benched configuration takes -- the step's DISTINCT sampled ids
    (gs_unique_ids) + the split-MFMA kernel of `form` -- although the fixture gathers far fewer rows than the 2048 the default
    threshold asks for (GS_POOL_DEDUP_MIN_ROWS / aggregator.dedup_min_rows)."""
    e.pool_f16 = form == "f16x2"
    assert e.split_pool
    model.aggregators[0].dedup_min_rows = 0


def _check_pool_kernel(model, form):
    a0 = model.aggregators[0]
    assert a0.last_pool_kernel == {"f16x2": "split16", "bf16x3": "split_bf16x3"}[form], a0.last_pool_kernel
    cnt, rows_total = a0.last_unique
    assert 0 < int(cnt.item()) <= rows_total


@pytest.mark.parametrize("form", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("name", ["sup_maxpool", "sup_maxpool_big"])
def test_supervised_maxpool_distinct_id_mlp_equals_reference_run(dev, name, form):
    """The kernel that dominates the benched max-pool step -- the pooling MLP (aggregators.py:176-181, layers.py:104-116) on the
    step's distinct ids with two fp16 pieces per operand (gs_dense_fwd_rows_split16; `bf16x3`: three bf16 pieces,
    gs_dense_fwd_rows_split_ws) -- fed the REFERENCE's own run: same comparisons as the default path, at the same 1e-4."""
    _supervised_steps(name, True, prepare=lambda e, model: _pool_through_distinct_ids(e, model, form),
                      after_step=lambda model: _check_pool_kernel(model, form))


def _supervised_steps(name, fuse, prepare=None, after_step=None):
    fx = Fixture(name)
    c = fx.cfg
    e, ph, adj_info, sampler, model = build_supervised(fx)
    model.fuse_head = model.fuse_sampler = model.fuse_tail = fuse
    if prepare is not None:
        prepare(e, model)
    mv = load_weights(model, fx, "init/")
    for s in range(fx.n_steps):
        p = "s%d/" % s
        batch, labels = fx[p + "batch"], fx[p + "labels"]
        sampler.inject_perms(fx.perms(p, fx.K))
        feed = {ph['batch']: batch, ph['labels']: labels, ph['batch_size']: len(batch)}
        if c.get("dropout"):
            # the reference's tf.nn.dropout draws, injected like the permutations (gs_dropout.keep_bits)
            e.inject_dropout_masks(reference_dropout_masks(model, fx, p))
            feed[ph['dropout']] = c["dropout"]                            # supervised_train.py:269
        loss, preds = model.train_step(feed)
        if after_step is not None:
            after_step(model)
        if name in ("sup_mean_tail", "sup_gcn_tail"):
            assert bool(getattr(model, "_tail_used", False)) == fuse     # the headline step's fused-tail launch (+ its GCN form)
        for k in range(fx.K):                                           # S1/S2: bit-exact
            assert np.array_equal(model.samples1[k + 1].cpu().numpy(), fx[p + "sampled%d" % k].reshape(-1)), (s, k)
        close(loss, fx[p + "32/loss"], "loss step %d" % s)
        close(preds, fx[p + "32/preds"], "preds step %d" % s)
        close(model.outputs1.numpy(), fx[p + "32/outputs1"], "outputs1 step %d" % s)
        for k, v in mv.items():
            close(v.grad.numpy(), fx[p + "32/grad/" + k], "grad/%s step %d" % (k, s))
        for k, v in mv.items():
            want, g = fx[p + "32/after/" + k], fx[p + "32/grad/" + k]
            got = v.numpy().reshape(want.shape)
            # Adam's first steps move every entry by ~lr * sign(g): entries whose gradient is numerically zero on one
            # side (|g| below fp32 noise) are decided by that noise -- compare the rest
            solid = np.abs(g) > max(1e-6 * max(1e-2, np.abs(g).max()), ADAM_KNEE)
            np.testing.assert_allclose(got[solid], want[solid], rtol=RTOL, atol=2e-5, err_msg="after/%s step %d" % (k, s))
            dead = (g == 0) & (v.grad.numpy().reshape(g.shape) == 0)       # e.g. weights behind units that never fire:
            np.testing.assert_allclose(got[dead], want[dead], rtol=0, atol=1e-6,     # only Adam's decaying moments move them
                                       err_msg="after/%s step %d (zero gradient)" % (k, s))
        # continue from the reference's parameters (each step is pinned by itself; Adam moments stay the device's)
        for k, v in mv.items():
            v.assign(fx[p + "32/after/" + k].astype(np.float32).reshape(v.numpy().shape))
        if model.embeds is not None:
            model._refresh_embeds()
        e.sync()


@pytest.mark.parametrize("name", ["sup_mean", "sup_gcn"])
def test_evaluation_on_the_test_adjacency_equals_reference(dev, name):
    """supervised_train.py:280-285: tf.assign(adj_info, test_adj) -> AdjInfo.assign; forward only."""
    fx = Fixture(name)
    e, ph, adj_info, sampler, model = build_supervised(fx)
    load_weights(model, fx, "s%d/32/after/" % (fx.n_steps - 1))
    adj_info.assign(PaddedAdjacency(fx["graph/adj_test"], e.device))
    batch, labels = fx["eval/batch"], fx["eval/labels"]
    sampler.inject_perms(fx.perms("eval/", fx.K))
    loss, preds = model.eval_step({ph['batch']: batch, ph['labels']: labels, ph['batch_size']: len(batch)})
    for k in range(fx.K):
        assert np.array_equal(model.samples1[k + 1].cpu().numpy(), fx["eval/sampled%d" % k].reshape(-1))
    close(loss, fx["eval/32/loss"], "loss")
    close(preds, fx["eval/32/preds"], "preds")


@pytest.mark.parametrize("name", UNSUP)
def test_unsupervised_steps_equal_reference_run(dev, name):
    """models.py:332-405: three sample() calls with their OWN permutations (perm index g * K + k), the reference's
    negatives, loss / MRR / affinities / embeddings / gradients / parameters after Adam."""
    _unsupervised_steps(name)


@pytest.mark.parametrize("form", ["f16x2", "bf16x3"])
def test_unsupervised_maxpool_distinct_id_mlp_equals_reference_run(dev, form):
    """As test_supervised_maxpool_distinct_id_mlp_equals_reference_run, for the unsupervised max-pool model (models.py:332-405)."""
    _unsupervised_steps("unsup_maxpool", prepare=lambda e, model: _pool_through_distinct_ids(e, model, form),
                        after_step=lambda model: _check_pool_kernel(model, form))


def _unsupervised_steps(name, prepare=None, after_step=None):
    fx = Fixture(name)
    c = fx.cfg
    K, n_neg = fx.K, c["neg_sample_size"]
    eng.reset_engine()
    inits.set_seed(1)
    e = eng.get_engine()
    ph = {'batch1': Placeholder('batch1'), 'batch2': Placeholder('batch2'), 'neg_samples': Placeholder('neg'),
          'dropout': Placeholder('dropout', 0.), 'batch_size': Placeholder('batch_size')}
    adj_info = AdjInfo(PaddedAdjacency(fx["graph/adj_train"], e.device))
    sampler = UniformNeighborSampler(adj_info)
    layer_infos = [SAGEInfo("node", sampler, s, fx.out_dim) for s in c["num_samples"]]
    model = SampleAndAggregate(ph, fx["graph/feats"], adj_info, fx["graph/deg"], layer_infos, concat=c["concat"],
                               aggregator_type=fx.agg, learning_rate=c["learning_rate"], weight_decay=c["weight_decay"],
                               neg_sample_size=n_neg)
    model.use_graphs = False
    if prepare is not None:
        prepare(e, model)
    mv = load_weights(model, fx, "init/", supervised=False)
    for s in range(fx.n_steps):
        p = "s%d/" % s
        b1, b2, neg = fx[p + "batch1"], fx[p + "batch2"], fx[p + "neg_samples"]
        B = len(b1)
        sampler.inject_perms(fx.perms(p, 3 * K))
        model.inject_negatives(neg)
        loss, ranks, aff_all, mrr, outputs1 = model.train_step({ph['batch1']: b1, ph['batch2']: b2, ph['batch_size']: B})
        if after_step is not None:
            after_step(model)
        assert bool(getattr(model, "_lp_tail_used", False)) == (name == "unsup_mean_tail")     # the fused two-launch tail
        roots = np.concatenate([b1, b2, neg])
        assert np.array_equal(model.samples1[0].cpu().numpy(), roots)
        for k in range(K):
            want = np.concatenate([fx[p + "sampled%d" % (g * K + k)].reshape(-1) for g in range(3)])
            assert np.array_equal(model.samples1[k + 1].cpu().numpy(), want), (s, k)
        close(loss, fx[p + "32/loss"], "loss step %d" % s)
        close(aff_all, fx[p + "32/aff_all"], "aff_all")
        close(outputs1, fx[p + "32/outputs1"], "outputs1")
        full = model.outputs_all.numpy()
        close(full[B:2 * B], fx[p + "32/outputs2"], "outputs2")
        close(full[2 * B:2 * B + n_neg], fx[p + "32/neg_outputs"], "neg_outputs")
        ref_aff = fx[p + "32/aff_all"]
        margin = np.abs(ref_aff[:, :-1] - ref_aff[:, -1:]).min(axis=1) > 1e-4           # float near-ties aside
        assert np.array_equal(np.asarray(ranks)[margin], fx[p + "32/ranks"][:, -1][margin])
        if margin.all():
            close(mrr, fx[p + "32/mrr"], "mrr")
        for k, v in mv.items():
            close(v.grad.numpy(), fx[p + "32/grad/" + k], "grad/%s step %d" % (k, s))
        for k, v in mv.items():
            want, g = fx[p + "32/after/" + k], fx[p + "32/grad/" + k]
            solid = np.abs(g) > max(1e-6 * max(1e-2, np.abs(g).max()), ADAM_KNEE)
            np.testing.assert_allclose(v.numpy().reshape(want.shape)[solid], want[solid], rtol=RTOL, atol=2e-5,
                                       err_msg="after/%s step %d" % (k, s))
        for k, v in mv.items():
            v.assign(fx[p + "32/after/" + k].astype(np.float32).reshape(v.numpy().shape))
        e.sync()
