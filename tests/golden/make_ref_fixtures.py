"""Generate tests/golden/ref_*.npz by EXECUTING THE REFERENCE'S OWN SOURCE.

    python tests/golden/make_ref_fixtures.py            # needs /root/reference (or $GRAPHSAGE_REFERENCE)

`/root/reference/graphsage/{inits,layers,neigh_samplers,aggregators,prediction,metrics,models,supervised_models,
minibatch}.py` are imported UNMODIFIED (nothing is copied into this repo); `import tensorflow` inside them resolves to
`tests/tf1_shim/tensorflow` (a torch-backed eager stand-in for the TF 1.x graph API -- see its docstring for what it
restates of TensorFlow).  Every number in the fixtures is therefore produced by the reference's code paths:

  S0  NodeMinibatchIterator / EdgeMinibatchIterator .construct_adj / .construct_test_adj   (minibatch.py:76-108, 227-259)
  S1  UniformNeighborSampler._call                                                          (neigh_samplers.py:24-29)
  S2  SampleAndAggregate.sample                                                             (models.py:254-275)
  A0-A4  SampleAndAggregate.aggregate + Mean/GCN/MaxPooling/MeanPooling aggregators         (models.py:278-330, aggregators.py)
  H0  SupervisedGraphsage.build/_loss/predict, clip, AdamOptimizer                          (supervised_models.py:78-126)
  N3  SampleAndAggregate._build/_loss/_accuracy/build (unsupervised), BipartiteEdgePredLayer (models.py:332-405, prediction.py)

on a small seeded graph that holds the edge cases of SURVEY 8c: degree-0 nodes, nodes with more neighbors than
max_degree, val/test nodes (all-pad rows under the train adjacency), `train_removed` edges, a short last batch,
`num_samples == max_degree`, evaluation after `tf.assign(adj_info, test_adj)`.

Each case runs twice -- `tf.float32` computing in float32 (the reference's precision) and in float64 (a twin that pins the
algebra to 1e-9) -- from the same seeds, so initial weights, permutations, negatives and batches are identical.
TF's random streams cannot be reproduced, so what the run drew (column permutations, negatives) is stored in the fixture
and INJECTED into the oracle / HIP path by the tests, exactly like the existing parity tests do.

The reference needs a networkx-1.x graph object only through `G.nodes() / G.node[n] / G.neighbors(n) / G[u][v] / G.edges()`;
`RefGraph` below provides those five accessors (the data container, no algorithm).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GRAPHSAGE_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(HERE, "..", "tf1_shim"), REF]

import tensorflow as tf  # noqa: E402  (the shim)

assert tf.__version__.endswith("-shim")
flags = tf.app.flags
FLAGS = flags.FLAGS
# the flags the model modules read (defined by the reference's train scripts: supervised_train.py:31-46,
# unsupervised_train.py:26-44)
flags.DEFINE_float('learning_rate', 0.01, '')
flags.DEFINE_float('weight_decay', 0.0, '')
flags.DEFINE_integer('neg_sample_size', 20, '')

import graphsage.layers as ref_layers  # noqa: E402
from graphsage.minibatch import EdgeMinibatchIterator, NodeMinibatchIterator  # noqa: E402
from graphsage.models import SAGEInfo, SampleAndAggregate  # noqa: E402
from graphsage.neigh_samplers import UniformNeighborSampler  # noqa: E402
from graphsage.supervised_models import SupervisedGraphsage  # noqa: E402

assert os.path.realpath(ref_layers.__file__).startswith(os.path.realpath(REF)), ref_layers.__file__


# ----------------------------------------------------------------------------------------------------------------
# the graph
# ----------------------------------------------------------------------------------------------------------------
class RefGraph(object):
    """networkx-1.x accessors over plain dicts (minibatch.py uses nothing else)."""

    def __init__(self, n, edges, val, test):
        self.node = {i: {'val': bool(val[i]), 'test': bool(test[i])} for i in range(n)}
        self.adj = {i: {} for i in range(n)}
        self._edges = []
        for u, v in edges:
            if v in self.adj[u]:
                continue
            attr = {'train_removed': bool(val[u] or test[u] or val[v] or test[v])}
            self.adj[u][v] = attr
            self.adj[v][u] = attr
            self._edges.append((u, v))

    def nodes(self):
        return list(self.node.keys())

    def neighbors(self, n):
        return list(self.adj[n].keys())

    def __getitem__(self, n):
        return self.adj[n]

    def edges(self):
        return list(self._edges)


def make_graph(seed=11, n=72, feat_dim=12, num_classes=5):
    rng = np.random.RandomState(seed)
    val = np.zeros(n, bool)
    test = np.zeros(n, bool)
    val[rng.choice(n, 8, replace=False)] = True
    rest = np.flatnonzero(~val)
    test[rng.choice(rest, 9, replace=False)] = True
    isolated = set(rng.choice(np.flatnonzero(~val & ~test), 3, replace=False).tolist())     # degree-0 train nodes
    hubs = rng.choice([i for i in range(n) if i not in isolated], 4, replace=False)          # deg > max_degree
    edges = []
    for h in hubs:
        for v in rng.choice(n, 22, replace=False):
            if v != h and v not in isolated:
                edges.append((int(h), int(v)))
    for _ in range(150):
        u, v = rng.randint(0, n, 2)
        if u != v and u not in isolated and v not in isolated:
            edges.append((int(u), int(v)))
    G = RefGraph(n, edges, val, test)
    feats = np.round(rng.randn(n, feat_dim) * 4) / 4                       # multiples of 1/4: exact in fp32
    feats = feats.astype(np.float32)
    single = rng.randint(0, num_classes, n)
    multi = (rng.rand(n, num_classes) < 0.4).astype(np.int64)
    return G, feats, single, multi


def csr_of(lists):
    rowptr = np.zeros(len(lists) + 1, np.int64)
    rowptr[1:] = np.cumsum([len(x) for x in lists])
    col = np.asarray([v for x in lists for v in x], np.int32)
    return rowptr, col


# ----------------------------------------------------------------------------------------------------------------
# helpers around the reference's objects
# ----------------------------------------------------------------------------------------------------------------
class RecordingSampler(object):
    """SAGEInfo.neigh_sampler is any callable (models.py:271-272); this one forwards to the reference's
    UniformNeighborSampler and keeps the output tensors so they can be fetched in the same Session.run."""

    def __init__(self, sampler):
        self.sampler, self.calls = sampler, []

    def __call__(self, inputs):
        out = self.sampler(inputs)
        self.calls.append((inputs[1], out))
        return out


def shuffle_node_of(t):
    stack, seen = [t], set()
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        if n.op_type == "random_shuffle":
            return n
        stack.extend(n.inputs)
    raise RuntimeError("no random_shuffle upstream")


def named_variables(model, supervised):
    out = {}
    for i, a in enumerate(model.aggregators):
        for k, v in a.vars.items():
            out["agg%d/%s" % (i, k)] = v
        mlp = getattr(a, "mlp_layers", [])
        assert len(mlp) <= 1
        for l in mlp:
            out["agg%d/mlp_weights" % i] = l.vars['weights']
            out["agg%d/mlp_bias" % i] = l.vars['bias']
    if supervised:
        out["node_pred/weights"] = model.node_pred.vars['weights']
        out["node_pred/bias"] = model.node_pred.vars['bias']
    if model.embeds is not None:
        out["embeds"] = model.embeds
    assert set(map(id, out.values())) == set(map(id, tf.trainable_variables())), "unnamed trainable variable"
    return out


def fresh(seed, real):
    tf.reset_default_graph()
    tf.shim.set_real(real)
    tf.set_random_seed(seed)
    ref_layers._LAYER_UIDS.clear()


def graph_arrays(G, feats, it, out):
    n = len(G.node)
    full = [G.neighbors(i) for i in range(n)]
    trainl = [[v for v in G.neighbors(i) if not G[i][v]['train_removed']] for i in range(n)]
    out["graph/full_rowptr"], out["graph/full_col"] = csr_of(full)
    out["graph/train_rowptr"], out["graph/train_col"] = csr_of(trainl)
    out["graph/val"] = np.asarray([G.node[i]['val'] for i in range(n)])
    out["graph/test"] = np.asarray([G.node[i]['test'] for i in range(n)])
    out["graph/adj_train"] = it.adj.astype(np.int32)
    out["graph/adj_test"] = it.test_adj.astype(np.int32)
    out["graph/deg"] = it.deg.astype(np.int64)
    out["graph/feats"] = np.vstack([feats, np.zeros((1, feats.shape[1]), np.float32)])   # supervised_train.py:133-135


# ----------------------------------------------------------------------------------------------------------------
# supervised cases
# ----------------------------------------------------------------------------------------------------------------
def run_supervised(cfg, real, out):
    G, feats, single, multi = make_graph()
    n = len(G.node)
    sigmoid = cfg["sigmoid"]
    C = multi.shape[1]
    id_map = {i: i for i in range(n)}
    class_map = {i: (multi[i].tolist() if sigmoid else int(single[i])) for i in range(n)}
    fresh(cfg["seed"], real)
    FLAGS.weight_decay = cfg["weight_decay"]
    FLAGS.learning_rate = cfg["learning_rate"]
    # supervised_train.py:112-120
    placeholders = {
        'labels': tf.placeholder(tf.float32, shape=(None, C), name='labels'),
        'batch': tf.placeholder(tf.int32, shape=(None), name='batch1'),
        'dropout': tf.placeholder_with_default(0., shape=(), name='dropout'),
        'batch_size': tf.placeholder(tf.int32, name='batch_size'),
    }
    drop_nodes = []
    orig_dropout = tf.nn.dropout

    def recording_dropout(x, keep_prob, **kw):                 # call order = aggregators.py:46-47 / layers.py:107 order
        node = orig_dropout(x, keep_prob, **kw)
        drop_nodes.append(node)
        return node
    tf.nn.dropout = recording_dropout
    np.random.seed(cfg["np_seed"])           # minibatch.py draws the padded tables from the global NumPy stream
    it = NodeMinibatchIterator(G, id_map, placeholders, class_map, C, batch_size=cfg["batch_size"],
                               max_degree=cfg["max_degree"])
    # supervised_train.py:147-148
    adj_info_ph = tf.placeholder(tf.int32, shape=it.adj.shape)
    adj_info = tf.Variable(adj_info_ph, trainable=False, name="adj_info")
    sampler = RecordingSampler(UniformNeighborSampler(adj_info))
    od = cfg["dim"] * (2 if cfg["aggregator_type"] == "gcn" else 1)        # supervised_train.py:175-176
    layer_infos = [SAGEInfo("node", sampler, s, od) for s in cfg["num_samples"]]
    features = np.vstack([feats, np.zeros((feats.shape[1],))]) if cfg.get("use_features", True) else None
    model = SupervisedGraphsage(C, placeholders, features, adj_info, it.deg, layer_infos=layer_infos,
                                aggregator_type=cfg["aggregator_type"], model_size=cfg.get("model_size", "small"), sigmoid_loss=sigmoid,
                                concat=cfg["concat"], identity_dim=cfg.get("identity_dim", 0), logging=False)
    tf.nn.dropout = orig_dropout
    nv = named_variables(model, True)
    names = sorted(nv)
    grads = tf.gradients(model.loss, [nv[k] for k in names])
    sess = tf.Session()
    sess.run(tf.global_variables_initializer(), feed_dict={adj_info_ph: it.adj})
    pre = real[-2:]
    if real == "float32":
        graph_arrays(G, feats, it, out)
        out["graph/labels"] = np.vstack([it._make_label_vec(i) for i in range(n)]).astype(np.float32)
        out["graph/train_nodes"] = np.asarray(it.train_nodes, np.int32)
        for k in names:
            out["init/" + k] = sess.run(nv[k]).astype(np.float32)
    # ---- batches: explicit node lists through the reference's feed constructor, or the reference's own epoch
    it.shuffle()
    feeds = []
    if cfg["batches"] == "epoch":
        while not it.end():
            feeds.append(it.next_minibatch_feed_dict())
    else:
        for nodes in cfg["batches"]:
            feeds.append(it.batch_feed_dict(nodes))
    K = len(layer_infos)
    for s, (feed, labels) in enumerate(feeds):
        fetches = [model.opt_op, model.loss, model.preds, model.outputs1, model.node_preds]
        fetches += [t for _, t in sampler.calls] + grads
        log0, dlog0 = len(tf.shim.log["shuffle"]), len(tf.shim.log["dropout"])
        if cfg.get("dropout"):
            feed.update({placeholders['dropout']: cfg["dropout"]})        # supervised_train.py:269
        res = sess.run(fetches, feed_dict=feed)
        perms = dict(tf.shim.log["shuffle"][log0:])
        masks = dict(tf.shim.log["dropout"][dlog0:])
        p = "s%d/" % s
        for j, dn in enumerate(drop_nodes if cfg.get("dropout") else []):
            if real == "float32":
                out[p + "mask%d" % j] = masks[id(dn)]                      # keep bits (uint8); scale is 1/(1-dropout)
            else:
                assert np.array_equal(out[p + "mask%d" % j], masks[id(dn)])
        if real == "float32":
            out[p + "batch"] = np.asarray(feed[placeholders['batch']], np.int32)
            out[p + "labels"] = np.asarray(labels, np.float32)
            for k, (ns, t) in enumerate(sampler.calls):
                out[p + "perm%d" % k] = perms[id(shuffle_node_of(t))].astype(np.int32)
                out[p + "sampled%d" % k] = res[5 + k].astype(np.int32)
        else:
            for k, (ns, t) in enumerate(sampler.calls):     # the twin run drew the same permutations
                assert np.array_equal(out[p + "perm%d" % k], perms[id(shuffle_node_of(t))])
        out[p + pre + "/loss"] = np.asarray(res[1])
        out[p + pre + "/preds"] = res[2]
        out[p + pre + "/outputs1"] = res[3]
        out[p + pre + "/node_preds"] = res[4]
        for k, g in zip(names, res[5 + K:]):
            out[p + pre + "/grad/" + k] = g
        for k in names:
            out[p + pre + "/after/" + k] = sess.run(nv[k])
    # ---- evaluation on the test adjacency (supervised_train.py:280-285: tf.assign(adj_info, minibatch.test_adj))
    if cfg.get("eval_nodes"):
        val_adj_info = tf.assign(adj_info, it.test_adj)
        train_adj_info = tf.assign(adj_info, it.adj)
        sess.run(val_adj_info)
        feed, labels = it.batch_feed_dict(cfg["eval_nodes"])
        log0 = len(tf.shim.log["shuffle"])
        res = sess.run([model.loss, model.preds] + [t for _, t in sampler.calls], feed_dict=feed)
        perms = dict(tf.shim.log["shuffle"][log0:])
        if real == "float32":
            out["eval/batch"] = np.asarray(feed[placeholders['batch']], np.int32)
            out["eval/labels"] = np.asarray(labels, np.float32)
            for k, (ns, t) in enumerate(sampler.calls):
                out["eval/perm%d" % k] = perms[id(shuffle_node_of(t))].astype(np.int32)
                out["eval/sampled%d" % k] = res[2 + k].astype(np.int32)
        out["eval/" + pre + "/loss"] = np.asarray(res[0])
        out["eval/" + pre + "/preds"] = res[1]
        sess.run(train_adj_info)
    out["n_steps"] = np.asarray(len(feeds))


# ----------------------------------------------------------------------------------------------------------------
# unsupervised cases
# ----------------------------------------------------------------------------------------------------------------
def run_unsupervised(cfg, real, out):
    G, feats, _, _ = make_graph()
    n = len(G.node)
    id_map = {i: i for i in range(n)}
    fresh(cfg["seed"], real)
    FLAGS.weight_decay = cfg["weight_decay"]
    FLAGS.learning_rate = cfg["learning_rate"]
    FLAGS.neg_sample_size = cfg["neg_sample_size"]
    # unsupervised_train.py:119-130
    placeholders = {
        'batch1': tf.placeholder(tf.int32, shape=(None), name='batch1'),
        'batch2': tf.placeholder(tf.int32, shape=(None), name='batch2'),
        'neg_samples': tf.placeholder(tf.int32, shape=(None,), name='neg_sample_size'),
        'dropout': tf.placeholder_with_default(0., shape=(), name='dropout'),
        'batch_size': tf.placeholder(tf.int32, name='batch_size'),
    }
    prng = np.random.RandomState(cfg["np_seed"] + 1)
    train_ok = [i for i in range(n) if not (G.node[i]['val'] or G.node[i]['test'])]
    pairs = []                                     # "random-walk co-occurrences": pairs of train nodes
    for _ in range(cfg["n_pairs"]):
        u = int(prng.choice(train_ok))
        nb = [v for v in G.neighbors(u) if not G[u][v]['train_removed']]
        pairs.append((u, int(prng.choice(nb)) if nb else u))
    np.random.seed(cfg["np_seed"])
    it = EdgeMinibatchIterator(G, id_map, placeholders, batch_size=cfg["batch_size"], max_degree=cfg["max_degree"],
                               num_neg_samples=cfg["neg_sample_size"], context_pairs=pairs)
    adj_info_ph = tf.placeholder(tf.int32, shape=it.adj.shape)
    adj_info = tf.Variable(adj_info_ph, trainable=False, name="adj_info")
    sampler = RecordingSampler(UniformNeighborSampler(adj_info))
    od = cfg["dim"] * (2 if cfg["aggregator_type"] == "gcn" else 1)
    layer_infos = [SAGEInfo("node", sampler, s, od) for s in cfg["num_samples"]]
    features = np.vstack([feats, np.zeros((feats.shape[1],))])
    model = SampleAndAggregate(placeholders, features, adj_info, it.deg, layer_infos=layer_infos,
                               aggregator_type=cfg["aggregator_type"], model_size=cfg.get("model_size", "small"), concat=cfg["concat"],
                               identity_dim=0, logging=False)
    nv = named_variables(model, False)
    names = sorted(nv)
    grads = tf.gradients(model.loss, [nv[k] for k in names])
    sess = tf.Session()
    sess.run(tf.global_variables_initializer(), feed_dict={adj_info_ph: it.adj})
    pre = real[-2:]
    if real == "float32":
        graph_arrays(G, feats, it, out)
        out["graph/pairs"] = np.asarray(pairs, np.int32)
        for k in names:
            out["init/" + k] = sess.run(nv[k]).astype(np.float32)
    feeds = []
    while not it.end():
        feeds.append(it.next_minibatch_feed_dict())
    K = len(layer_infos)
    assert len(sampler.calls) == 3 * K            # models.py:347-357: batch1, batch2, negatives
    for s, feed in enumerate(feeds):
        fetches = [model.opt_op, model.loss, model.mrr, model.ranks, model.aff_all, model.outputs1, model.outputs2,
                   model.neg_outputs, model.neg_samples]
        nf = len(fetches)
        fetches += [t for _, t in sampler.calls] + grads
        log0 = len(tf.shim.log["shuffle"])
        res = sess.run(fetches, feed_dict=feed)
        perms = dict(tf.shim.log["shuffle"][log0:])
        p = "s%d/" % s
        if real == "float32":
            out[p + "batch1"] = np.asarray(feed[placeholders['batch1']], np.int32)
            out[p + "batch2"] = np.asarray(feed[placeholders['batch2']], np.int32)
            out[p + "neg_samples"] = res[8].astype(np.int32)
            for k, (ns, t) in enumerate(sampler.calls):
                out[p + "perm%d" % k] = perms[id(shuffle_node_of(t))].astype(np.int32)
                out[p + "sampled%d" % k] = res[nf + k].astype(np.int32)
        else:
            assert np.array_equal(out[p + "neg_samples"], res[8])
        out[p + pre + "/loss"] = np.asarray(res[1])
        out[p + pre + "/mrr"] = np.asarray(res[2])
        out[p + pre + "/ranks"] = res[3]
        out[p + pre + "/aff_all"] = res[4]
        out[p + pre + "/outputs1"] = res[5]
        out[p + pre + "/outputs2"] = res[6]
        out[p + pre + "/neg_outputs"] = res[7]
        for k, g in zip(names, res[nf + 3 * K:]):
            out[p + pre + "/grad/" + k] = g
        for k in names:
            out[p + pre + "/after/" + k] = sess.run(nv[k])
    out["n_steps"] = np.asarray(len(feeds))


# ----------------------------------------------------------------------------------------------------------------
# stand-alone operator calls (the boundary of SURVEY 8b: sampler((ids, n)), aggregator((self, neigh)))
# ----------------------------------------------------------------------------------------------------------------
def run_operators(real, out):
    from graphsage.aggregators import GCNAggregator, MaxPoolingAggregator, MeanAggregator, MeanPoolingAggregator
    fresh(5, real)
    FLAGS.weight_decay = 0.0
    rng = np.random.RandomState(21)
    pre = real[-2:]
    n, s, d, o = 9, 4, 6, 8
    self_vecs = (np.round(rng.randn(n, d) * 8) / 8).astype(np.float32)
    neigh_vecs = (np.round(rng.randn(n, s, d) * 8) / 8).astype(np.float32)
    sess = tf.Session()
    if real == "float32":
        out["op/self_vecs"], out["op/neigh_vecs"] = self_vecs, neigh_vecs
    for tag, cls, kw in [("mean_concat", MeanAggregator, dict(concat=True)),
                         ("mean_add", MeanAggregator, dict(concat=False)),   # bias=True raises in the reference
                                                                              # (aggregators.py:35 reads output_dim
                                                                              # before :41 sets it)
                         ("mean_id", MeanAggregator, dict(concat=True, act=lambda x: x)),
                         ("gcn", GCNAggregator, dict()),
                         ("maxpool", MaxPoolingAggregator, dict(concat=True)),
                         ("meanpool", MeanPoolingAggregator, dict(concat=False))]:
        agg = cls(d, o, dropout=0., **kw)
        y = agg((tf.constant(self_vecs), tf.constant(neigh_vecs)))
        vs = dict(agg.vars)
        for l in getattr(agg, "mlp_layers", []):
            vs["mlp_weights"], vs["mlp_bias"] = l.vars['weights'], l.vars['bias']
        if "mlp_bias" in vs:
            sess.run(tf.assign(vs["mlp_bias"], (np.arange(512) % 7 - 3) / 16.0))
        names = sorted(vs)
        seedw = tf.constant((np.round(rng.randn(n, y_dim(cls, o, kw)) * 8) / 8).astype(np.float32))
        gs = tf.gradients(tf.reduce_sum(y * seedw), [vs[k] for k in names])
        res = sess.run([y, seedw] + gs)
        if real == "float32":
            out["op/%s/dy" % tag] = res[1].astype(np.float32)
            for k in names:
                out["op/%s/var/%s" % (tag, k)] = sess.run(vs[k]).astype(np.float32)
        out["op/%s/%s/y" % (tag, pre)] = res[0]
        for k, g in zip(names, res[2:]):
            out["op/%s/%s/grad/%s" % (tag, pre, k)] = g
    # the sampler operator on a hand-made table, including num_samples == max_degree
    adj = rng.randint(0, 30, size=(31, 6)).astype(np.int32)
    adj[30] = 30
    adj[7] = 30                              # an all-pad row (val node under the train adjacency / degree 0)
    sampler = UniformNeighborSampler(tf.Variable(tf.constant(adj, dtype=tf.int32), trainable=False))
    ids = np.asarray([3, 7, 7, 30, 0, 29, 12], np.int32)
    if real == "float32":
        out["op/sampler/adj"], out["op/sampler/ids"] = adj, ids
        for num in (1, 4, 6):
            t = sampler((tf.constant(ids, dtype=tf.int32), num))
            log0 = len(tf.shim.log["shuffle"])
            got = sess.run(t)
            out["op/sampler/perm_%d" % num] = tf.shim.log["shuffle"][log0][1].astype(np.int32)
            out["op/sampler/out_%d" % num] = got.astype(np.int32)


def y_dim(cls, o, kw):
    return o if cls.__name__ == "GCNAggregator" or not kw.get("concat") else 2 * o


# ----------------------------------------------------------------------------------------------------------------
SUP_CASES = {
    # name: config.  dims are multiples of 4 (the HIP concat kernels need out_dim % 4 == 0)
    "sup_mean": dict(aggregator_type="mean", concat=True, sigmoid=False, num_samples=[4, 3], dim=16, max_degree=8,
                     batch_size=16, batches="epoch", weight_decay=0.01, learning_rate=0.01, seed=1, np_seed=101,
                     eval_nodes=[0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 71]),
    "sup_mean_add_sigmoid": dict(aggregator_type="mean", concat=False, sigmoid=True, num_samples=[3, 2], dim=16,
                                 max_degree=5, batch_size=16, batches=[[4, 9, 9, 30, 2], [11, 12, 13, 14, 15, 16, 17]],
                                 weight_decay=0.0, learning_rate=0.02, seed=2, np_seed=102),
    "sup_gcn": dict(aggregator_type="gcn", concat=False, sigmoid=False, num_samples=[4, 3], dim=8, max_degree=8,
                    batch_size=16, batches=[list(range(20, 33)), list(range(40, 47))], weight_decay=0.005,
                    learning_rate=0.01, seed=3, np_seed=103, eval_nodes=list(range(0, 72, 5))),
    "sup_maxpool": dict(aggregator_type="maxpool", concat=True, sigmoid=False, num_samples=[4, 3], dim=16, max_degree=8,
                        batch_size=16, batches=[list(range(10, 22)), list(range(50, 59))], weight_decay=0.01,
                        learning_rate=0.01, seed=4, np_seed=104),
    "sup_meanpool_sigmoid": dict(aggregator_type="meanpool", concat=True, sigmoid=True, num_samples=[3, 3], dim=16,
                                 max_degree=6, batch_size=16, batches=[list(range(0, 10)), list(range(60, 72))],
                                 weight_decay=0.0, learning_rate=0.01, seed=5, np_seed=105),
    "sup_mean_3layer": dict(aggregator_type="mean", concat=True, sigmoid=False, num_samples=[3, 2, 2], dim=16,
                            max_degree=6, batch_size=16, batches=[list(range(30, 41)), list(range(3, 9))],
                            weight_decay=0.01, learning_rate=0.01, seed=6, np_seed=106),
    "sup_mean_full_degree": dict(aggregator_type="mean", concat=True, sigmoid=False, num_samples=[4, 4], dim=16,
                                 max_degree=4, batch_size=16, batches=[list(range(15, 27))], weight_decay=0.0,
                                 learning_rate=0.01, seed=7, np_seed=107),               # num_samples == max_degree
    "sup_mean_identity": dict(aggregator_type="mean", concat=True, sigmoid=False, num_samples=[4, 3], dim=16,
                              max_degree=8, batch_size=16, batches=[list(range(5, 17)), list(range(33, 44))],
                              weight_decay=0.01, learning_rate=0.01, seed=8, np_seed=108, identity_dim=6),
    # the headline model's own widths (dim_1 = dim_2 = 128, concat, fan-out <= 11): the device takes its fused-tail launch
    "sup_mean_tail": dict(aggregator_type="mean", concat=True, sigmoid=False, num_samples=[4, 3], dim=128, max_degree=8,
                          batch_size=32, batches=[list(range(8, 32)) + [40, 41, 42, 43, 44, 45, 46, 47, 48]],
                          weight_decay=0.001, learning_rate=0.01, seed=13, np_seed=113),
    # ... and the GCN model at widths its fused tail takes (dim 64 -> 2 * 64 = 128 per layer, supervised_train.py:175-185)
    "sup_gcn_tail": dict(aggregator_type="gcn", concat=False, sigmoid=False, num_samples=[4, 3], dim=64, max_degree=8,
                         batch_size=32, batches=[list(range(2, 30)) + [50, 51, 52, 53], list(range(30, 47))],
                         weight_decay=0.001, learning_rate=0.01, seed=17, np_seed=117),
    "sup_mean_dropout": dict(aggregator_type="mean", concat=True, sigmoid=False, num_samples=[4, 3], dim=16, max_degree=8,
                             batch_size=16, batches=[list(range(22, 35)), list(range(44, 52))], weight_decay=0.01,
                             learning_rate=0.01, seed=11, np_seed=111, dropout=0.3),
    "sup_maxpool_dropout": dict(aggregator_type="maxpool", concat=True, sigmoid=True, num_samples=[3, 2], dim=16,
                                max_degree=6, batch_size=16, batches=[list(range(12, 23))], weight_decay=0.0,
                                learning_rate=0.01, seed=12, np_seed=112, dropout=0.25),
    # FLAGS.model_size = "big": the pooling MLP is 1024 wide (aggregators.py:139-142)
    "sup_maxpool_big": dict(aggregator_type="maxpool", concat=True, sigmoid=False, num_samples=[3, 2], dim=16, max_degree=6,
                            batch_size=16, batches=[list(range(25, 36))], weight_decay=0.01, learning_rate=0.01, seed=15,
                            np_seed=115, model_size="big"),
}
UNSUP_CASES = {
    # embedding widths of 64 (the device's link-prediction launch takes d in {64, 128, 256, 512})
    "unsup_mean": dict(aggregator_type="mean", concat=True, num_samples=[4, 3], dim=32, max_degree=8, batch_size=12,
                       n_pairs=30, neg_sample_size=6, weight_decay=0.01, learning_rate=0.01, seed=9, np_seed=109),
    # ... at widths the device's fused layer-1 + link-prediction launches take (2 * 64 = 128 per layer; 12 pairs = one full and
    # one ragged group of 8 pairs)
    "unsup_mean_tail": dict(aggregator_type="mean", concat=True, num_samples=[4, 3], dim=64, max_degree=8, batch_size=12,
                            n_pairs=30, neg_sample_size=6, weight_decay=0.01, learning_rate=0.01, seed=18, np_seed=118),
    "unsup_gcn": dict(aggregator_type="gcn", concat=False, num_samples=[3, 3], dim=32, max_degree=6, batch_size=10,
                      n_pairs=18, neg_sample_size=5, weight_decay=0.0, learning_rate=0.02, seed=10, np_seed=110),
    # the pooling aggregator under the unsupervised objective (unsupervised_train.py:186-196: hidden_dim 512)
    "unsup_meanpool": dict(aggregator_type="meanpool", concat=True, num_samples=[3, 2], dim=32, max_degree=6, batch_size=7,
                           n_pairs=17, neg_sample_size=4, weight_decay=0.0, learning_rate=0.01, seed=16, np_seed=116),
    "unsup_maxpool": dict(aggregator_type="maxpool", concat=True, num_samples=[3, 2], dim=32, max_degree=6, batch_size=8,
                          n_pairs=16, neg_sample_size=4, weight_decay=0.005, learning_rate=0.01, seed=14, np_seed=114),
}


def main():
    only = set(sys.argv[1:])           # optional: names of the fixtures to (re)generate
    if only:
        for table in (SUP_CASES, UNSUP_CASES):
            for k in [k for k in table if k not in only]:
                del table[k]
    for name, cfg in SUP_CASES.items():
        out = {"cfg": np.asarray(json.dumps(dict(cfg, kind="supervised")))}
        for real in ("float32", "float64"):
            run_supervised(cfg, real, out)
        save(name, out)
    for name, cfg in UNSUP_CASES.items():
        out = {"cfg": np.asarray(json.dumps(dict(cfg, kind="unsupervised")))}
        for real in ("float32", "float64"):
            run_unsupervised(cfg, real, out)
        save(name, out)
    if only and "operators" not in only:
        return
    out = {}
    for real in ("float32", "float64"):
        run_operators(real, out)
    save("operators", out)


def save(name, out):
    path = os.path.join(os.environ.get("REF_FIXTURE_DIR", HERE), "ref_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-28s %4d arrays %8d bytes" % (os.path.basename(path), len(out), os.path.getsize(path)))


if __name__ == "__main__":
    main()
