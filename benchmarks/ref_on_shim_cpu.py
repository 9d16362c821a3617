"""cpu_baseline kind "reference-on-shim": the REFERENCE'S OWN training step timed on host cores.

    python benchmarks/ref_on_shim_cpu.py [out.json]          # needs /root/reference (or $GRAPHSAGE_REFERENCE): runs in the build
                                                             # container only -- the reference cannot travel to the GPU box

`/root/reference/graphsage/supervised_models.py` (+ models / aggregators / layers / neigh_samplers / minibatch, all UNMODIFIED)
is imported on the torch-backed TF 1.x stand-in of tests/tf1_shim and driven exactly like supervised_train.py:258-275 --
`sess.run([model.opt_op, model.loss, model.preds], feed_dict)` per mini-batch -- at the shapes of BASELINE configs[0]
(example_supervised.sh:1: toy-PPI-shaped graph N = 14,755, F = 50, C = 121 multi-hot, --sigmoid, graphsage_mean, B = 512,
fan-out 25 x 10, dims 128 / 128, max_degree 128).  Beside it, on the same cores and the same graph, the torch-CPU PORT that
bench.py's `cpu_baseline` / `aux.toy_ppi.cpu_port` legs time on the GPU box (oracle/cpu_baseline.py): the ratio of the two says
how much of the port's number is the stand-in's overhead (a lazy dataflow graph evaluated per Session.run) -- TensorFlow 1.8
itself cannot be installed here, so this is the closest available timing of the reference's graph, and it is TEST
INFRASTRUCTURE: nothing under graphsage_amd/ or bench.py's timed region touches it.  bench.py attaches the committed record
(profiles/r06_ref_on_shim_cpu.json) to its JSON line as `cpu_baseline.reference_on_shim`, labelled PROFILE-SOURCED."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_ref_on_shim_cpu.json")
    import make_ref_fixtures as mrf            # puts tests/tf1_shim and the reference on sys.path, defines the flags
    import torch
    tf = mrf.tf
    from graphsage_amd.utils import synthetic_graph
    N, F, C, B, s1, s2, D, MAXD = 14755, 50, 121, 512, 25, 10, 128, 128
    STEPS = int(os.environ.get("REF_SHIM_STEPS", "12"))
    G0 = synthetic_graph(n_nodes=N, feat_dim=F, num_classes=C, avg_degree=28, seed=123, multilabel=True)
    val, test = G0.val_mask, G0.test_mask
    und = {(int(min(u, v)), int(max(u, v))) for u, v in zip(G0.src, G0.dst) if u != v}
    G = mrf.RefGraph(N, sorted(und), val, test)
    labels = G0.label_matrix()[:N]
    class_map = {i: labels[i].astype(int).tolist() for i in range(N)}
    id_map = {i: i for i in range(N)}
    mrf.fresh(123, "float32")
    mrf.FLAGS.weight_decay, mrf.FLAGS.learning_rate = 0.0, 0.01
    placeholders = {                                                            # supervised_train.py:112-120
        'labels': tf.placeholder(tf.float32, shape=(None, C), name='labels'),
        'batch': tf.placeholder(tf.int32, shape=(None), name='batch1'),
        'dropout': tf.placeholder_with_default(0., shape=(), name='dropout'),
        'batch_size': tf.placeholder(tf.int32, name='batch_size'),
    }
    np.random.seed(123)
    t0 = time.time()
    it = mrf.NodeMinibatchIterator(G, id_map, placeholders, class_map, C, batch_size=B, max_degree=MAXD)
    t_adj = time.time() - t0
    adj_info_ph = tf.placeholder(tf.int32, shape=it.adj.shape)
    adj_info = tf.Variable(adj_info_ph, trainable=False, name="adj_info")      # supervised_train.py:147-148
    sampler = mrf.UniformNeighborSampler(adj_info)
    layer_infos = [mrf.SAGEInfo("node", sampler, s1, D), mrf.SAGEInfo("node", sampler, s2, D)]
    feats = np.vstack([G0.feats[:N], np.zeros((F,))])
    model = mrf.SupervisedGraphsage(C, placeholders, feats, adj_info, it.deg, layer_infos=layer_infos, aggregator_type="mean",
                                    model_size="small", sigmoid_loss=True, concat=True, identity_dim=0, logging=False)
    sess = tf.Session()
    sess.run(tf.global_variables_initializer(), feed_dict={adj_info_ph: it.adj})
    it.shuffle()
    ts, losses = [], []
    for s in range(STEPS):
        feed, lab = it.next_minibatch_feed_dict()
        t1 = time.time()
        outs = sess.run([model.opt_op, model.loss, model.preds], feed_dict=feed)          # supervised_train.py:275
        ts.append(time.time() - t1)
        losses.append(float(outs[1]))
    ref_s = float(np.median(ts[2:]))
    edges = B * (s2 + s2 * s1)
    # ---- the port on the same graph / cores
    from oracle.cpu_baseline import CpuSupervisedMean
    from graphsage_amd.utils import build_csr, padded_from_csr
    nt = val | test
    rp, col = build_csr(N, G0.src, G0.dst, keep=~(nt[G0.src] | nt[G0.dst]))
    adj, deg = padded_from_csr(rp, col, N, MAXD, np.random.RandomState(123))
    port = CpuSupervisedMean(G0.padded_features(), adj, [F, D, D], C, [s1, s2], sigmoid_loss=True, seed=123)
    train_nodes = np.nonzero(~nt & (deg[:N] > 0))[0].astype(np.int32)
    order = np.random.RandomState(1).permutation(train_nodes)
    lab_all = G0.label_matrix()
    tp = []
    for s in range(STEPS):
        b = order[s * B:(s + 1) * B]
        t1 = time.time()
        port.train_step(b, lab_all[b])
        tp.append(time.time() - t1)
    port_s = float(np.median(tp[2:]))
    rec = {"kind": "reference-on-shim",
           "what": "sess.run([model.opt_op, model.loss, model.preds]) of /root/reference/graphsage/supervised_models.py (unmodified) on "
                   "tests/tf1_shim (torch-backed TF 1.x stand-in), supervised_train.py:275",
           "config": "configs[0]-shaped: N=%d, F=%d, C=%d multi-hot, --sigmoid graphsage_mean, B=%d, fan-out %dx%d, dims %d/%d, "
                     "max_degree %d; synthetic toy-PPI-shaped graph (the reference's blobs are stripped)" % (N, F, C, B, s1, s2, D, D, MAXD),
           "host": "build container", "cores": int(torch.get_num_threads()), "steps_timed": STEPS - 2,
           "reference_on_shim": {"s_per_step": ref_s, "value": edges / ref_s, "unit": "sampled-edges/s", "loss_first_last": [losses[0], losses[-1]]},
           "port_same_cores": {"s_per_step": port_s, "value": edges / port_s, "unit": "sampled-edges/s"},
           "construct_adj_s": t_adj,
           "note": "TensorFlow 1.8 cannot be installed in this image: the reference's graph is timed on the stand-in, whose lazy "
                   "dataflow evaluation and autograd-based tf.gradients are slower than TF's C++ executor would be -- an upper bound "
                   "on the reference's time per step, next to the port (the same op sequence as eager torch) on the same cores"}
    with open(out_path, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
