"""Supervised training driver with the flags, loop structure, log-line format and output files of
graphsage/supervised_train.py (flags :28-57, loop :262-312, stats files :314-330), driving the MI355X
engine.  `sess.run(...)` becomes `model.train_step(feed_dict)` / `model.eval_step(feed_dict)`.

    python -m graphsage_amd.supervised_train --train_prefix ./example_data/toy-ppi --model graphsage_mean --sigmoid
    python -m graphsage_amd.supervised_train --synthetic ppi --model graphsage_mean --sigmoid --epochs 2

`--synthetic {ppi,reddit,small}` generates a graph of the reference dataset's shape (the datasets themselves
are not shipped with the reference: example_data/.MISSING_LARGE_BLOBS).  `--sampler {csr,padded}` selects the
MI355X-native CSR sampler (default) or the reference's padded-table sampler.
"""
from __future__ import division, print_function

import argparse
import os
import time

import numpy as np

# Set random seed (supervised_train.py:20-22)
seed = 123
np.random.seed(seed)


def build_flags(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    b = lambda v: str(v).lower() in ("1", "true", "yes")
    p.add_argument('--log_device_placement', type=b, default=False)
    # core params..
    p.add_argument('--model', default='graphsage_mean', help='model names. See README for possible values.')
    p.add_argument('--learning_rate', type=float, default=0.01, help='initial learning rate.')
    p.add_argument('--model_size', default='small', help="Can be big or small; model specific def'ns")
    p.add_argument('--train_prefix', default='', help='prefix identifying training data. must be specified.')
    # left to default values in main experiments
    p.add_argument('--epochs', type=int, default=10, help='number of epochs to train.')
    p.add_argument('--dropout', type=float, default=0.0, help='dropout rate (1 - keep probability).')
    p.add_argument('--weight_decay', type=float, default=0.0, help='weight for l2 loss on embedding matrix.')
    p.add_argument('--max_degree', type=int, default=128, help='maximum node degree.')
    p.add_argument('--samples_1', type=int, default=25, help='number of samples in layer 1')
    p.add_argument('--samples_2', type=int, default=10, help='number of samples in layer 2')
    p.add_argument('--samples_3', type=int, default=0, help='number of users samples in layer 3. (Only for mean model)')
    p.add_argument('--dim_1', type=int, default=128, help='Size of output dim (final is 2x this, if using concat)')
    p.add_argument('--dim_2', type=int, default=128, help='Size of output dim (final is 2x this, if using concat)')
    p.add_argument('--random_context', type=b, default=True, help='Whether to use random context or direct edges')
    p.add_argument('--batch_size', type=int, default=512, help='minibatch size.')
    p.add_argument('--sigmoid', action='store_true', default=False, help='whether to use sigmoid loss')
    p.add_argument('--identity_dim', type=int, default=0, help='identity embedding features dimension. Default 0.')
    # logging, saving, validation settings etc.
    p.add_argument('--base_log_dir', default='.', help='base directory for logging and saving embeddings')
    p.add_argument('--validate_iter', type=int, default=5000, help='how often to run a validation minibatch.')
    p.add_argument('--validate_batch_size', type=int, default=256, help='how many nodes per validation sample.')
    p.add_argument('--gpu', type=int, default=1, help='which gpu to use (ignored: one process per GPU, LOCAL_RANK).')
    p.add_argument('--print_every', type=int, default=5, help='How often to print training info.')
    p.add_argument('--max_total_steps', type=int, default=10 ** 10, help='Maximum total number of iterations')
    # additions of this engine
    p.add_argument('--synthetic', default='', help='ppi | reddit | small: generate a graph of that shape')
    p.add_argument('--sampler', default='csr', help='csr (MI355X-native) | padded (reference table semantics)')
    p.add_argument('--sampler_law', default='reference',
                   help='sampling law of the CSR sampler: reference (the reference\'s joint law on a virtual padded '
                        '[N+1, max_degree] table, minibatch.py:227-245 + neigh_samplers.py:24-29) | iid (independent '
                        'draws with replacement from the full neighbor list) | distinct (per-row without replacement)')
    p.add_argument('--feed_path', default='device',
                   help='device: epoch order + labels resident in HBM, host fetches only printed steps (CSR sampler); '
                        'host: the reference feed_dict path, one host round trip per step')
    return p.parse_args(argv)


FLAGS = None


def f1_micro_macro(y_true, y_pred, multilabel):
    """sklearn.metrics.f1_score(average="micro") and (average="macro") in NumPy (same definitions: per-class
    tp/fp/fn; macro averages the per-class F1 over the classes that occur in y_true or y_pred -- every column for
    multilabel indicator input -- with F1 = 0 where tp + fp + fn == 0 ... == 0 only for absent columns, which sklearn
    scores 0 as well).  The reference calls sklearn every print_every steps (supervised_train.py:63-70); at ~0.15 ms per
    training step that call would dominate the loop."""
    if multilabel:
        t = np.asarray(y_true) > 0.5
        p = np.asarray(y_pred) > 0.5
        tp = (t & p).sum(axis=0).astype(np.float64)
        fp = (~t & p).sum(axis=0).astype(np.float64)
        fn = (t & ~p).sum(axis=0).astype(np.float64)
    else:
        t = np.asarray(y_true).astype(np.int64)
        p = np.asarray(y_pred).astype(np.int64)
        classes = np.union1d(t, p)
        C = int(classes.max()) + 1 if classes.size else 1
        hit = t == p
        tp_all = np.bincount(t[hit], minlength=C).astype(np.float64)
        fp_all = np.bincount(p[~hit], minlength=C).astype(np.float64)
        fn_all = np.bincount(t[~hit], minlength=C).astype(np.float64)
        tp, fp, fn = tp_all[classes], fp_all[classes], fn_all[classes]
    den = 2 * tp + fp + fn
    per_class = np.where(den > 0, 2 * tp / np.where(den > 0, den, 1), 0.0)
    D = 2 * tp.sum() + fp.sum() + fn.sum()
    micro = float(2 * tp.sum() / D) if D > 0 else 0.0
    macro = float(per_class.mean()) if per_class.size else 0.0
    return micro, macro


def calc_f1(y_true, y_pred):
    """supervised_train.py:63-70"""
    y_pred = np.array(y_pred, copy=True)
    if not FLAGS.sigmoid:
        return f1_micro_macro(np.argmax(y_true, axis=1), np.argmax(y_pred, axis=1), False)
    return f1_micro_macro(y_true, y_pred > 0.5, True)


def evaluate(model, minibatch_iter, placeholders, size=None):
    """supervised_train.py:73-79"""
    t_test = time.time()
    feed_dict_val, labels = minibatch_iter.node_val_feed_dict(size)
    loss, preds = model.eval_step(feed_dict_val)
    mic, mac = calc_f1(labels, preds)
    return loss, mic, mac, (time.time() - t_test)


def log_dir():
    """supervised_train.py:81-89"""
    parts = FLAGS.train_prefix.split("/")
    tag = parts[-2] if len(parts) >= 2 else (FLAGS.synthetic or "data")
    d = FLAGS.base_log_dir + "/sup-" + tag
    d += "/{model:s}_{model_size:s}_{lr:0.4f}/".format(model=FLAGS.model, model_size=FLAGS.model_size,
                                                      lr=FLAGS.learning_rate)
    if not os.path.exists(d):
        os.makedirs(d)
    return d


def incremental_evaluate(model, minibatch_iter, size, test=False):
    """supervised_train.py:91-110"""
    t_test = time.time()
    val_losses, val_preds, labels = [], [], []
    iter_num = 0
    finished = False
    while not finished:
        feed_dict_val, batch_labels, finished, _ = minibatch_iter.incremental_node_val_feed_dict(size, iter_num, test=test)
        if len(batch_labels) > 0:
            loss, preds = model.eval_step(feed_dict_val)
            val_preds.append(preds)
            labels.append(batch_labels)
            val_losses.append(loss)
        iter_num += 1
    val_preds = np.vstack(val_preds)
    labels = np.vstack(labels)
    f1_scores = calc_f1(labels, val_preds)
    return np.mean(val_losses), f1_scores[0], f1_scores[1], (time.time() - t_test)


def construct_placeholders(num_classes):
    """supervised_train.py:112-120"""
    from .models import Placeholder
    return {'labels': Placeholder('labels'), 'batch': Placeholder('batch1'),
            'dropout': Placeholder('dropout', 0.), 'batch_size': Placeholder('batch_size')}


def load_graph():
    from . import utils
    if FLAGS.synthetic == 'ppi':       # toy-PPI shape: N=14,755, F=50, C=121 multi-label
        return utils.synthetic_graph(n_nodes=14755, feat_dim=50, num_classes=121, avg_degree=28, seed=seed,
                                     multilabel=True, val_frac=0.1, test_frac=0.1)
    if FLAGS.synthetic == 'reddit':
        return utils.reddit_shaped(seed=seed)
    if FLAGS.synthetic == 'small':
        return utils.synthetic_graph(n_nodes=3000, feat_dim=50, num_classes=7, avg_degree=8, seed=seed,
                                     multilabel=FLAGS.sigmoid)
    if not FLAGS.train_prefix:
        raise SystemExit("--train_prefix (or --synthetic) must be specified")
    return utils.load_data(FLAGS.train_prefix)


def train(G):
    from . import engine as eng
    from .minibatch import NodeMinibatchIterator
    from .models import SAGEInfo
    from .neigh_samplers import AdjInfo, CSRAdjacency, PaddedAdjacency, UniformNeighborSampler
    from .supervised_models import SupervisedGraphsage

    num_classes = G.num_classes
    features = G.padded_features()            # pad with dummy zero vector (:133-135)
    placeholders = construct_placeholders(num_classes)
    minibatch = NodeMinibatchIterator(G, None, placeholders, None, num_classes, batch_size=FLAGS.batch_size,
                                      max_degree=FLAGS.max_degree, build_padded=(FLAGS.sampler == 'padded'))
    e = eng.get_engine()
    if FLAGS.sampler == 'padded':
        train_adj = PaddedAdjacency(minibatch.adj, e.device)
        test_adj = PaddedAdjacency(minibatch.test_adj, e.device)
    else:
        train_adj = CSRAdjacency(minibatch.train_csr[0], minibatch.train_csr[1], G.n_nodes, e.device)
        test_adj = CSRAdjacency(minibatch.test_csr[0], minibatch.test_csr[1], G.n_nodes, e.device)
    adj_info = AdjInfo(train_adj)
    sampler = UniformNeighborSampler(adj_info, law=FLAGS.sampler_law, max_degree=FLAGS.max_degree)

    kw = dict(model_size=FLAGS.model_size, sigmoid_loss=FLAGS.sigmoid, identity_dim=FLAGS.identity_dim,
              learning_rate=FLAGS.learning_rate, weight_decay=FLAGS.weight_decay, logging=True)
    if FLAGS.model == 'graphsage_mean':       # :150-171
        if FLAGS.samples_3 != 0:
            layer_infos = [SAGEInfo("node", sampler, FLAGS.samples_1, FLAGS.dim_1),
                           SAGEInfo("node", sampler, FLAGS.samples_2, FLAGS.dim_2),
                           SAGEInfo("node", sampler, FLAGS.samples_3, FLAGS.dim_2)]
        elif FLAGS.samples_2 != 0:
            layer_infos = [SAGEInfo("node", sampler, FLAGS.samples_1, FLAGS.dim_1),
                           SAGEInfo("node", sampler, FLAGS.samples_2, FLAGS.dim_2)]
        else:
            layer_infos = [SAGEInfo("node", sampler, FLAGS.samples_1, FLAGS.dim_1)]
        model = SupervisedGraphsage(num_classes, placeholders, features, adj_info, minibatch.deg, layer_infos, **kw)
    elif FLAGS.model == 'gcn':                # :172-188
        layer_infos = [SAGEInfo("node", sampler, FLAGS.samples_1, 2 * FLAGS.dim_1),
                       SAGEInfo("node", sampler, FLAGS.samples_2, 2 * FLAGS.dim_2)]
        model = SupervisedGraphsage(num_classes, placeholders, features, adj_info, minibatch.deg,
                                    layer_infos=layer_infos, aggregator_type="gcn", concat=False, **kw)
    elif FLAGS.model in ('graphsage_maxpool', 'graphsage_meanpool', 'graphsage_seq'):   # :190-236
        layer_infos = [SAGEInfo("node", sampler, FLAGS.samples_1, FLAGS.dim_1),
                       SAGEInfo("node", sampler, FLAGS.samples_2, FLAGS.dim_2)]
        model = SupervisedGraphsage(num_classes, placeholders, features, adj_info, minibatch.deg,
                                    layer_infos=layer_infos, aggregator_type=FLAGS.model.split('_')[1], **kw)
    else:
        raise Exception('Error: model name unrecognized.')

    # Train model (:254-312).  Same loop, log line and validation cadence as the reference; the difference is WHERE the
    # batches come from: with the CSR sampler the shuffled epoch order and the label table live in HBM
    # (attach_device_epoch), and the steps between two host-visible events (a printed line, a validation, the end of
    # the epoch) are replayed as hipGraphs with no host round trip -- loss / preds are fetched only for the step whose
    # statistics are printed (`--feed_path host` keeps the reference's per-step feed_dict path).
    device_path = (FLAGS.sampler == 'csr' and FLAGS.feed_path == 'device')
    total_steps = 0
    avg_time = 0.0
    epoch_val_costs = []
    val_cost = val_f1_mic = val_f1_mac = 0.0
    B = FLAGS.batch_size
    if device_path:
        model.attach_device_epoch(minibatch.train_nodes, minibatch.label_matrix)

    def validate():
        adj_info.assign(test_adj)                             # sess.run(val_adj_info.op) (:280)
        if FLAGS.validate_batch_size == -1:
            res = incremental_evaluate(model, minibatch, FLAGS.batch_size)
        else:
            res = evaluate(model, minibatch, placeholders, FLAGS.validate_batch_size)
        adj_info.assign(train_adj)                            # sess.run(train_adj_info.op) (:285)
        return res

    for epoch in range(FLAGS.epochs):
        minibatch.shuffle()
        it = 0
        print('Epoch: %04d' % (epoch + 1))
        epoch_val_costs.append(0)
        if device_path:
            placeholders['dropout'].value = FLAGS.dropout
            model.set_epoch_order(minibatch.train_nodes)
            n_train = len(minibatch.train_nodes)
            n_iters = (n_train + B - 1) // B
            while it < n_iters:
                # the next iteration whose results the host looks at: a validation (it % validate_iter == 0), a printed
                # line (total_steps % print_every == 0) or the last, possibly short, batch of the epoch
                quiet = 0
                while (it + quiet < n_iters - 1 and (it + quiet) % FLAGS.validate_iter != 0
                       and (total_steps + quiet) % FLAGS.print_every != 0
                       and total_steps + quiet + 1 <= FLAGS.max_total_steps):
                    quiet += 1
                t = time.time()
                if quiet:
                    model.train_steps_device(B, quiet)                  # no host round trip
                it += quiet
                total_steps += quiet
                n = min(B, n_train - it * B)
                train_cost, preds = model.train_step_device(n, fetch=True)      # Training step (:275)
                batch_nodes = minibatch.train_nodes[it * B: it * B + n]
                labels = minibatch.label_matrix[batch_nodes]
                if it % FLAGS.validate_iter == 0:
                    val_cost, val_f1_mic, val_f1_mac, duration = validate()
                    epoch_val_costs[-1] += val_cost
                dt = (time.time() - t) / (quiet + 1)
                avg_time = (avg_time * (total_steps - quiet) + dt * (quiet + 1)) / (total_steps + 1)
                if total_steps % FLAGS.print_every == 0:
                    train_f1_mic, train_f1_mac = calc_f1(labels, preds)
                    print("Iter:", '%04d' % it,
                          "train_loss=", "{:.5f}".format(train_cost),
                          "train_f1_mic=", "{:.5f}".format(train_f1_mic),
                          "train_f1_mac=", "{:.5f}".format(train_f1_mac),
                          "val_loss=", "{:.5f}".format(val_cost),
                          "val_f1_mic=", "{:.5f}".format(val_f1_mic),
                          "val_f1_mac=", "{:.5f}".format(val_f1_mac),
                          "time=", "{:.5f}".format(avg_time))
                it += 1
                total_steps += 1
                if total_steps > FLAGS.max_total_steps:
                    break
            if total_steps > FLAGS.max_total_steps:
                break
            continue
        while not minibatch.end():
            feed_dict, labels = minibatch.next_minibatch_feed_dict()
            feed_dict.update({placeholders['dropout']: FLAGS.dropout})
            t = time.time()
            train_cost, preds = model.train_step(feed_dict)          # Training step (:275)
            if it % FLAGS.validate_iter == 0:
                val_cost, val_f1_mic, val_f1_mac, duration = validate()
                epoch_val_costs[-1] += val_cost
            avg_time = (avg_time * total_steps + time.time() - t) / (total_steps + 1)
            if total_steps % FLAGS.print_every == 0:
                train_f1_mic, train_f1_mac = calc_f1(labels, preds)
                print("Iter:", '%04d' % it,
                      "train_loss=", "{:.5f}".format(train_cost),
                      "train_f1_mic=", "{:.5f}".format(train_f1_mic),
                      "train_f1_mac=", "{:.5f}".format(train_f1_mac),
                      "val_loss=", "{:.5f}".format(val_cost),
                      "val_f1_mic=", "{:.5f}".format(val_f1_mic),
                      "val_f1_mac=", "{:.5f}".format(val_f1_mac),
                      "time=", "{:.5f}".format(avg_time))
            it += 1
            total_steps += 1
            if total_steps > FLAGS.max_total_steps:
                break
        if total_steps > FLAGS.max_total_steps:
            break

    print("Optimization Finished!")
    adj_info.assign(test_adj)
    val_cost, val_f1_mic, val_f1_mac, duration = incremental_evaluate(model, minibatch, FLAGS.batch_size)
    print("Full validation stats:",
          "loss=", "{:.5f}".format(val_cost),
          "f1_micro=", "{:.5f}".format(val_f1_mic),
          "f1_macro=", "{:.5f}".format(val_f1_mac),
          "time=", "{:.5f}".format(duration))
    with open(log_dir() + "val_stats.txt", "w") as fp:
        fp.write("loss={:.5f} f1_micro={:.5f} f1_macro={:.5f} time={:.5f}".format(val_cost, val_f1_mic, val_f1_mac, duration))
    print("Writing test set stats to file (don't peak!)")
    val_cost, val_f1_mic, val_f1_mac, duration = incremental_evaluate(model, minibatch, FLAGS.batch_size, test=True)
    with open(log_dir() + "test_stats.txt", "w") as fp:
        fp.write("loss={:.5f} f1_micro={:.5f} f1_macro={:.5f}".format(val_cost, val_f1_mic, val_f1_mac))
    return val_f1_mic


def main(argv=None):
    global FLAGS
    FLAGS = build_flags(argv)
    print("Loading training data..")
    G = load_graph()
    print("Done loading training data..")
    return train(G)


if __name__ == '__main__':
    main()
