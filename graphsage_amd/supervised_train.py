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
    return p.parse_args(argv)


FLAGS = None


def calc_f1(y_true, y_pred):
    """supervised_train.py:63-70"""
    from sklearn import metrics
    y_pred = np.array(y_pred, copy=True)
    if not FLAGS.sigmoid:
        y_true = np.argmax(y_true, axis=1)
        y_pred = np.argmax(y_pred, axis=1)
    else:
        y_pred[y_pred > 0.5] = 1
        y_pred[y_pred <= 0.5] = 0
    return metrics.f1_score(y_true, y_pred, average="micro"), metrics.f1_score(y_true, y_pred, average="macro")


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
    sampler = UniformNeighborSampler(adj_info)

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

    # Train model (:254-312)
    total_steps = 0
    avg_time = 0.0
    epoch_val_costs = []
    val_cost = val_f1_mic = val_f1_mac = 0.0
    for epoch in range(FLAGS.epochs):
        minibatch.shuffle()
        it = 0
        print('Epoch: %04d' % (epoch + 1))
        epoch_val_costs.append(0)
        while not minibatch.end():
            feed_dict, labels = minibatch.next_minibatch_feed_dict()
            feed_dict.update({placeholders['dropout']: FLAGS.dropout})
            t = time.time()
            train_cost, preds = model.train_step(feed_dict)          # Training step (:275)
            if it % FLAGS.validate_iter == 0:
                adj_info.assign(test_adj)                             # sess.run(val_adj_info.op) (:280)
                if FLAGS.validate_batch_size == -1:
                    val_cost, val_f1_mic, val_f1_mac, duration = incremental_evaluate(model, minibatch, FLAGS.batch_size)
                else:
                    val_cost, val_f1_mic, val_f1_mac, duration = evaluate(model, minibatch, placeholders,
                                                                          FLAGS.validate_batch_size)
                adj_info.assign(train_adj)                            # sess.run(train_adj_info.op) (:285)
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
