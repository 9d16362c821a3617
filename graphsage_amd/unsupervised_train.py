"""Unsupervised training driver with the flags, loop structure, log-line format and output files of
graphsage/unsupervised_train.py (flags :25-55, loop :246-316, save_val_embeddings :94-117), driving the MI355X engine.

    python -m graphsage_amd.unsupervised_train --train_prefix ./example_data/toy-ppi --model graphsage_mean --max_total_steps 1000
    python -m graphsage_amd.unsupervised_train --synthetic small --model graphsage_mean --epochs 1

Models: graphsage_mean | gcn | graphsage_maxpool | graphsage_meanpool (graphsage_seq and n2v are out of scope).
"""
from __future__ import division, print_function

import argparse
import os
import time

import numpy as np

seed = 123
np.random.seed(seed)
FLAGS = None


def build_flags(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    b = lambda v: str(v).lower() in ("1", "true", "yes")
    p.add_argument('--log_device_placement', type=b, default=False)
    p.add_argument('--model', default='graphsage', help='model names. See README for possible values.')
    p.add_argument('--learning_rate', type=float, default=0.00001, help='initial learning rate.')
    p.add_argument('--model_size', default='small', help="Can be big or small; model specific def'ns")
    p.add_argument('--train_prefix', default='', help='name of the object file that stores the training data.')
    p.add_argument('--epochs', type=int, default=1, help='number of epochs to train.')
    p.add_argument('--dropout', type=float, default=0.0, help='dropout rate (1 - keep probability).')
    p.add_argument('--weight_decay', type=float, default=0.0, help='weight for l2 loss on embedding matrix.')
    p.add_argument('--max_degree', type=int, default=100, help='maximum node degree.')
    p.add_argument('--samples_1', type=int, default=25, help='number of samples in layer 1')
    p.add_argument('--samples_2', type=int, default=10, help='number of users samples in layer 2')
    p.add_argument('--dim_1', type=int, default=128, help='Size of output dim (final is 2x this, if using concat)')
    p.add_argument('--dim_2', type=int, default=128, help='Size of output dim (final is 2x this, if using concat)')
    p.add_argument('--random_context', type=b, default=True, help='Whether to use random context or direct edges')
    p.add_argument('--neg_sample_size', type=int, default=20, help='number of negative samples')
    p.add_argument('--batch_size', type=int, default=512, help='minibatch size.')
    p.add_argument('--n2v_test_epochs', type=int, default=1, help='Number of new SGD epochs for n2v.')
    p.add_argument('--identity_dim', type=int, default=0)
    p.add_argument('--save_embeddings', type=b, default=True, help='whether to save embeddings for all nodes after training')
    p.add_argument('--base_log_dir', default='.', help='base directory for logging and saving embeddings')
    p.add_argument('--validate_iter', type=int, default=5000, help='how often to run a validation minibatch.')
    p.add_argument('--validate_batch_size', type=int, default=256, help='how many nodes per validation sample.')
    p.add_argument('--gpu', type=int, default=1, help='ignored: one process per GPU (LOCAL_RANK)')
    p.add_argument('--print_every', type=int, default=50, help='How often to print training info.')
    p.add_argument('--max_total_steps', type=int, default=10 ** 10, help='Maximum total number of iterations')
    p.add_argument('--synthetic', default='', help='ppi | reddit | small: generate a graph of that shape')
    p.add_argument('--sampler', default='csr', help='csr (MI355X-native) | padded (reference table semantics)')
    p.add_argument('--sampler_law', default='reference',
                   help='sampling law of the CSR sampler: reference (the reference\'s joint law on a virtual padded '
                        '[N+1, max_degree] table, minibatch.py:227-245 + neigh_samplers.py:24-29) | iid (independent '
                        'draws with replacement from the full neighbor list) | distinct (per-row without replacement)')
    p.add_argument('--max_walk_pairs', type=int, default=2000000, help='cap on generated random-walk pairs (synthetic data)')
    return p.parse_args(argv)


def log_dir():
    """unsupervised_train.py:61-69"""
    parts = FLAGS.train_prefix.split("/")
    tag = parts[-2] if len(parts) >= 2 else (FLAGS.synthetic or "data")
    d = FLAGS.base_log_dir + "/unsup-" + tag
    d += "/{model:s}_{model_size:s}_{lr:0.6f}/".format(model=FLAGS.model, model_size=FLAGS.model_size, lr=FLAGS.learning_rate)
    if not os.path.exists(d):
        os.makedirs(d)
    return d


def evaluate(model, minibatch_iter, size=None):
    """unsupervised_train.py:72-77"""
    t_test = time.time()
    feed_dict_val = minibatch_iter.val_feed_dict(size)
    loss, ranks, mrr, _ = model.eval_step(feed_dict_val)
    return loss, ranks, mrr, (time.time() - t_test)


def incremental_evaluate(model, minibatch_iter, size):
    """unsupervised_train.py:79-92"""
    t_test = time.time()
    finished = False
    val_losses, val_mrrs = [], []
    iter_num = 0
    while not finished:
        feed_dict_val, finished, edges = minibatch_iter.incremental_val_feed_dict(size, iter_num)
        iter_num += 1
        if len(edges) == 0:
            continue
        loss, ranks, mrr, _ = model.eval_step(feed_dict_val)
        val_losses.append(loss)
        val_mrrs.append(mrr)
    return np.mean(val_losses), np.mean(val_mrrs), (time.time() - t_test)


def save_val_embeddings(model, minibatch_iter, size, out_dir, mod=""):
    """unsupervised_train.py:94-117: embeddings of every node via (n, n) pairs -> val.npy + val.txt"""
    val_embeddings, nodes = [], []
    seen = set()
    finished = False
    iter_num = 0
    name = "val"
    while not finished:
        feed_dict_val, finished, edges = minibatch_iter.incremental_embed_feed_dict(size, iter_num)
        iter_num += 1
        if len(edges) == 0:
            continue
        _, _, _, outputs1 = model.eval_step(feed_dict_val)
        # ONLY SAVE FOR embeds1 because of planetoid
        for i, edge in enumerate(edges):
            if not edge[0] in seen:
                val_embeddings.append(outputs1[i, :])
                ids = getattr(minibatch_iter.G, "node_ids", None)     # val.txt lists ORIGINAL node ids (:106-110)
                nodes.append(ids[edge[0]] if ids is not None else edge[0])
                seen.add(edge[0])
    if not os.path.exists(out_dir):
        os.makedirs(out_dir)
    val_embeddings = np.vstack(val_embeddings)
    np.save(out_dir + name + mod + ".npy", val_embeddings)
    with open(out_dir + name + mod + ".txt", "w") as fp:
        fp.write("\n".join(map(str, nodes)))


def construct_placeholders():
    """unsupervised_train.py:119-130"""
    from .models import Placeholder
    return {'batch1': Placeholder('batch1'), 'batch2': Placeholder('batch2'), 'neg_samples': Placeholder('neg_sample_size'),
            'dropout': Placeholder('dropout', 0.), 'batch_size': Placeholder('batch_size')}


def train(G, context_pairs):
    from . import engine as eng
    from .minibatch import EdgeMinibatchIterator
    from .models import SAGEInfo, SampleAndAggregate
    from .neigh_samplers import AdjInfo, CSRAdjacency, PaddedAdjacency, UniformNeighborSampler

    features = G.padded_features()
    placeholders = construct_placeholders()
    minibatch = EdgeMinibatchIterator(G, None, placeholders, batch_size=FLAGS.batch_size, max_degree=FLAGS.max_degree,
                                      context_pairs=context_pairs if FLAGS.random_context else None,
                                      build_padded=(FLAGS.sampler == 'padded'))
    e = eng.get_engine()
    if FLAGS.sampler == 'padded':
        train_adj, test_adj = PaddedAdjacency(minibatch.adj, e.device), PaddedAdjacency(minibatch.test_adj, e.device)
    else:
        train_adj = CSRAdjacency(minibatch.train_csr[0], minibatch.train_csr[1], G.n_nodes, e.device)
        test_adj = CSRAdjacency(minibatch.test_csr[0], minibatch.test_csr[1], G.n_nodes, e.device)
    adj_info = AdjInfo(train_adj)
    sampler = UniformNeighborSampler(adj_info, law=FLAGS.sampler_law, max_degree=FLAGS.max_degree)
    kw = dict(model_size=FLAGS.model_size, identity_dim=FLAGS.identity_dim, learning_rate=FLAGS.learning_rate,
              weight_decay=FLAGS.weight_decay, neg_sample_size=FLAGS.neg_sample_size, logging=True)
    if FLAGS.model in ('graphsage_mean', 'graphsage'):          # unsupervised_train.py:160-172
        layer_infos = [SAGEInfo("node", sampler, FLAGS.samples_1, FLAGS.dim_1),
                       SAGEInfo("node", sampler, FLAGS.samples_2, FLAGS.dim_2)]
        model = SampleAndAggregate(placeholders, features, adj_info, minibatch.deg, layer_infos=layer_infos, **kw)
    elif FLAGS.model == 'gcn':                                   # :173-187
        layer_infos = [SAGEInfo("node", sampler, FLAGS.samples_1, 2 * FLAGS.dim_1),
                       SAGEInfo("node", sampler, FLAGS.samples_2, 2 * FLAGS.dim_2)]
        model = SampleAndAggregate(placeholders, features, adj_info, minibatch.deg, layer_infos=layer_infos,
                                   aggregator_type="gcn", concat=False, **kw)
    elif FLAGS.model in ('graphsage_maxpool', 'graphsage_meanpool'):   # :203-230
        layer_infos = [SAGEInfo("node", sampler, FLAGS.samples_1, FLAGS.dim_1),
                       SAGEInfo("node", sampler, FLAGS.samples_2, FLAGS.dim_2)]
        model = SampleAndAggregate(placeholders, features, adj_info, minibatch.deg, layer_infos=layer_infos,
                                   aggregator_type=FLAGS.model.split('_')[1], **kw)
    else:
        raise Exception('Error: model name unrecognized.')

    train_shadow_mrr = None
    shadow_mrr = None
    total_steps = 0
    avg_time = 0.0
    epoch_val_costs = []
    val_cost = val_mrr = 0.0
    for epoch in range(FLAGS.epochs):
        minibatch.shuffle()
        it = 0
        print('Epoch: %04d' % (epoch + 1))
        epoch_val_costs.append(0)
        while not minibatch.end():
            feed_dict = minibatch.next_minibatch_feed_dict()
            feed_dict.update({placeholders['dropout']: FLAGS.dropout})
            t = time.time()
            train_cost, ranks, aff_all, train_mrr, outputs1 = model.train_step(feed_dict)     # :273-274
            if train_shadow_mrr is None:
                train_shadow_mrr = train_mrr
            else:
                train_shadow_mrr -= (1 - 0.99) * (train_shadow_mrr - train_mrr)
            if it % FLAGS.validate_iter == 0:
                adj_info.assign(test_adj)
                val_cost, ranks, val_mrr, duration = evaluate(model, minibatch, size=FLAGS.validate_batch_size)
                adj_info.assign(train_adj)
                epoch_val_costs[-1] += val_cost
            if shadow_mrr is None:
                shadow_mrr = val_mrr
            else:
                shadow_mrr -= (1 - 0.99) * (shadow_mrr - val_mrr)
            avg_time = (avg_time * total_steps + time.time() - t) / (total_steps + 1)
            if total_steps % FLAGS.print_every == 0:
                print("Iter:", '%04d' % it,
                      "train_loss=", "{:.5f}".format(train_cost),
                      "train_mrr=", "{:.5f}".format(train_mrr),
                      "train_mrr_ema=", "{:.5f}".format(train_shadow_mrr),
                      "val_loss=", "{:.5f}".format(val_cost),
                      "val_mrr=", "{:.5f}".format(val_mrr),
                      "val_mrr_ema=", "{:.5f}".format(shadow_mrr),
                      "time=", "{:.5f}".format(avg_time))
            it += 1
            total_steps += 1
            if total_steps > FLAGS.max_total_steps:
                break
        if total_steps > FLAGS.max_total_steps:
            break
    print("Optimization Finished!")
    if FLAGS.save_embeddings:
        adj_info.assign(test_adj)
        save_val_embeddings(model, minibatch, FLAGS.validate_batch_size, log_dir())
    return shadow_mrr


def main(argv=None):
    global FLAGS
    FLAGS = build_flags(argv)
    from . import utils
    from . import supervised_train as st
    print("Loading training data..")
    st.FLAGS = argparse.Namespace(synthetic=FLAGS.synthetic, train_prefix=FLAGS.train_prefix, sigmoid=False)
    G = st.load_graph()
    pairs = None
    if FLAGS.random_context:
        walks = FLAGS.train_prefix + "-walks.txt"
        if FLAGS.train_prefix and os.path.exists(walks):          # utils.py:70-74
            pairs = utils.load_walk_pairs(walks, G)                # original ids -> rows via id_map
        else:
            rp, col = utils.build_csr(G.n_nodes, G.src, G.dst, keep=~G.train_removed)
            train_nodes = np.where(~(G.val_mask | G.test_mask))[0]
            pairs = utils.run_random_walks(rp, col, train_nodes, max_pairs=FLAGS.max_walk_pairs)
    print("Done loading training data..")
    return train(G, pairs)


if __name__ == '__main__':
    main()
