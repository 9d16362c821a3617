"""NodeMinibatchIterator -- same attributes and methods as graphsage/minibatch.py:178-320
(.adj .test_adj .deg .train_nodes .val_nodes .test_nodes, next_minibatch_feed_dict(), shuffle(),
end(), node_val_feed_dict(), incremental_node_val_feed_dict(), ...), built from a GraphData edge
list instead of a networkx graph (networkx<=1.11 is not installable).

Besides the reference's padded tables it exposes the CSR views used by the MI355X-native sampler:
.train_csr / .test_csr  (rowptr, col) with identical edge filtering.
"""
import numpy as np

from .utils import build_csr, padded_from_csr

np.random.seed(123)  # minibatch.py:6


class NodeMinibatchIterator(object):
    """
    This minibatch iterator iterates over nodes for supervised learning.

    G -- GraphData (edge list + node attributes)
    id2idx -- kept for signature compatibility (identity map; GraphData is already indexed)
    placeholders -- dict of feed slots
    label_map -- ignored when G carries labels (signature compatibility)
    num_classes -- number of output classes
    batch_size -- size of the minibatches
    max_degree -- maximum size of the downsampled adjacency lists
    """

    def __init__(self, G, id2idx, placeholders, label_map, num_classes, batch_size=100, max_degree=25,
                 build_padded=True, **kwargs):
        self.G = G
        self.nodes = np.arange(G.n_nodes)
        self.id2idx = id2idx
        self.placeholders = placeholders
        self.batch_size = batch_size
        self.max_degree = max_degree
        self.batch_num = 0
        self.label_map = label_map
        self.num_classes = num_classes
        self.label_matrix = G.label_matrix()

        no_train = G.val_mask | G.test_mask
        # train view: edges that are not train_removed; val/test rows are skipped (minibatch.py:232-236)
        self.train_csr = build_csr(G.n_nodes, G.src, G.dst, keep=~G.train_removed)
        self.test_csr = build_csr(G.n_nodes, G.src, G.dst)                          # :247-259
        self.deg = np.diff(self.train_csr[0]).astype(np.int64)
        self.deg[no_train] = 0
        if build_padded:
            rng = np.random.RandomState(123)
            self.adj, _ = self.construct_adj(rng)
            self.test_adj = self.construct_test_adj(rng)
        else:
            self.adj = self.test_adj = None

        self.val_nodes = np.where(G.val_mask)[0]
        self.test_nodes = np.where(G.test_mask)[0]
        self.no_train_nodes_set = set(np.where(no_train)[0].tolist())
        train = np.where(~no_train)[0]
        # don't train on nodes that only have edges to test set (minibatch.py:214-215)
        self.train_nodes = train[self.deg[train] > 0]

    def _make_label_vec(self, node):
        return self.label_matrix[node]

    def construct_adj(self, rng=np.random):
        rowptr, col = self.train_csr
        return padded_from_csr(rowptr, col, self.G.n_nodes, self.max_degree, rng)

    def construct_test_adj(self, rng=np.random):
        rowptr, col = self.test_csr
        return padded_from_csr(rowptr, col, self.G.n_nodes, self.max_degree, rng)[0]

    def end(self):
        return self.batch_num * self.batch_size >= len(self.train_nodes)

    def batch_feed_dict(self, batch_nodes, val=False):
        batch1 = np.asarray(batch_nodes, dtype=np.int32)
        labels = self.label_matrix[batch1]
        feed_dict = dict()
        feed_dict.update({self.placeholders['batch_size']: len(batch1)})
        feed_dict.update({self.placeholders['batch']: batch1})
        feed_dict.update({self.placeholders['labels']: labels})
        return feed_dict, labels

    def node_val_feed_dict(self, size=None, test=False):
        val_nodes = self.test_nodes if test else self.val_nodes
        if size is not None:
            val_nodes = np.random.choice(val_nodes, size, replace=True)
        ret_val = self.batch_feed_dict(val_nodes)
        return ret_val[0], ret_val[1]

    def incremental_node_val_feed_dict(self, size, iter_num, test=False):
        val_nodes = self.test_nodes if test else self.val_nodes
        val_node_subset = val_nodes[iter_num * size:min((iter_num + 1) * size, len(val_nodes))]
        ret_val = self.batch_feed_dict(val_node_subset)
        return ret_val[0], ret_val[1], (iter_num + 1) * size >= len(val_nodes), val_node_subset

    def num_training_batches(self):
        return len(self.train_nodes) // self.batch_size + 1

    def next_minibatch_feed_dict(self):
        start_idx = self.batch_num * self.batch_size
        self.batch_num += 1
        end_idx = min(start_idx + self.batch_size, len(self.train_nodes))
        batch_nodes = self.train_nodes[start_idx: end_idx]
        return self.batch_feed_dict(batch_nodes)

    def incremental_embed_feed_dict(self, size, iter_num):
        node_list = self.nodes
        val_nodes = node_list[iter_num * size:min((iter_num + 1) * size, len(node_list))]
        return self.batch_feed_dict(val_nodes), (iter_num + 1) * size >= len(node_list), val_nodes

    def shuffle(self):
        """ Re-shuffle the training set.  Also reset the batch number."""
        self.train_nodes = np.random.permutation(self.train_nodes)
        self.batch_num = 0
