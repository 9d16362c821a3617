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
        self.nodes = np.where(G.present)[0]          # G.nodes(): nodes removed by load_data are not iterated
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



class EdgeMinibatchIterator(object):
    """This minibatch iterator iterates over batches of sampled edges or random pairs of co-occuring edges
    (graphsage/minibatch.py:8-176), built from a GraphData instead of a networkx graph.

    G -- GraphData;  id2idx -- ignored (identity);  placeholders -- dict of feed slots
    context_pairs -- if not None, an int array [n, 2] of co-occurring node pairs (from random walks)
    batch_size -- size of the minibatches;  max_degree -- size of the padded adjacency lists
    """

    def __init__(self, G, id2idx, placeholders, context_pairs=None, batch_size=100, max_degree=25, n2v_retrain=False,
                 fixed_n2v=False, build_padded=True, **kwargs):
        if n2v_retrain:
            raise NotImplementedError("the node2vec baseline is out of scope (SURVEY §2 #12)")
        self.G = G
        self.id2idx = id2idx
        self.placeholders = placeholders
        self.batch_size = batch_size
        self.max_degree = max_degree
        self.batch_num = 0
        self.nodes = np.random.permutation(np.where(G.present)[0])
        no_train = G.val_mask | G.test_mask
        self.train_csr = build_csr(G.n_nodes, G.src, G.dst, keep=~G.train_removed)
        self.test_csr = build_csr(G.n_nodes, G.src, G.dst)
        self.deg = np.diff(self.train_csr[0]).astype(np.int64)
        self.deg[no_train] = 0
        if build_padded:
            rng = np.random.RandomState(123)
            self.adj, _ = padded_from_csr(self.train_csr[0], self.train_csr[1], G.n_nodes, max_degree, rng)
            self.test_adj = padded_from_csr(self.test_csr[0], self.test_csr[1], G.n_nodes, max_degree, rng)[0]
        else:
            self.adj = self.test_adj = None
        if context_pairs is None:
            edges = np.stack([G.src, G.dst], axis=1)
        else:
            edges = np.asarray(context_pairs, dtype=np.int32).reshape(-1, 2)
        self.train_edges = self.edges = np.random.permutation(edges)
        self.train_edges = self._remove_isolated(self.train_edges)
        self.val_edges = np.stack([G.src, G.dst], axis=1)[G.train_removed]          # minibatch.py:45
        print(int((~no_train & G.present).sum()), 'train nodes')
        print(int(no_train.sum()), 'test nodes')
        self.val_set_size = len(self.val_edges)

    def _remove_isolated(self, edge_list):
        """minibatch.py:60-74: drop pairs with a zero-train-degree endpoint unless an endpoint is a test node."""
        n1, n2 = edge_list[:, 0], edge_list[:, 1]
        G = self.G
        iso = (self.deg[n1] == 0) | (self.deg[n2] == 0)
        c1 = (~G.test_mask[n1]) | G.val_mask[n1]
        c2 = (~G.test_mask[n2]) | G.val_mask[n2]
        return edge_list[~(iso & c1 & c2)]

    def end(self):
        return self.batch_num * self.batch_size >= len(self.train_edges)

    def batch_feed_dict(self, batch_edges):
        batch_edges = np.asarray(batch_edges, dtype=np.int32).reshape(-1, 2)
        feed_dict = dict()
        feed_dict.update({self.placeholders['batch_size']: len(batch_edges)})
        feed_dict.update({self.placeholders['batch1']: batch_edges[:, 0]})
        feed_dict.update({self.placeholders['batch2']: batch_edges[:, 1]})
        return feed_dict

    def next_minibatch_feed_dict(self):
        start_idx = self.batch_num * self.batch_size
        self.batch_num += 1
        end_idx = min(start_idx + self.batch_size, len(self.train_edges))
        return self.batch_feed_dict(self.train_edges[start_idx: end_idx])

    def num_training_batches(self):
        return len(self.train_edges) // self.batch_size + 1

    def val_feed_dict(self, size=None):
        edge_list = self.val_edges
        if size is None:
            return self.batch_feed_dict(edge_list)
        ind = np.random.permutation(len(edge_list))
        return self.batch_feed_dict(edge_list[ind[:min(size, len(ind))]])

    def incremental_val_feed_dict(self, size, iter_num):
        edge_list = self.val_edges
        val_edges = edge_list[iter_num * size:min((iter_num + 1) * size, len(edge_list))]
        return self.batch_feed_dict(val_edges), (iter_num + 1) * size >= len(self.val_edges), val_edges

    def incremental_embed_feed_dict(self, size, iter_num):
        node_list = self.nodes
        val_nodes = node_list[iter_num * size:min((iter_num + 1) * size, len(node_list))]
        val_edges = np.stack([val_nodes, val_nodes], axis=1)
        return self.batch_feed_dict(val_edges), (iter_num + 1) * size >= len(node_list), val_edges

    def label_val(self):
        G = self.G
        e = np.stack([G.src, G.dst], axis=1)
        return e[~G.train_removed], e[G.train_removed]

    def shuffle(self):
        """ Re-shuffle the training set.  Also reset the batch number."""
        self.train_edges = np.random.permutation(self.train_edges)
        self.nodes = np.random.permutation(self.nodes)
        self.batch_num = 0
