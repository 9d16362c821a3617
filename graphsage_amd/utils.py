"""Graph containers, synthetic dataset generators and adjacency builders.

Replaces the networkx/json loading of graphsage/utils.py:19-75 for this engine: the reference's
datasets are not shipped (example_data/.MISSING_LARGE_BLOBS) and networkx<=1.11 is not installable,
so inputs are synthetic graphs of the reference's shapes.  The adjacency semantics
of minibatch.py:227-259 are reproduced exactly:
  * train adjacency: only train nodes have rows; only edges whose both endpoints are train nodes
    (`train_removed` is set when either endpoint is val/test, utils.py:55-60);
  * test adjacency: all nodes, all edges;
  * pad id = N; padded rows are down/up-sampled ONCE to max_degree.
"""
import json
import os

import numpy as np


class GraphData(object):
    """Undirected graph as an edge list plus node attributes (what load_data returns, in arrays)."""

    def __init__(self, n_nodes, src, dst, feats, labels, val_mask, test_mask, multilabel=False, present=None,
                 id_map=None, node_ids=None):
        self.n_nodes = int(n_nodes)
        # present[i] is False for rows of the id map whose node was dropped from the graph (utils.py:43-50: nodes
        # lacking val/test annotations are removed); such rows keep their feature row but have no edges and belong to
        # no split.  id_map: original node id (str) -> row; node_ids: row -> original node id (utils.py:32-33).
        self.present = np.ones(self.n_nodes, dtype=bool) if present is None else np.asarray(present, dtype=bool)
        self.id_map = id_map
        self.node_ids = node_ids
        self.src = np.ascontiguousarray(src, dtype=np.int32)
        self.dst = np.ascontiguousarray(dst, dtype=np.int32)
        self.feats = feats                      # float32 [N, F] (no pad row yet)
        self.labels = labels                    # int64 [N] class ids, or float32 [N, C] multi-hot
        self.val_mask = np.asarray(val_mask, dtype=bool)
        self.test_mask = np.asarray(test_mask, dtype=bool)
        self.multilabel = multilabel
        # utils.py:55-60: an edge is train_removed if either endpoint is a val or test node
        nt = self.val_mask | self.test_mask
        self.train_removed = nt[self.src] | nt[self.dst]

    @property
    def num_classes(self):
        if self.multilabel:
            return int(self.labels.shape[1])
        return int(self.labels.max()) + 1

    def label_matrix(self):
        """[N+1, C] float32 one-hot / multi-hot rows (minibatch.py:217-225); row N (pad) is zeros."""
        C = self.num_classes
        out = np.zeros((self.n_nodes + 1, C), dtype=np.float32)
        if self.multilabel:
            out[: self.n_nodes] = self.labels
        else:
            out[np.arange(self.n_nodes), self.labels] = 1.0
        return out

    def padded_features(self):
        """features = vstack([features, zeros(F)])  (supervised_train.py:133-135); cached.  None when the dataset has
        no feature file (identity features only, utils.py:41-43 / supervised_train.py:132)."""
        if self.feats is None:
            return None
        if getattr(self, "_padded", None) is None:
            self._padded = np.vstack([self.feats, np.zeros((1, self.feats.shape[1]), dtype=np.float32)])
        return self._padded


def standardize_on_train(feats, train_mask, chunk=16384):
    """StandardScaler fit on the train rows, applied to all rows (utils.py:62-68).  In place and
    chunked: a 233K x 602 table is 561 MB and page-faulting temporaries dominate otherwise."""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    n, f = feats.shape
    cnt = 0
    s1 = np.zeros(f, dtype=np.float64)
    s2 = np.zeros(f, dtype=np.float64)
    for a in range(0, n, chunk):
        blk = feats[a:a + chunk][train_mask[a:a + chunk]]
        cnt += blk.shape[0]
        s1 += blk.sum(axis=0, dtype=np.float64)
        s2 += np.einsum("ij,ij->j", blk, blk, dtype=np.float64)
    mean = s1 / max(cnt, 1)
    var = np.maximum(s2 / max(cnt, 1) - mean * mean, 0.0)
    std = np.sqrt(var)
    std[std == 0] = 1.0
    mean32, inv32 = mean.astype(np.float32), (1.0 / std).astype(np.float32)
    for a in range(0, n, chunk):
        blk = feats[a:a + chunk]
        blk -= mean32
        blk *= inv32
    return feats


def build_csr(n_nodes, src, dst, keep=None):
    """Symmetric CSR (rowptr int64, col int32) from an undirected edge list, via the C++ builder."""
    from . import ops
    return ops.build_csr_host(src, dst, n_nodes, symmetrize=True, keep_mask=keep)


def build_csr_numpy(n_nodes, src, dst, keep=None):
    """Pure-NumPy CSR builder (host logic check for the C++ builder)."""
    if keep is not None:
        src, dst = src[keep], dst[keep]
    loops = src == dst
    s = np.concatenate([src, dst[~loops]]).astype(np.int64)
    d = np.concatenate([dst, src[~loops]]).astype(np.int64)
    order = np.lexsort((d, s))
    s, d = s[order], d[order]
    if s.size:  # one edge per node pair, like networkx.Graph
        first = np.ones(s.size, dtype=bool)
        first[1:] = (s[1:] != s[:-1]) | (d[1:] != d[:-1])
        s, d = s[first], d[first]
    rowptr = np.zeros(n_nodes + 1, dtype=np.int64)
    np.add.at(rowptr, s + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, d.astype(np.int32)


def padded_from_csr(rowptr, col, n_nodes, max_degree, rng):
    """The reference's padded table (minibatch.py:227-259) from a CSR, vectorised:
    deg > max_degree -> sample max_degree WITHOUT replacement; 0 < deg < max_degree -> WITH
    replacement; deg == 0 -> all pad (= n_nodes).  Returns (adj [N+1, max_degree] int32, deg)."""
    deg = np.diff(rowptr)
    adj = np.full((n_nodes + 1, max_degree), n_nodes, dtype=np.int32)
    small = np.where((deg > 0) & (deg <= max_degree))[0]
    if small.size:
        pick = (rng.random_sample((small.size, max_degree)) * deg[small][:, None]).astype(np.int64)
        exact = deg[small] == max_degree
        if exact.any():
            pick[exact] = np.arange(max_degree)[None, :]
        adj[small] = col[rowptr[small][:, None] + pick]
    big = np.where(deg > max_degree)[0]
    for i in big:  # few rows in practice (only hubs)
        adj[i] = rng.choice(col[rowptr[i]:rowptr[i + 1]], max_degree, replace=False)
    return adj, deg.astype(np.int64)


def synthetic_graph(n_nodes=2000, feat_dim=50, num_classes=7, avg_degree=10, seed=123, multilabel=False,
                    p_in=0.8, val_frac=0.1, test_frac=0.2, feat_noise=1.0, power_law=True, dtype=np.float32,
                    feat_signal=0.5):
    """Planted-community graph with learnable labels (SURVEY.md §8d config 2/3 shape: power-law degrees,
    labels = communities, features = community centroid + noise)."""
    rng = np.random.RandomState(seed)
    comm = rng.randint(0, num_classes, size=n_nodes)
    if power_law:
        w = (rng.pareto(2.0, size=n_nodes) + 1.0)
        w = np.minimum(w, np.percentile(w, 99.9))
    else:
        w = np.ones(n_nodes)
    n_edges = int(n_nodes * avg_degree / 2)
    p = w / w.sum()
    src = rng.choice(n_nodes, size=n_edges, p=p)
    # destination: same community with prob p_in, else anywhere (degree-weighted)
    dst = rng.choice(n_nodes, size=n_edges, p=p)
    same = rng.random_sample(n_edges) < p_in
    order = np.argsort(comm, kind="stable")
    starts = np.searchsorted(comm[order], np.arange(num_classes))
    ends = np.searchsorted(comm[order], np.arange(num_classes), side="right")
    cs = comm[src]
    span = np.maximum(ends[cs] - starts[cs], 1)
    dst_same = order[starts[cs] + (rng.random_sample(n_edges) * span).astype(np.int64) % span]
    dst = np.where(same, dst_same, dst)
    keep = src != dst
    src, dst = src[keep].astype(np.int32), dst[keep].astype(np.int32)
    centroids = rng.normal(size=(num_classes, feat_dim)).astype(np.float32)
    # float32 generation in place (no float64 temporaries: 561 MB tables page-fault slowly)
    g32 = np.random.default_rng(seed + 1)
    feats = g32.standard_normal((n_nodes, feat_dim), dtype=np.float32)
    if feat_noise != 1.0:
        feats *= np.float32(feat_noise)
    for a in range(0, n_nodes, 16384):
        feats[a:a + 16384] += centroids[comm[a:a + 16384]] * np.float32(feat_signal)
    r = rng.random_sample(n_nodes)
    val_mask = r < val_frac
    test_mask = (r >= val_frac) & (r < val_frac + test_frac)
    feats = standardize_on_train(feats, ~(val_mask | test_mask))
    if multilabel:
        proj = rng.normal(size=(num_classes, num_classes))
        labels = ((np.eye(num_classes)[comm] @ proj + 0.3 * rng.normal(size=(n_nodes, num_classes))) > 0.3)
        labels = labels.astype(np.float32)
    else:
        labels = comm.astype(np.int64)
    return GraphData(n_nodes, src, dst, feats, labels, val_mask, test_mask, multilabel=multilabel)


def reddit_shaped(avg_degree=50, seed=123, n_nodes=232965, feat_dim=602, num_classes=41, feat_signal=0.5):
    """Synthetic graph with Reddit's shape: N=232,965, F=602, C=41 single-label, 66/10/24 split."""
    return synthetic_graph(n_nodes=n_nodes, feat_dim=feat_dim, num_classes=num_classes, avg_degree=avg_degree,
                           seed=seed, val_frac=0.10, test_frac=0.24, feat_signal=feat_signal)


def load_data(prefix, normalize=True, load_walks=False):
    """Reader for the reference's on-disk format (utils.py:19-75): <prefix>-G.json (node-link),
    -id_map.json, -class_map.json, -feats.npy [, -walks.txt].  Returns a GraphData whose rows are the id map's
    indices; `.id_map` / `.node_ids` translate between original node ids and rows, `.present` marks the nodes kept
    in the graph (nodes without 'val'/'test' annotations are removed, utils.py:43-50) and, with load_walks,
    `.walks` holds the co-occurrence pairs of -walks.txt mapped to rows (minibatch.py:116-118; pairs naming a node
    that is not in the graph are dropped, minibatch.py:64-66)."""
    G = json.load(open(prefix + "-G.json"))
    id_map = json.load(open(prefix + "-id_map.json"))
    class_map = json.load(open(prefix + "-class_map.json"))
    n = len(id_map)
    id_map = {str(k): int(v) for k, v in id_map.items()}
    nodes = G["nodes"]
    val_mask = np.zeros(n, dtype=bool)
    test_mask = np.zeros(n, dtype=bool)
    present = np.zeros(n, dtype=bool)
    node_ids = [None] * n
    for k, v in id_map.items():
        node_ids[v] = k
    link_ids = []
    broken_count = 0
    for nd in nodes:
        link_ids.append(nd["id"])
        i = id_map[str(nd["id"])]
        node_ids[i] = nd["id"]                  # keep the original type (int ids stay ints in val.txt)
        if 'val' not in nd or 'test' not in nd:
            broken_count += 1                   # G.remove_node(node) (utils.py:46-49)
            continue
        present[i] = True
        val_mask[i] = bool(nd["val"])
        test_mask[i] = bool(nd["test"])
    print("Removed {:d} nodes that lacked proper annotations due to networkx versioning issues".format(broken_count))
    # networkx<=1.11 node-link JSON: "source"/"target" are positions in the "nodes" list
    src = np.array([id_map[str(link_ids[l["source"]])] for l in G["links"]], dtype=np.int32)
    dst = np.array([id_map[str(link_ids[l["target"]])] for l in G["links"]], dtype=np.int32)
    if src.size:
        keep = present[src] & present[dst]      # removing a node removes its edges
        src, dst = src[keep], dst[keep]
    if os.path.exists(prefix + "-feats.npy"):
        feats = np.load(prefix + "-feats.npy").astype(np.float32)
    else:
        print("No features present.. Only identity features will be used.")       # utils.py:41-43
        feats = None
    first = next(iter(class_map.values()))
    multilabel = isinstance(first, list)
    if multilabel:
        labels = np.zeros((n, len(first)), dtype=np.float32)
        for k, v in class_map.items():
            labels[id_map[str(k)]] = v
    else:
        labels = np.zeros(n, dtype=np.int64)
        for k, v in class_map.items():
            labels[id_map[str(k)]] = int(v)
    if normalize and feats is not None:
        feats = standardize_on_train(feats, present & ~(val_mask | test_mask))   # utils.py:62-68
    data = GraphData(n, src, dst, feats, labels, val_mask, test_mask, multilabel=multilabel, present=present,
                     id_map=id_map, node_ids=node_ids)
    if load_walks:
        data.walks = load_walk_pairs(prefix + "-walks.txt", data)
    return data


def load_walk_pairs(path, G):
    """<prefix>-walks.txt (one "node1 node2" pair of ORIGINAL node ids per line, utils.py:70-74) -> int32 [n, 2] rows
    of the id map (minibatch.py:116-118).  Pairs naming an unknown or removed node are dropped (minibatch.py:64-66)."""
    pairs = []
    missing = 0
    id_map = G.id_map
    with open(path) as fp:
        for line in fp:
            tok = line.split()
            if len(tok) < 2:
                continue
            a = id_map.get(tok[0]) if id_map is not None else int(tok[0])
            b = id_map.get(tok[1]) if id_map is not None else int(tok[1])
            if a is None or b is None or not (0 <= a < G.n_nodes and 0 <= b < G.n_nodes) \
                    or not (G.present[a] and G.present[b]):
                missing += 1
                continue
            pairs.append((a, b))
    print("Unexpected missing:", missing)
    return np.asarray(pairs, dtype=np.int32).reshape(-1, 2)


WALK_LEN = 5      # utils.py:16
N_WALKS = 50      # utils.py:17


def run_random_walks(rowptr, col, nodes, num_walks=N_WALKS, walk_len=WALK_LEN, seed=123, max_pairs=None):
    """Co-occurrence pairs from uniform random walks (utils.py:77-92), vectorised over all walks:
    for each start node and each of num_walks walks, walk walk_len steps and emit (start, current) whenever the
    current node differs from the start.  Runs on the TRAIN subgraph CSR (the reference takes G.subgraph(train
    nodes), :101-103).  Returns int32 [n_pairs, 2]."""
    rng = np.random.RandomState(seed)
    nodes = np.asarray(nodes, dtype=np.int64)
    deg = np.diff(rowptr)
    nodes = nodes[deg[nodes] > 0]                       # :80-81
    if max_pairs is not None:
        per_node = max(1, (walk_len - 1) * num_walks)
        keep = max(1, min(len(nodes), int(np.ceil(max_pairs / per_node))))
        nodes = rng.choice(nodes, size=keep, replace=False)
    start = np.repeat(nodes, num_walks)
    cur = start.copy()
    out = []
    for j in range(walk_len):
        if j > 0:
            m = cur != start                            # self co-occurrences are useless (:87-88)
            out.append(np.stack([start[m], cur[m]], axis=1))
        d = deg[cur]
        nxt = col[rowptr[cur] + (rng.random_sample(cur.shape[0]) * d).astype(np.int64)]
        cur = np.where(d > 0, nxt, cur)
    pairs = np.concatenate(out, axis=0).astype(np.int32) if out else np.zeros((0, 2), np.int32)
    if max_pairs is not None and len(pairs) > max_pairs:
        pairs = pairs[rng.permutation(len(pairs))[:max_pairs]]
    return pairs


def rmat_csr_device(n_nodes, n_edges, device, abcd=(0.57, 0.19, 0.19, 0.05), seed=123):
    """R-MAT graph (Chakrabarti et al.) generated and turned into CSR entirely in HBM: BASELINE configs[4] is
    N=10^7 / E=2*10^8 directed edges, far beyond what the NumPy generators above build in reasonable time.  Each of the
    ceil(log2 N) levels picks a quadrant with probabilities (a, b, c, d); ids are folded into [0, N) by modulo;
    duplicates are kept ("dedup off": the sampler draws with replacement from the neighbor list anyway).
    Returns (rowptr int64 [N+1], col int32 [E]) device tensors.  torch is used as plumbing for synthetic data only."""
    import torch
    a, b, c, _ = abcd
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    levels = max(1, int(np.ceil(np.log2(max(n_nodes, 2)))))
    src = torch.zeros(n_edges, dtype=torch.int64, device=device)
    dst = torch.zeros(n_edges, dtype=torch.int64, device=device)
    for _ in range(levels):
        r = torch.rand(n_edges, device=device, generator=g)
        src_bit = (r >= a + b)
        dst_bit = ((r >= a) & (r < a + b)) | (r >= a + b + c)
        src.mul_(2).add_(src_bit)
        dst.mul_(2).add_(dst_bit)
        del r, src_bit, dst_bit
    src.remainder_(n_nodes)
    dst.remainder_(n_nodes)
    src, perm = torch.sort(src)
    col = dst[perm].to(torch.int32)
    del dst, perm
    counts = torch.bincount(src, minlength=n_nodes)
    rowptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, dim=0, out=rowptr[1:])
    return rowptr, col


class DeviceGraph(object):
    """A synthetic Reddit-shaped dataset generated and kept in HBM (see reddit_shaped_device): CSR train / test views,
    the [N+1, ld] feature table (row N = zero pad row), the [N+1, C] label table, split masks.  `host_view()` copies what
    the CPU baseline needs back to the host."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def host_view(self, max_degree=128, seed=123):
        """(features [N+1, F], train padded adjacency, test padded adjacency, label matrix [N+1, C]) as NumPy arrays --
        the inputs of the torch-CPU port of the reference graph (minibatch.py:227-259 tables from the same edges)."""
        rng = np.random.RandomState(seed)
        feats = self.feats.numpy()
        rp, col = self.train_csr[0].cpu().numpy(), self.train_csr[1].cpu().numpy()
        adj, _ = padded_from_csr(rp, col, self.n_nodes, max_degree, rng)
        rp, col = self.test_csr[0].cpu().numpy(), self.test_csr[1].cpu().numpy()
        test_adj, _ = padded_from_csr(rp, col, self.n_nodes, max_degree, rng)
        return feats, adj, test_adj, self.label_table.numpy()


def reddit_shaped_device(device, n_nodes=232965, feat_dim=602, num_classes=41, avg_degree=492, seed=123, p_in=0.8,
                         val_frac=0.10, test_frac=0.24, feat_signal=0.5):
    """The planted-community generator of synthetic_graph()/reddit_shaped() run ON THE DEVICE (torch as plumbing for
    synthetic data only, like rmat_csr_device): Reddit's real average degree (~492) means 57 M undirected
    edges, which the NumPy generator needs minutes for.  Same construction: power-law (Pareto-2, capped) endpoint
    weights, destination in the source's community with probability p_in, one edge per node pair (networkx.Graph),
    66/10/24 split, features = community centroid * feat_signal + N(0,1) standardised on the train rows
    (utils.py:62-68), train view = edges between train nodes (utils.py:55-60 + minibatch.py:232-236).
    Returns a DeviceGraph."""
    import torch
    from .ops import Mat
    N, C = int(n_nodes), int(num_classes)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    comm = torch.randint(0, C, (N,), device=device, generator=g)
    u = torch.rand(N, device=device, generator=g, dtype=torch.float64)
    w = (1.0 - u).clamp_(min=1e-12).pow(-0.5)                               # Pareto(2) + 1
    w = torch.minimum(w, torch.quantile(w, 0.999))
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    n_edges = int(N * avg_degree / 2)

    def draw(n):
        return torch.searchsorted(cdf, torch.rand(n, device=device, generator=g, dtype=torch.float64)).clamp_(max=N - 1)

    src = draw(n_edges)
    dst = draw(n_edges)
    order = torch.argsort(comm, stable=True)
    counts = torch.bincount(comm, minlength=C)
    starts = torch.cumsum(counts, 0) - counts
    cs = comm[src]
    span = counts[cs].clamp(min=1)
    pick = (torch.rand(n_edges, device=device, generator=g, dtype=torch.float64) * span).long()
    dst_same = order[starts[cs] + torch.minimum(pick, span - 1)]
    same = torch.rand(n_edges, device=device, generator=g) < p_in
    dst = torch.where(same, dst_same, dst)
    del dst_same, same, pick, span, cs
    keep = src != dst
    src, dst = src[keep], dst[keep]
    del keep
    key = torch.cat([src * N + dst, dst * N + src])                         # symmetric, one entry per ordered pair
    del src, dst
    key = torch.unique(key)                                                 # sorted by (row, col); duplicates dropped
    s, d = key // N, key % N
    del key
    r = torch.rand(N, device=device, generator=g)
    val_mask = r < val_frac
    test_mask = (r >= val_frac) & (r < val_frac + test_frac)
    nt = val_mask | test_mask

    def csr(rows, cols):
        rowptr = torch.zeros(N + 1, dtype=torch.int64, device=device)
        torch.cumsum(torch.bincount(rows, minlength=N), 0, out=rowptr[1:])
        return rowptr, cols.to(torch.int32).contiguous()

    test_csr = csr(s, d)
    tr = ~(nt[s] | nt[d])
    train_csr = csr(s[tr], d[tr])
    del s, d, tr
    deg = (train_csr[0][1:] - train_csr[0][:-1])
    train_nodes = torch.nonzero(~nt & (deg > 0)).reshape(-1).to(torch.int32)
    # features, standardised on the train rows, written straight into the padded [N+1, ld] table
    feats = Mat.zeros(N + 1, feat_dim, device, ld_multiple=32)
    centroids = torch.randn(C, feat_dim, device=device, generator=g)
    x = feats.buf[:N, :feat_dim]
    x.normal_(generator=g)
    x += centroids[comm] * float(feat_signal)
    xt = x[~nt]
    mean = xt.mean(dim=0, dtype=torch.float64)
    std = (xt.to(torch.float64) - mean).pow_(2).mean(dim=0).sqrt_()
    del xt
    std[std == 0] = 1.0
    x -= mean.to(torch.float32)
    x *= (1.0 / std).to(torch.float32)
    label_table = Mat.zeros(N + 1, C, device)
    label_table.buf[torch.arange(N, device=device), comm] = 1.0
    torch.cuda.synchronize() if device.type == "cuda" else None
    deg_np = deg.cpu().numpy().astype(np.int64)
    deg_np[nt.cpu().numpy()] = 0
    return DeviceGraph(n_nodes=N, num_classes=C, feat_dim=int(feat_dim), feats=feats, label_table=label_table, labels=comm,
                       val_mask=val_mask, test_mask=test_mask, train_csr=train_csr, test_csr=test_csr,
                       train_nodes=train_nodes.cpu().numpy(), val_nodes=torch.nonzero(val_mask).reshape(-1).to(torch.int32).cpu().numpy(),
                       deg=deg_np, n_edges_undirected=int(test_csr[1].numel() // 2), avg_degree=avg_degree)


def random_walk_pairs_device(rowptr, col, nodes, num_walks=N_WALKS, walk_len=WALK_LEN, seed=123, max_pairs=None):
    """run_random_walks() on the device (CSR tensors in HBM): (start, current) co-occurrence pairs of uniform walks
    (utils.py:77-92).  Returns an int32 [n_pairs, 2] device tensor."""
    import torch
    device = rowptr.device
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    nodes = torch.as_tensor(nodes, device=device).to(torch.int64)
    deg = rowptr[1:] - rowptr[:-1]
    nodes = nodes[deg[nodes] > 0]
    if max_pairs is not None:
        per_node = max(1, (walk_len - 1) * num_walks)
        keep = max(1, min(int(nodes.numel()), int(np.ceil(max_pairs / per_node))))
        nodes = nodes[torch.randperm(nodes.numel(), device=device, generator=g)[:keep]]
    start = nodes.repeat_interleave(num_walks)
    cur = start.clone()
    out = []
    for j in range(walk_len):
        if j > 0:
            m = cur != start
            out.append(torch.stack([start[m], cur[m]], dim=1))
        dcur = deg[cur]
        off = (torch.rand(cur.numel(), device=device, generator=g, dtype=torch.float64) * dcur).long()
        off = torch.minimum(off, (dcur - 1).clamp(min=0))
        nxt = col[(rowptr[cur] + off).clamp(max=col.numel() - 1)].to(torch.int64)
        cur = torch.where(dcur > 0, nxt, cur)
    pairs = torch.cat(out, dim=0).to(torch.int32)
    if max_pairs is not None and pairs.shape[0] > max_pairs:
        pairs = pairs[torch.randperm(pairs.shape[0], device=device, generator=g)[:max_pairs]]
    return pairs
