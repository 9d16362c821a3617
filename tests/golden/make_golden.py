"""Generates tests/golden/tiny_mean.npz: a hand-checkable fixture (N=6, F=4, fan-out 2x2, integer-valued
features and weights so every mean/matmul is exact in fp32), with expected outputs computed by plain
Python loops written independently of oracle/graphsage_oracle.py (no NumPy broadcasting tricks).

The reference ships no golden vectors and cannot be executed here (TensorFlow 1.x is not installable),
so this fixture pins the ORACLE to the arithmetic of aggregators.py:43-64 / models.py:254-330 by hand.

    python tests/golden/make_golden.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    N, F, max_deg = 6, 4, 3
    # padded adjacency, pad id = N = 6 (minibatch.py:228); node 5 has no train neighbors
    adj = np.array([[1, 2, 2], [0, 3, 3], [0, 4, 0], [1, 1, 1], [2, 2, 5], [6, 6, 6], [6, 6, 6]], dtype=np.int32)
    feats = np.array([[1, 0, 2, -1], [0, 2, 2, 4], [4, -2, 0, 2], [2, 2, -4, 0], [-2, 4, 6, 2], [8, 0, 0, -4],
                      [0, 0, 0, 0]], dtype=np.float32)  # row 6 = zero pad row (supervised_train.py:135)
    batch = np.array([0, 4, 5], dtype=np.int32)
    ns = [2, 2]                    # samples_1 (outer hop), samples_2 (hop next to the batch)
    perms = [np.array([2, 0, 1]), np.array([1, 2, 0])]
    dims = [4, 2, 2]
    # integer weights: layer0 [4,2] x2, layer1 [4,2] x2 (concat doubles the input dim, models.py:305)
    W0_self = np.array([[1, 0], [0, 1], [1, 1], [-1, 0]], dtype=np.float32)
    W0_neigh = np.array([[2, -2], [0, 2], [-2, 0], [2, 2]], dtype=np.float32)
    W1_self = np.array([[1, 0], [0, 1], [1, -1], [0, 1]], dtype=np.float32)
    W1_neigh = np.array([[2, 0], [0, 2], [-2, 2], [2, 0]], dtype=np.float32)

    # ---- S1/S2 with plain loops: out[i][j] = adj[ids[i]][perm[j]], j < num_samples
    def sampler(ids, num_samples, perm):
        return [[int(adj[i][perm[j]]) for j in range(num_samples)] for i in ids]

    s1 = sampler(batch, ns[1], perms[0])                      # k=0 uses layer_infos[1].num_samples (t = K-1-k)
    samples1 = [v for row in s1 for v in row]
    s2 = sampler(samples1, ns[0], perms[1])
    samples2 = [v for row in s2 for v in row]

    def mean_agg(self_rows, neigh_groups, Ws, Wn, relu):
        out = []
        for sv, group in zip(self_rows, neigh_groups):
            d = len(sv)
            mean = [sum(g[c] for g in group) / float(len(group)) for c in range(d)]
            o = Ws.shape[1]
            fs = [sum(sv[c] * Ws[c][k] for c in range(d)) for k in range(o)]
            fn = [sum(mean[c] * Wn[c][k] for c in range(d)) for k in range(o)]
            v = fs + fn                                     # concat [from_self, from_neighs] (aggregators.py:58)
            out.append([max(x, 0.0) for x in v] if relu else v)
        return out

    X = feats.tolist()
    B = len(batch)
    h_self0 = [X[i] for i in batch]
    h_self1 = [X[i] for i in samples1]
    g0 = [[X[samples1[i * ns[1] + j]] for j in range(ns[1])] for i in range(B)]
    g1 = [[X[samples2[i * ns[0] + j]] for j in range(ns[0])] for i in range(B * ns[1])]
    l0_hop0 = mean_agg(h_self0, g0, W0_self.tolist() and W0_self, W0_neigh, True)
    l0_hop1 = mean_agg(h_self1, g1, W0_self, W0_neigh, True)
    g_l1 = [[l0_hop1[i * ns[1] + j] for j in range(ns[1])] for i in range(B)]
    out = mean_agg(l0_hop0, g_l1, W1_self, W1_neigh, False)   # last layer: identity act (models.py:307-310)

    np.savez(os.path.join(HERE, "tiny_mean.npz"), adj=adj, feats=feats, batch=batch, num_samples=np.array(ns),
             perm0=perms[0], perm1=perms[1], dims=np.array(dims), W0_self=W0_self, W0_neigh=W0_neigh,
             W1_self=W1_self, W1_neigh=W1_neigh, samples1=np.array(samples1, dtype=np.int32),
             samples2=np.array(samples2, dtype=np.int32), l0_hop0=np.array(l0_hop0, dtype=np.float32),
             l0_hop1=np.array(l0_hop1, dtype=np.float32), out=np.array(out, dtype=np.float32))
    print("samples1", samples1)
    print("samples2", samples2)
    print("out", out)


if __name__ == "__main__":
    main()
