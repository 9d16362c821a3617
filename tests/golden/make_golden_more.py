"""Generates tests/golden/tiny_gcn_maxpool.npz and tests/golden/hash_kat.npz.

* tiny_gcn_maxpool: the tiny_mean graph (N=6, F=4, fan-out 2x2) pushed through ONE GCN layer call and ONE MaxPool
  layer call with integer weights, computed with plain Python loops written independently of
  oracle/graphsage_oracle.py (aggregators.py:101-116 and :168-195 + layers.py:104-116), exact in fp32.
* hash_kat: known answers of the counter hash shared by the CSR sampler, the unigram negative sampler and dropout,
  computed with arbitrary-precision Python ints (independent of NumPy's uint64 wrap-around arithmetic).

The reference has no fixtures and cannot run here (TF 1.x): these pin the ORACLE / hash restatements by hand.

    python tests/golden/make_golden_more.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
M64 = (1 << 64) - 1


def mix64(z):
    z &= M64
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & M64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & M64
    z ^= z >> 31
    return z


def main():
    g = np.load(os.path.join(HERE, "tiny_mean.npz"))
    X = g["feats"].tolist()
    batch = g["batch"].tolist()
    samples1 = g["samples1"].tolist()
    s = 2
    self_rows = [X[i] for i in batch]
    groups = [[X[samples1[i * s + j]] for j in range(s)] for i in range(len(batch))]

    # ---- GCN (aggregators.py:101-116): mean over neighbors AND self, one weight, relu
    W_gcn = [[1, -1, 0], [0, 2, 1], [2, 0, -1], [-1, 1, 1]]
    gcn = []
    for sv, grp in zip(self_rows, groups):
        means = [(sum(r[c] for r in grp) + sv[c]) / float(len(grp) + 1) for c in range(4)]
        out = [sum(means[c] * W_gcn[c][k] for c in range(4)) for k in range(3)]
        gcn.append([max(v, 0.0) for v in out])

    # ---- MaxPool (aggregators.py:168-195): relu(Dense) over every neighbor, max over neighbors, two matmuls, concat
    W_mlp = [[1, 0, -1], [0, 1, 1], [1, -1, 0], [0, 1, -1]]
    b_mlp = [0, -1, 2]
    W_self = [[1, 0], [0, 1], [1, 1], [-1, 0]]
    W_neigh = [[1, -1], [2, 0], [0, 1]]
    mp = []
    for sv, grp in zip(self_rows, groups):
        h = [[max(sum(r[c] * W_mlp[c][k] for c in range(4)) + b_mlp[k], 0.0) for k in range(3)] for r in grp]
        pooled = [max(hr[k] for hr in h) for k in range(3)]
        fn = [sum(pooled[k] * W_neigh[k][o] for k in range(3)) for o in range(2)]
        fs = [sum(sv[c] * W_self[c][o] for c in range(4)) for o in range(2)]
        mp.append([max(v, 0.0) for v in fs + fn])
    np.savez(os.path.join(HERE, "tiny_gcn_maxpool.npz"), W_gcn=np.array(W_gcn, np.float32), gcn_out=np.array(gcn, np.float32),
             W_mlp=np.array(W_mlp, np.float32), b_mlp=np.array(b_mlp, np.float32), W_self=np.array(W_self, np.float32),
             W_neigh=np.array(W_neigh, np.float32), maxpool_out=np.array(mp, np.float32))

    # ---- hash known answers
    G, R = 0x9E3779B97F4A7C15, 0xD1342543DE82EF95
    mix_in = [0, 1, 0xDEADBEEF, M64, 123456789012345678]
    mix_out = [mix64(v) for v in mix_in]
    # CSR sampler (gs_sample.hip): key = mix64(seed ^ step*G ^ hop<<56); u = mix64(key + row*R + j); pos = (u>>32)*deg>>32
    rowptr = [0, 3, 3, 8, 9]
    col = [1, 2, 3, 0, 1, 2, 3, 3, 0]
    ids, ns_, seed, step, hop, row_off = [0, 1, 2, 3, 2], 4, 123, 7, 1, 10
    key = mix64(seed ^ ((step * G) & M64) ^ (hop << 56))
    picked = []
    for i, node in enumerate(ids):
        deg = rowptr[node + 1] - rowptr[node]
        row = []
        for j in range(ns_):
            u = mix64((key + (i + row_off) * R + j) & M64)
            row.append(col[rowptr[node] + (((u >> 32) * deg) >> 32)] if deg else 4)   # pad id = n_nodes = 4
        picked.append(row)
    # dropout (gs_common.h): key = mix64(seed ^ clock*G ^ site<<32 ^ 0xD0<<56); h = mix64(key + (row0+r)*R + q);
    # element e of float4 q kept iff ((h >> 16e) & 0xffff) >= round(rate * 65536)
    dseed, clock, site, row0, n_rows, d, rate = 77, 5, 19, 1000, 3, 10, 0.25
    thresh = int(rate * 65536 + 0.5)
    dkey = mix64(dseed ^ ((clock * G) & M64) ^ (site << 32) ^ (0xD0 << 56))
    keep = []
    for r in range(n_rows):
        row = []
        for c in range(d):
            h = mix64((dkey + (row0 + r) * R + c // 4) & M64)
            row.append(1 if ((h >> (16 * (c % 4))) & 0xFFFF) >= thresh else 0)
        keep.append(row)
    np.savez(os.path.join(HERE, "hash_kat.npz"), mix_in=np.array(mix_in, np.uint64), mix_out=np.array(mix_out, np.uint64),
             rowptr=np.array(rowptr, np.int64), col=np.array(col, np.int32), ids=np.array(ids, np.int32),
             csr_args=np.array([ns_, seed, step, hop, row_off, 4], np.int64), picked=np.array(picked, np.int32),
             drop_args=np.array([dseed, clock, site, row0, n_rows, d], np.int64), drop_rate=np.float32(rate),
             keep=np.array(keep, np.int32))
    print("gcn", gcn)
    print("maxpool", mp)
    print("picked", picked)
    print("keep", keep)


if __name__ == "__main__":
    main()
