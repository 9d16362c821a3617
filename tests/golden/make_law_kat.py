"""Generates tests/golden/law_kat.npz: known answers of the CSR sampler's sampling LAWS (GS_LAW_REFERENCE /
GS_LAW_DISTINCT: gs_perm_index, gs_table_entry, gs_draw in graphsage_amd/csrc/gs_sample_dev.h) computed with
arbitrary-precision Python ints and plain loops, independent of NumPy's uint64 arithmetic and of
oracle/sampler_hash.py's vectorised cycle walking.

The reference has no fixtures and cannot run here (TF 1.x); what is pinned is the NEW sampler's integer stream, whose
LAW restates minibatch.py:227-245 (padded table: choice without replacement above max_degree, with replacement
below) and neigh_samplers.py:24-29 (one column permutation per call shared by all rows).

    python tests/golden/make_law_kat.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
M64 = (1 << 64) - 1
G, R = 0x9E3779B97F4A7C15, 0xD1342543DE82EF95


def mix64(z):
    z &= M64
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & M64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & M64
    z ^= z >> 31
    return z


M32 = (1 << 32) - 1


def fmix32(h):
    h &= M32
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M32
    h ^= h >> 16
    return h


def perm_index(key, j, n):
    if n <= 1:
        return 0
    bits = max(2, (n - 1).bit_length())
    a = bits >> 1
    b = bits - a
    mA, mB = (1 << a) - 1, (1 << b) - 1
    rounds = 24 if bits <= 4 else (12 if bits <= 6 else 8)
    k0, k1 = key & M32, (key >> 32) & M32
    x = j
    while True:
        L, Rr = x >> b, x & mB
        for r in range(rounds):
            kr = (k0 + r * 0x9E3779B9) & M32
            if r % 2 == 0:
                L ^= ((fmix32((Rr + kr) & M32) ^ k1) >> 7) & mA
            else:
                Rr ^= ((fmix32((L + kr) & M32) ^ k1) >> 7) & mB
        x = (L << b) | Rr
        if x < n:
            return x


def table_key(seed, v):
    return mix64(seed ^ 0x7AB1E5EED ^ ((v * R) & M64))


def table_entry(tkey, c, deg, M):
    if deg > M:
        return perm_index(tkey, c, deg)
    if deg == M:
        return c
    return ((mix64((tkey + c) & M64) >> 32) * deg) >> 32


def draw(law, M, seed, callkey, grow, j, s, node, deg):
    if law == 1:
        c = perm_index(mix64(callkey ^ 0xC0115), j, M)
        return table_entry(table_key(seed, node), c, deg, M)
    rowkey = (callkey + grow * R) & M64
    if law == 2:
        eff = M if (M > 0 and deg > M) else deg
        c = perm_index(mix64(rowkey), j, eff) if eff >= s else ((mix64((rowkey + j) & M64) >> 32) * eff) >> 32
        return perm_index(table_key(seed, node), c, deg) if eff != deg else c
    return ((mix64((rowkey + j) & M64) >> 32) * deg) >> 32


def main():
    # graph: degrees 0, 3, 8 (== M), 20 (> M), 5; max_degree M = 8, pad id = n_nodes = 5
    degs = [0, 3, 8, 20, 5]
    rowptr = [0]
    for d in degs:
        rowptr.append(rowptr[-1] + d)
    col = [(7 * k + 3) % 5 for k in range(rowptr[-1])]
    # make node 3's list distinct values so that "distinct" can be checked by eye: ids 100..119 are not nodes, fine for col
    for k in range(rowptr[3], rowptr[4]):
        col[k] = 100 + (k - rowptr[3])
    ids = [3, 0, 2, 1, 4, 3, 5]
    M, s, seed, step, hop, row_off, pad = 8, 4, 123, 7, 1, 10, 5
    callkey = mix64(seed ^ ((step * G) & M64) ^ (hop << 56))
    out = {}
    for law in (1, 2):
        for cap in ((M,) if law == 1 else (M, 0)):
            rows = []
            for i, node in enumerate(ids):
                deg = degs[node] if node < 5 else 0
                rows.append([col[rowptr[node] + draw(law, cap, seed, callkey, i + row_off, j, s, node, deg)] if deg else pad
                             for j in range(s)])
            out["picked_law%d_cap%d" % (law, cap)] = np.array(rows, np.int32)
    perm_cases = [(0xDEADBEEF0BADF00D, n) for n in (1, 2, 5, 8, 37, 128, 129, 1000)]
    perms = {("perm_%d" % n): np.array([perm_index(k, j, n) for j in range(n)], np.int64) for k, n in perm_cases}
    for n, p in perms.items():
        assert sorted(p.tolist()) == list(range(len(p))), n
    np.savez(os.path.join(HERE, "law_kat.npz"), rowptr=np.array(rowptr, np.int64), col=np.array(col, np.int32),
             ids=np.array(ids, np.int32), args=np.array([M, s, seed, step, hop, row_off, pad], np.int64),
             perm_key=np.uint64(0xDEADBEEF0BADF00D), **out, **perms)
    for k, v in out.items():
        print(k, v.tolist())


if __name__ == "__main__":
    main()
