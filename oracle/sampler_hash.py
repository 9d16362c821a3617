"""CPU restatement (numpy uint64) of the CSR sampler of graphsage_amd/csrc/gs_sample.hip.

TEST INFRASTRUCTURE ONLY (see oracle/graphsage_oracle.py header).  The reference has no CSR
sampler: its sampler is the padded-table one (neigh_samplers.py:24-29, restated in
graphsage_oracle.uniform_neighbor_sampler).  This file pins the NEW sampler's integer stream so
the HIP kernel can be checked bit-exactly, and so tests can check its distribution against the
reference's per-slot marginal (uniform over the neighbor set; pad id for degree-0 rows,
minibatch.py:228,238-239).
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)
_R = np.uint64(0xD1342543DE82EF95)


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z ^ (z >> np.uint64(30))
        z = z * _M1
        z = z ^ (z >> np.uint64(27))
        z = z * _M2
        z = z ^ (z >> np.uint64(31))
    return z


LAW_IID, LAW_REFERENCE, LAW_DISTINCT = 0, 1, 2      # GS_LAW_* of include/graphsage_amd.h


def fmix32(h):
    """murmur3 finalizer on uint32 arrays (gs_fmix32)."""
    h = np.asarray(h).astype(np.uint32)
    with np.errstate(over="ignore"):
        h = h ^ (h >> np.uint32(16))
        h = h * np.uint32(0x85EBCA6B)
        h = h ^ (h >> np.uint32(13))
        h = h * np.uint32(0xC2B2AE35)
        h = h ^ (h >> np.uint32(16))
    return h


def perm_index(key, j, n):
    """gs_perm_index (gs_sample_dev.h): j-th element of the keyed pseudo-random permutation of [0, n) -- alternating
    Feistel on bits = max(2, ceil(log2 n)) bits (high part floor(bits/2), low part the rest; even rounds hash the low part
    into the high part, odd rounds the reverse; 8 rounds, 12 for bits <= 6, 24 for bits <= 4; 32-bit arithmetic),
    cycle-walked back into [0, n).  key (uint64) / j / n broadcast; returns uint64."""
    u32 = np.uint32
    key = np.asarray(key, dtype=np.uint64)
    k0 = (key & np.uint64(0xFFFFFFFF)).astype(u32)
    k1 = (key >> np.uint64(32)).astype(u32)
    k0, k1, j, n = np.broadcast_arrays(k0, k1, np.asarray(j).astype(u32), np.asarray(n).astype(u32))
    nm1 = np.maximum(n, u32(1)) - u32(1)
    bits = np.zeros(n.shape, dtype=u32)
    t = nm1.copy()
    while (t > 0).any():                       # bits = bit_length(n - 1)
        bits += (t > 0).astype(u32)
        t = t >> u32(1)
    bits = np.maximum(bits, u32(2))
    a = bits >> u32(1)
    b = bits - a
    mA = (u32(1) << a) - u32(1)
    mB = (u32(1) << b) - u32(1)
    rounds = np.where(bits <= 4, 24, np.where(bits <= 6, 12, 8))
    todo = n > u32(1)
    # a walk started outside [0, n) need not come back: callers mask such lanes, clamp them here
    x = np.where(todo, np.minimum(j, nm1), u32(0)).astype(u32)
    while todo.any():
        L, R = x >> b, x & mB
        with np.errstate(over="ignore"):
            for r in range(24):
                live = r < rounds
                kr = k0 + u32(r) * u32(0x9E3779B9)
                if r % 2 == 0:
                    L = np.where(live, L ^ (((fmix32(R + kr) ^ k1) >> u32(7)) & mA), L)
                else:
                    R = np.where(live, R ^ (((fmix32(L + kr) ^ k1) >> u32(7)) & mB), R)
        y = ((L << b) | R).astype(u32)
        x = np.where(todo, y, x)
        todo = todo & (x >= n)
    return x.astype(np.uint64)


def _table_key(seed, v):
    with np.errstate(over="ignore"):
        return mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ np.uint64(0x7AB1E5EED)
                     ^ ((np.asarray(v, dtype=np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF)) * _R))


def table_entry(tkey, c, deg, M):
    """gs_table_entry: entry c of a node's VIRTUAL padded row (minibatch.py:240-243) -> position in its neighbor list."""
    tkey, c, deg = np.broadcast_arrays(np.asarray(tkey, dtype=np.uint64), np.asarray(c, dtype=np.uint64),
                                       np.asarray(deg, dtype=np.uint64))
    M = np.uint64(M)
    with np.errstate(over="ignore"):
        iid = ((mix64(tkey + c) >> np.uint64(32)) * deg) >> np.uint64(32)
    out = np.where(deg == M, c, iid)
    big = deg > M
    if big.any():
        out = out.copy()
        out[big] = perm_index(tkey[big], c[big], deg[big])
    return out


def virtual_padded_table(rowptr, col, n_nodes, pad_id, seed, max_degree, nodes=None):
    """The [len(nodes), max_degree] rows of the virtual padded table GS_LAW_REFERENCE samples from (the analogue of
    NodeMinibatchIterator.construct_adj's `adj`); row of a degree-0 node = all pad."""
    nodes = np.arange(n_nodes) if nodes is None else np.asarray(nodes)
    beg = np.asarray(rowptr)[nodes].astype(np.int64)
    deg = (np.asarray(rowptr)[nodes + 1] - beg).astype(np.int64)
    c = np.arange(max_degree, dtype=np.uint64)[None, :]
    k = table_entry(_table_key(seed, nodes)[:, None], c, np.maximum(deg, 1)[:, None], max_degree).astype(np.int64)
    picked = np.asarray(col)[np.where(deg[:, None] > 0, beg[:, None] + k, 0)]
    return np.where(deg[:, None] > 0, picked, pad_id).astype(np.int32)


def call_columns(seed, step, hop, num_samples, max_degree):
    """The num_samples distinct columns GS_LAW_REFERENCE uses for call (step, hop): head of ONE keyed permutation of
    [0, max_degree) shared by all rows (neigh_samplers.py:27-28)."""
    with np.errstate(over="ignore"):
        key = mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(step) * _G) ^ (np.uint64(hop) << np.uint64(56)))
        ck = mix64(key ^ np.uint64(0xC0115))
    return perm_index(ck, np.arange(num_samples, dtype=np.uint64), np.uint64(max_degree)).astype(np.int64)


def sample_uniform_csr(rowptr, col, n_nodes, pad_id, ids, num_samples, seed, step, hop, global_row_offset=0,
                       law=LAW_IID, max_degree=0):
    ids = np.asarray(ids, dtype=np.int64)
    n = ids.shape[0]
    valid = (ids >= 0) & (ids < n_nodes)
    safe = np.where(valid, ids, 0)
    beg = np.where(valid, rowptr[safe], 0).astype(np.int64)
    deg = np.where(valid, rowptr[safe + 1] - rowptr[safe], 0).astype(np.int64)
    degu = np.maximum(deg, 1)[:, None].astype(np.uint64)
    with np.errstate(over="ignore"):
        key = mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(step) * _G) ^ (np.uint64(hop) << np.uint64(56)))
        i = (np.arange(n, dtype=np.uint64) + np.uint64(global_row_offset))[:, None]
        j = np.arange(num_samples, dtype=np.uint64)[None, :]
        rowkey = key + i * _R
        if law == LAW_REFERENCE:
            assert 0 < num_samples <= max_degree
            cols = call_columns(seed, step, hop, num_samples, max_degree).astype(np.uint64)[None, :]
            k = table_entry(_table_key(seed, safe)[:, None], cols, degu, max_degree)
        elif law == LAW_DISTINCT:
            M = np.uint64(max_degree)
            capped = (degu > M) if max_degree > 0 else np.zeros_like(degu, dtype=bool)
            eff = np.where(capped, M, degu)
            iid = ((mix64(rowkey + j) >> np.uint64(32)) * eff) >> np.uint64(32)
            wor = perm_index(mix64(rowkey), j, eff)
            c = np.where(eff >= np.uint64(num_samples), wor, iid)
            k = np.where(capped, perm_index(_table_key(seed, safe)[:, None], c, degu), c)
        else:
            k = ((mix64(rowkey + j) >> np.uint64(32)) * degu) >> np.uint64(32)
    k = k.astype(np.int64)
    pos = beg[:, None] + k
    has = deg[:, None] > 0
    pos = np.where(has, pos, 0)
    picked = np.asarray(col)[pos] if len(col) else np.zeros_like(pos)
    out = np.where(has, picked, pad_id).astype(np.int32)
    return out


def sample_uniform_csr_segments(rowptr, col, n_nodes, pad_id, ids, num_samples, seed, step, hop, n_hops, seg_rows,
                                global_row_offset=0, law=LAW_REFERENCE, max_degree=0):
    """One hop of the UNSUPERVISED pass over the roots [batch1 | batch2 | negatives] under GS_LAW_REFERENCE: the reference
    makes three sample() calls per step (models.py:347-357) and every sampler call shuffles its own columns
    (neigh_samplers.py:27), so the rows of segment g (row boundaries `seg_rows` = (start of batch2's rows, start of the
    negatives' rows) at THIS hop) use the call id g * n_hops + hop -- six independent permutations per step
    (gs_fanout_desc.seg_begin).  The other laws draw per row and ignore the segments."""
    ids = np.asarray(ids)
    if law != LAW_REFERENCE or seg_rows is None:
        return sample_uniform_csr(rowptr, col, n_nodes, pad_id, ids, num_samples, seed, step, hop, global_row_offset, law,
                                  max_degree)
    bounds = [0, int(seg_rows[0]), int(seg_rows[1]), len(ids)]
    parts = []
    for g in range(3):
        lo, hi = bounds[g], bounds[g + 1]
        parts.append(sample_uniform_csr(rowptr, col, n_nodes, pad_id, ids[lo:hi], num_samples, seed, step, g * n_hops + hop,
                                        global_row_offset + lo, law, max_degree))
    return np.concatenate(parts, axis=0)


def unigram_cdf_u32(degrees, distortion=0.75):
    """Fixed-point CDF of tf.nn.fixed_unigram_candidate_sampler(unigrams=degrees, distortion=0.75)
    (models.py:336-343): cdf[i] = floor(2^32 * P(node <= i)), last entry forced to 2^32-1."""
    w = np.power(np.asarray(degrees, dtype=np.float64), distortion)
    c = np.cumsum(w) / w.sum()
    cdf = np.minimum(np.floor(c * 4294967296.0), 4294967295.0).astype(np.uint32)
    cdf[-1] = np.uint32(4294967295)
    return cdf


def sample_unigram(cdf, n_neg, seed, clock, slot_offset=0):
    """Restatement of the negative draw of gs_unsup_stage: slot t -> first index whose cdf exceeds a 32-bit hash.
    `slot_offset`: the fan-out sampler's staging keys slot t by root_offset + t (data-parallel ranks draw different
    negatives); 0 on one GPU."""
    with np.errstate(over="ignore"):
        key = mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(clock) * _G) ^ (np.uint64(0xFF) << np.uint64(56)))
        r = (mix64(key + np.arange(n_neg, dtype=np.uint64) + np.uint64(slot_offset)) >> np.uint64(32)).astype(np.uint64)
    idx = np.searchsorted(cdf.astype(np.uint64), r, side="right")
    return np.minimum(idx, len(cdf) - 1).astype(np.int32)


def dropout_mask(seed, clock, site, row0, n_rows, d, rate):
    """Scaled keep mask ({0, 1/(1-rate)}, float32 [n_rows, d]) of the device dropout (gs_common.h: gs_drop_key /
    gs_drop4): element (row, col) is kept iff the 16-bit field (col % 4) of
    mix64(key + (row0 + row) * R + col // 4) is >= round(rate * 2^16).  The reference draws its masks from TF's RNG
    (tf.nn.dropout, aggregators.py:46-47); like the sampler's permutations they are injected into the oracle so that
    both sides see identical masks."""
    rate32 = np.float32(rate)
    thresh = int(np.float32(rate32 * np.float32(65536.0) + np.float32(0.5)))
    thresh = min(max(thresh, 1), 65535)
    scale = np.float32(1.0) / (np.float32(1.0) - rate32)
    with np.errstate(over="ignore"):
        key = mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(clock) * _G) ^ (np.uint64(site) << np.uint64(32))
                    ^ (np.uint64(0xD0) << np.uint64(56)))
        rows = (np.arange(n_rows, dtype=np.uint64) + np.uint64(row0))[:, None]
        q = np.arange((d + 3) // 4, dtype=np.uint64)[None, :]
        h = mix64(key + rows * _R + q)
    bits = np.stack([(h >> np.uint64(16 * e)) & np.uint64(0xFFFF) for e in range(4)], axis=-1)
    keep = bits.reshape(n_rows, -1)[:, :d] >= np.uint64(thresh)
    return keep.astype(np.float32) * scale
