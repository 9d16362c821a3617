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


def sample_uniform_csr(rowptr, col, n_nodes, pad_id, ids, num_samples, seed, step, hop, global_row_offset=0):
    ids = np.asarray(ids, dtype=np.int64)
    n = ids.shape[0]
    with np.errstate(over="ignore"):
        key = mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(step) * _G) ^ (np.uint64(hop) << np.uint64(56)))
        i = (np.arange(n, dtype=np.uint64) + np.uint64(global_row_offset))[:, None]
        j = np.arange(num_samples, dtype=np.uint64)[None, :]
        u = mix64(key + i * _R + j)
    r = (u >> np.uint64(32)).astype(np.uint64)
    valid = (ids >= 0) & (ids < n_nodes)
    safe = np.where(valid, ids, 0)
    beg = np.where(valid, rowptr[safe], 0).astype(np.int64)
    deg = np.where(valid, rowptr[safe + 1] - rowptr[safe], 0).astype(np.int64)
    k = ((r * deg[:, None].astype(np.uint64)) >> np.uint64(32)).astype(np.int64)
    pos = beg[:, None] + k
    has = deg[:, None] > 0
    pos = np.where(has, pos, 0)
    picked = np.asarray(col)[pos] if len(col) else np.zeros_like(pos)
    out = np.where(has, picked, pad_id).astype(np.int32)
    return out


def unigram_cdf_u32(degrees, distortion=0.75):
    """Fixed-point CDF of tf.nn.fixed_unigram_candidate_sampler(unigrams=degrees, distortion=0.75)
    (models.py:336-343): cdf[i] = floor(2^32 * P(node <= i)), last entry forced to 2^32-1."""
    w = np.power(np.asarray(degrees, dtype=np.float64), distortion)
    c = np.cumsum(w) / w.sum()
    cdf = np.minimum(np.floor(c * 4294967296.0), 4294967295.0).astype(np.uint32)
    cdf[-1] = np.uint32(4294967295)
    return cdf


def sample_unigram(cdf, n_neg, seed, clock):
    """Restatement of the negative draw of gs_unsup_stage: slot t -> first index whose cdf exceeds a 32-bit hash."""
    with np.errstate(over="ignore"):
        key = mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(clock) * _G) ^ (np.uint64(0xFF) << np.uint64(56)))
        r = (mix64(key + np.arange(n_neg, dtype=np.uint64)) >> np.uint64(32)).astype(np.uint64)
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
