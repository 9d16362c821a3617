"""Typed Python wrappers over the C ABI (one function per entry point of include/graphsage_amd.h).

`Mat` is a row-major fp32 device matrix with a padded leading dimension (what every kernel expects:
16-byte aligned base, ld % 4 == 0).  torch only owns the memory.
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT_IDENTITY, ACT_RELU, call, ptr  # noqa: F401


def round_up(x, m):
    return (x + m - 1) // m * m


class Mat(object):
    """fp32 [rows, d] matrix stored in a [rows, ld] buffer, ld % 4 == 0 (pad columns are zero)."""

    __slots__ = ("buf", "d")

    def __init__(self, buf, d):
        assert buf.dim() == 2 and buf.dtype == torch.float32 and buf.stride(1) == 1
        assert buf.stride(0) % 4 == 0 and buf.data_ptr() % 16 == 0 and d <= buf.shape[1]
        self.buf = buf
        self.d = d

    @staticmethod
    def zeros(rows, d, device, ld_multiple=4):
        ld = round_up(max(d, 1), ld_multiple)
        return Mat(torch.zeros((rows, ld), dtype=torch.float32, device=device), d)

    @staticmethod
    def from_numpy(a, device, ld_multiple=4):
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim == 1:
            a = a[None, :]
        m = Mat.zeros(a.shape[0], a.shape[1], device, ld_multiple)
        m.buf[:, : a.shape[1]].copy_(torch.from_numpy(a))
        return m

    @property
    def rows(self):
        return self.buf.shape[0]

    @property
    def ld(self):
        return self.buf.stride(0)

    @property
    def ptr(self):
        return ptr(self.buf)

    def view(self):
        """Logical [rows, d] view (torch tensor)."""
        return self.buf[:, : self.d]

    def numpy(self):
        return self.view().detach().cpu().numpy()

    def rows_slice(self, r0, r1):
        return Mat(self.buf[r0:r1], self.d)

    def cols_slice(self, c0, c1):
        """Columns [c0, c1) as a matrix with the same leading dimension (c0 % 4 == 0)."""
        assert c0 % 4 == 0 and c1 <= self.buf.shape[1]
        return Mat(self.buf[:, c0:c1], c1 - c0)


def current_stream():
    return torch.cuda.current_stream().cuda_stream


def _s(stream):
    return current_stream() if stream is None else stream


# ------------------------------------------------------------------------------------------ K1
def sample_padded(adj, ids, col_perm, num_samples, out=None, stream=None):
    n = ids.numel()
    if out is None:
        out = torch.empty((n * num_samples,), dtype=torch.int32, device=ids.device)
    call("gs_sample_padded", ptr(adj), adj.shape[0], adj.shape[1], ptr(ids), n, ptr(col_perm), num_samples,
         ptr(out), _s(stream))
    return out


def sample_uniform_csr(rowptr, col, n_nodes, pad_id, ids, num_samples, seed, step=0, step_dev=None, hop=0,
                       global_row_offset=0, out=None, stream=None, law=0, max_degree=0):
    n = ids.numel()
    if out is None:
        out = torch.empty((n * num_samples,), dtype=torch.int32, device=ids.device)
    call("gs_sample_uniform_csr", ptr(rowptr), ptr(col), n_nodes, pad_id, ptr(ids), n, num_samples,
         seed & 0xFFFFFFFFFFFFFFFF, step, ptr(step_dev), hop, global_row_offset, law, max_degree, ptr(out), _s(stream))
    return out


def sample_fanout_csr(rowptr, col, n_nodes, pad_id, fans, offsets, ids_all, B, seed, step_dev=None, hop0=0,
                      root_offset=0, order=None, cursor_dev=None, label_table=None, labels_out=None, stream=None,
                      law=0, max_degree=0):
    """Fused multi-hop sampler (+ optional batch/label staging); see gs_sample_fanout_csr."""
    import ctypes
    fan = (ctypes.c_int32 * len(fans))(*fans)
    off = (ctypes.c_int64 * len(offsets))(*offsets)
    call("gs_sample_fanout_csr", ptr(rowptr), ptr(col), n_nodes, pad_id, len(fans), ctypes.addressof(fan),
         ctypes.addressof(off), ptr(ids_all), B, seed & 0xFFFFFFFFFFFFFFFF, 0, ptr(step_dev), hop0, root_offset,
         ptr(order), order.numel() if order is not None else 0, ptr(cursor_dev),
         label_table.ptr if label_table is not None else None, label_table.ld if label_table is not None else 0,
         label_table.d if label_table is not None else 0, labels_out.ptr if labels_out is not None else None,
         labels_out.ld if labels_out is not None else 0, law, max_degree, _s(stream))


def fanout_desc(rowptr, col, n_nodes, pad_id, fans, offsets, ids_all, B, seed, step_dev=None, hop0=0, root_offset=0,
                order=None, cursor_dev=None, label_table=None, labels_out=None, law=0, max_degree=0, unsup=None,
                padded_table=None, segments=None):
    """unsup = (pairs [n_pairs, 2] int32, n_pair_roots, cdf (uint32 bits), guide or None, guide_bits, n_neg, neg_seed): the
    roots are staged by the launch itself as [pairs[:, 0] | pairs[:, 1] | negatives] (see gs_fanout_desc).
    The arguments of sample_fanout_csr as a struct gs_fanout_desc (for gs_flat_reduce_adam_sample).  The tensors are
    kept alive on the descriptor object."""
    q = _lib.FanoutDesc()
    q.rowptr, q.col, q.n_nodes, q.pad_id = ptr(rowptr), ptr(col), n_nodes, pad_id
    q.n_hops = len(fans)
    for k, f in enumerate(fans):
        q.fan[k] = f
    for k, o in enumerate(offsets[: len(fans) + 1]):
        q.offsets[k] = o
    q.ids_all, q.B, q.seed, q.step, q.step_dev = ptr(ids_all), B, seed & 0xFFFFFFFFFFFFFFFF, 0, ptr(step_dev)
    q.hop0, q.root_offset = hop0, root_offset
    q.law, q.max_degree = law, max_degree
    q.order, q.n_order, q.cursor_dev = ptr(order), (order.numel() if order is not None else 0), ptr(cursor_dev)
    if label_table is not None:
        q.label_table, q.ld_table, q.C = label_table.ptr, label_table.ld, label_table.d
    if labels_out is not None:
        q.labels_out, q.ld_out = labels_out.ptr, labels_out.ld
    q.padded_table = ptr(padded_table)
    if segments is not None:       # root boundaries of the reference's three sample() calls (gs_fanout_desc.seg_begin)
        q.seg_begin[0], q.seg_begin[1] = int(segments[0]), int(segments[1])
    q._keep = (rowptr, col, ids_all, step_dev, order, cursor_dev, label_table, labels_out, unsup, padded_table)
    if unsup is not None:
        pairs, n_pair_roots, cdf, guide, guide_bits, n_neg, neg_seed = unsup
        q.pairs, q.n_pairs, q.n_pair_roots = ptr(pairs), pairs.shape[0], n_pair_roots
        q.cdf, q.guide, q.n_cdf = ptr(cdf), ptr(guide), cdf.numel()
        q.n_neg, q.guide_bits, q.neg_seed = n_neg, guide_bits, neg_seed & 0xFFFFFFFFFFFFFFFF
    return q


def build_padded_table(rowptr, col, n_nodes, pad_id, max_degree, seed, stream=None):
    """The reference law's padded table [n_nodes + 1, max_degree] (int32, device) from the CSR (gs_build_padded_table)."""
    table = torch.empty((n_nodes + 1) * max_degree, dtype=torch.int32, device=rowptr.device)
    call("gs_build_padded_table", ptr(rowptr), ptr(col), n_nodes, pad_id, max_degree, seed & 0xFFFFFFFFFFFFFFFF, ptr(table),
         _s(stream))
    return table


def sample_fanout_desc(desc, stream=None):
    """Launch the fused fan-out sampler from a descriptor built by fanout_desc()."""
    call("gs_sample_fanout_desc", ctypes.addressof(desc), _s(stream))


def select_batch(order, cursor_dev, n, out, stream=None):
    call("gs_select_batch", ptr(order), order.numel(), ptr(cursor_dev), n, ptr(out), _s(stream))
    return out


def advance_counter(counter_dev, delta, stream=None):
    call("gs_advance_counter", ptr(counter_dev), delta, _s(stream))


# ------------------------------------------------------------------------------------------ K2
def gather_rows(X, ids, out=None, stream=None):
    n = ids.numel()
    if out is None:
        out = Mat.zeros(n, X.d, ids.device)
    call("gs_gather_rows", X.ptr, X.ld, ptr(ids), n, X.d, out.ptr, out.ld, _s(stream))
    return out


def input_grad_pull(out, rows, d, d_self=None, n_self=0, segments=(), mask_y=None, stream=None):
    """One-launch input gradient of a layer (gs_input_grad_pull).  segments: (src Mat, row0, n, s, scale) tuples:
    out[row0 + i*s + j] += scale * src[i]."""
    q = _lib.PullDesc()
    q.d_self, q.ld_self, q.n_self = (d_self.ptr, d_self.ld, n_self) if d_self is not None else (None, 0, 0)
    assert len(segments) <= _lib.GS_PULL_MAX
    q.n_seg, q.d = len(segments), d
    for k, (src, row0, n, s, scale) in enumerate(segments):
        q.src[k], q.ld_src[k], q.row0[k], q.n[k], q.s[k], q.scale[k] = src.ptr, src.ld, row0, n, s, scale
    q.mask_y, q.ldy = (mask_y.ptr, mask_y.ld) if mask_y is not None else (None, 0)
    q.out, q.ldo, q.rows = out.ptr, out.ld, rows
    call("gs_input_grad_pull", ctypes.addressof(q), _s(stream))
    return out


def dropout_desc(seed, clock_dev, site, rate, row0=0, keep=None):
    """struct gs_dropout; None when rate == 0 (dropout off).  keep: uint8 device tensor [rows, ld] of injected keep bits
    (parity tests: the masks the reference run drew), addressed by the call's global row index."""
    if not rate:
        return None
    d = _lib.Dropout(int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(clock_dev), int(site), float(rate), int(row0), None, 0)
    if keep is not None:
        assert keep.dtype == torch.uint8 and keep.dim() == 2 and keep.stride(1) == 1 and keep.stride(0) % 4 == 0
        d.keep_bits, d.keep_ld = keep.data_ptr(), keep.stride(0)
        d._keep = keep
    return d


def dropout_rows(X, ids, n, drop, out, stream=None):
    """out[i] = mask * X[ids[i] or i] / keep_prob (tf.nn.dropout); also its own backward.  In place when ids is None."""
    call("gs_dropout_rows", X.ptr, X.ld, ptr(ids), n, X.d, ctypes.addressof(drop) if drop is not None else None,
         out.ptr, out.ld, _s(stream))
    return out


def gather_mean_fwd(X, idx, n, s, out=None, self_src=None, self_idx=None, drop=None, stream=None):
    """idx: int32 [n*s] or None (contiguous groups).  self_src (Mat) switches to the GCN mean.  `drop`: dropout of
    every gathered neighbor row before the mean."""
    if out is None:
        out = Mat.zeros(n, X.d, X.buf.device)
    if drop is not None:
        call("gs_gather_mean_dropout_fwd", X.ptr, X.ld, ptr(idx), n, s, X.d,
             self_src.ptr if self_src is not None else None, self_src.ld if self_src is not None else 0,
             ptr(self_idx), out.ptr, out.ld, ctypes.addressof(drop), _s(stream))
        return out
    call("gs_gather_mean_fwd", X.ptr, X.ld, ptr(idx), n, s, X.d,
         self_src.ptr if self_src is not None else None, self_src.ld if self_src is not None else 0,
         ptr(self_idx), out.ptr, out.ld, _s(stream))
    return out


def mean_bwd(d_mean, n, s, scale, d_neigh, mask_y=None, accumulate=False, stream=None):
    call("gs_mean_bwd", d_mean.ptr, d_mean.ld, n, s, d_mean.d, scale,
         mask_y.ptr if mask_y is not None else None, mask_y.ld if mask_y is not None else 0,
         d_neigh.ptr, d_neigh.ld, 1 if accumulate else 0, _s(stream))
    return d_neigh


# ------------------------------------------------------------------------------------------ K3
def sage_dense_fwd(self_m, self_idx, agg, agg_idx, n, W_self, W_neigh, out_dim, concat, act, bias, out,
                   stream=None):
    call("gs_sage_dense_fwd",
         self_m.ptr if self_m is not None else None, self_m.ld if self_m is not None else 0, ptr(self_idx),
         self_m.d if self_m is not None else 0,
         agg.ptr, agg.ld, ptr(agg_idx), agg.d, n,
         W_self.ptr if W_self is not None else None, W_self.ld if W_self is not None else 0,
         W_neigh.ptr, W_neigh.ld, out_dim, 1 if concat else 0, act, ptr(bias), out.ptr, out.ld, _s(stream))
    return out


def gather_job(X, idx, n, s, out, self_src=None, self_idx=None):
    """Descriptor of one gather+mean job for sage_dense_fwd_cogather (same arguments as gather_mean_fwd)."""
    j = _lib.GatherDesc()
    j.X, j.idx, j.out = X.ptr, ptr(idx), out.ptr
    j.self_src = self_src.ptr if self_src is not None else None
    j.self_idx = ptr(self_idx)
    j.ldx, j.ld_self, j.ldo, j.n, j.s, j.d = X.ld, (self_src.ld if self_src is not None else 0), out.ld, n, s, X.d
    return j


def split_gather_jobs(jobs, frac):
    """Split a gather job list into (head, tail): `tail` gets the last (1 - frac) of the ROWS of the largest job (its
    idx / out / self pointers advanced accordingly), `head` everything else.  Used to spread one step's gather over
    two horizontally fused launches."""
    if not jobs or frac >= 1.0:
        return list(jobs or ()), []
    big = max(range(len(jobs)), key=lambda i: jobs[i].n * jobs[i].s)
    j = jobs[big]
    n_head = int(j.n * max(frac, 0.0))
    if n_head >= j.n:
        return list(jobs), []
    tail = _lib.GatherDesc()
    for name, _ in _lib.GatherDesc._fields_:
        setattr(tail, name, getattr(j, name))
    tail.n = j.n - n_head
    tail.idx = (j.idx + 4 * n_head * j.s) if j.idx else None
    tail.out = j.out + 4 * n_head * j.ldo
    if j.self_src:
        if j.self_idx:
            tail.self_idx = j.self_idx + 4 * n_head
        else:
            tail.self_src = j.self_src + 4 * n_head * j.ld_self
    head = list(jobs)
    if n_head == 0:
        del head[big]
    else:
        h = _lib.GatherDesc()
        for name, _ in _lib.GatherDesc._fields_:
            setattr(h, name, getattr(j, name))
        h.n = n_head
        head[big] = h
    return head, [tail]


def sage_dense_fwd_cogather(self_m, self_idx, agg, agg_idx, n, W_self, W_neigh, out_dim, concat, act, bias, out,
                            jobs, stream=None):
    """gs_sage_dense_fwd + the gather jobs in ONE horizontally fused launch."""
    import ctypes
    arr = (_lib.GatherDesc * max(len(jobs), 1))(*jobs)
    call("gs_sage_dense_fwd_cogather", self_m.ptr if self_m is not None else None, self_m.ld if self_m is not None else 0,
         ptr(self_idx), self_m.d if self_m is not None else 0, agg.ptr, agg.ld, ptr(agg_idx),
         agg.d, n, W_self.ptr if W_self is not None else None, W_self.ld if W_self is not None else 0,
         W_neigh.ptr, W_neigh.ld, out_dim, 1 if concat else 0, act, ptr(bias),
         out.ptr, out.ld, ctypes.addressof(arr), len(jobs), _s(stream))
    return out


def sage_dense_fwd_stream(self_m, self_idx, agg, n, W_self, W_neigh, out_dim, act, bias, out, jobs, stream=None):
    """gs_sage_dense_fwd_stream: split-K stream contraction workgroups + the gather jobs in ONE launch (self rows
    optionally gathered through self_idx; agg dense)."""
    import ctypes
    jobs = list(jobs or ())
    arr = (_lib.GatherDesc * max(len(jobs), 1))(*jobs)
    call("gs_sage_dense_fwd_stream", self_m.ptr if self_m is not None else None, self_m.ld if self_m is not None else 0,
         ptr(self_idx), agg.ptr, agg.ld, agg.d, n, W_self.ptr if W_self is not None else None,
         W_self.ld if W_self is not None else 0, W_neigh.ptr, W_neigh.ld, out_dim, act, ptr(bias), out.ptr, out.ld,
         ctypes.addressof(arr), len(jobs), _s(stream))
    return out


def sage_dense_fwd_stream2(self_m, self_idx, agg, n, W_self, W_neigh, out_dim, act, bias, out, jobs=(), stream=None):
    """gs_sage_dense_fwd_stream2: the stream contraction with different reduction lengths of the two terms (pooling aggregators:
    self rows of self_m.d features gathered through self_idx | pooled rows of agg.d = hidden_dim), concat output."""
    import ctypes
    jobs = list(jobs or ())
    arr = (_lib.GatherDesc * max(len(jobs), 1))(*jobs)
    call("gs_sage_dense_fwd_stream2", self_m.ptr, self_m.ld, ptr(self_idx), self_m.d, agg.ptr, agg.ld, agg.d, n, W_self.ptr, W_self.ld,
         W_neigh.ptr, W_neigh.ld, out_dim, act, ptr(bias), out.ptr, out.ld, ctypes.addressof(arr), len(jobs), _s(stream))
    return out


def sage_dense_fwd_tiled3(self_m, self_idx, agg, n, W_self, W_neigh, out_dim, act, bias, out, jobs=(), stream=None):
    """gs_sage_dense_fwd_tiled3: the layer-0 contraction LDS-tiled on the bf16 matrix pipe (fp32 operands cut into three bf16 pieces
    INSIDE the kernel: same arguments as sage_dense_fwd_stream2), concat output; self_m None = one term (GCN)."""
    import ctypes
    jobs = list(jobs or ())
    arr = (_lib.GatherDesc * max(len(jobs), 1))(*jobs)
    call("gs_sage_dense_fwd_tiled3", self_m.ptr if self_m is not None else None, self_m.ld if self_m is not None else 0,
         ptr(self_idx), self_m.d if self_m is not None else 0, agg.ptr, agg.ld, agg.d, n,
         W_self.ptr if W_self is not None else None, W_self.ld if W_self is not None else 0, W_neigh.ptr, W_neigh.ld, out_dim, act,
         ptr(bias), out.ptr, out.ld, ctypes.addressof(arr), len(jobs), _s(stream))
    return out


def split_rows_words(K, N):
    """int32 words of gs_split_rows' output (gs_split_rows_bytes / 4): groups of 8 k up to an even count of 32-k stages."""
    import ctypes
    n = ctypes.c_int64()
    call("gs_split_rows_bytes", int(K), int(N), ctypes.byref(n))
    return int(n.value) // 4


_SPLIT_WS_WORDS = None


def split_tiled_ws_words():
    """fp32 words of the workspace gs_dense_fwd_rows_split_ws wants (one 128 x 256 partial tile per CU)."""
    global _SPLIT_WS_WORDS
    if _SPLIT_WS_WORDS is None:
        import ctypes
        n = ctypes.c_int64()
        call("gs_dense_fwd_rows_split_ws_bytes", ctypes.byref(n))
        _SPLIT_WS_WORDS = int(n.value) // 4
    return _SPLIT_WS_WORDS


def split_rows(W, out=None, stream=None):
    """gs_split_rows: the three bf16 pieces of W^T ([groups of 8 k][3][N][8] bf16, as an int32 tensor) for the split-MFMA
    contractions; W is a Mat [K, N]."""
    K, N = W.rows, W.d
    if out is None:
        out = torch.empty(split_rows_words(K, N), dtype=torch.int32, device=W.buf.device)
    call("gs_split_rows", W.ptr, W.ld, K, N, ptr(out), _s(stream))
    return out


def split_rows_f16_words(K, N):
    """int32 words of gs_split_rows_f16's output: the two fp16 pieces of the scaled weights + the column exponents."""
    import ctypes
    n = ctypes.c_int64()
    call("gs_split_rows_f16_bytes", int(K), int(N), ctypes.byref(n))
    return int(n.value) // 4


def split_rows_f16(W, out=None, stream=None):
    """gs_split_rows_f16: W [K, N] as two fp16 pieces per element under a power-of-two scale per COLUMN ([groups of 8 k][2][N][8]
    fp16 + N int32 exponents, as an int32 tensor) for gs_dense_fwd_rows_split16."""
    K, N = W.rows, W.d
    if out is None:
        out = torch.empty(split_rows_f16_words(K, N), dtype=torch.int32, device=W.buf.device)
    call("gs_split_rows_f16", W.ptr, W.ld, K, N, ptr(out), _s(stream))
    return out


def split_table_f16(X, stream=None):
    """gs_split_table_f16: a (constant) feature table as two fp16 pieces per element under a power-of-two scale per ROW.
    Returns (X2 int32 tensor: [rows][2][KP] fp16, row exponents int32 [rows])."""
    import ctypes
    tb, eb = ctypes.c_int64(), ctypes.c_int64()
    call("gs_split_table_f16_bytes", X.rows, X.d, ctypes.byref(tb), ctypes.byref(eb))
    X2 = torch.empty(int(tb.value) // 4, dtype=torch.int32, device=X.buf.device)
    rexp = torch.empty(int(eb.value) // 4, dtype=torch.int32, device=X.buf.device)
    call("gs_split_table_f16", X.ptr, X.ld, X.rows, X.d, ptr(X2), ptr(rexp), _s(stream))
    return X2, rexp


def dense_wgrad(A, a_idx, dZ, col0, out_dim, n, n_slabs, slabs, ld_slab, stream=None):
    """slabs: flat fp32 tensor with room for n_slabs * A.d * ld_slab floats."""
    call("gs_dense_wgrad", A.ptr, A.ld, ptr(a_idx), A.d, dZ.ptr, dZ.ld, col0, out_dim, n, n_slabs, ptr(slabs),
         ld_slab, _s(stream))


def sage_dense_dgrad(dZ, n, out_dim, fwd_concat, W_self, W_neigh, d_in, dX2, stream=None):
    call("gs_sage_dense_dgrad", dZ.ptr, dZ.ld, n, out_dim, 1 if fwd_concat else 0, W_self.ptr, W_self.ld,
         W_neigh.ptr, W_neigh.ld, d_in, dX2.ptr, dX2.ld, _s(stream))
    return dX2


def stage_batch(order, cursor_dev, n, batch, label_table, labels_out, stream=None):
    call("gs_stage_batch", ptr(order), order.numel(), ptr(cursor_dev), n, ptr(batch), label_table.ptr, label_table.ld,
         label_table.d, labels_out.ptr, labels_out.ld, _s(stream))


def dense_dgrad(dZ, col0, out_dim, n, W, dX, accumulate=False, stream=None):
    call("gs_dense_dgrad", dZ.ptr, dZ.ld, col0, out_dim, n, W.ptr, W.ld, W.rows, dX.ptr, dX.ld,
         1 if accumulate else 0, _s(stream))
    return dX


def act_bwd(dY, Y, n, n_cols, act, dZ, stream=None):
    call("gs_act_bwd", dY.ptr, dY.ld, Y.ptr if Y is not None else None, Y.ld if Y is not None else 0, n, n_cols,
         act, dZ.ptr, dZ.ld, _s(stream))
    return dZ


def colsum_slabs(Z, n, n_cols, n_slabs, slabs, ld_slab, stream=None):
    call("gs_colsum_slabs", Z.ptr, Z.ld, n, n_cols, n_slabs, ptr(slabs), ld_slab, _s(stream))


def gemm(transA, transB, M, N, K, A, B, C, a_row_idx=None, bias=None, act=ACT_IDENTITY, stream=None):
    call("gs_gemm_f32", 1 if transA else 0, 1 if transB else 0, M, N, K, A.ptr, A.ld, ptr(a_row_idx), B.ptr, B.ld,
         ptr(bias), act, C.ptr, C.ld, _s(stream))
    return C


# ------------------------------------------------------------------------------------------ K4
def dense_pool_max_fwd(X, idx, n, s, W, bias, pooled, argmax, stream=None):
    """gs_dense_pool_max_fwd: pooled = max over each group's s rows of relu(X[idx] . W + bias), + arg-max."""
    call("gs_dense_pool_max_fwd", X.ptr, X.ld, ptr(idx), X.d, n, s, W.ptr, W.ld, W.d, ptr(bias), pooled.ptr, pooled.ld,
         ptr(argmax), argmax.stride(0), _s(stream))


def segment_max_fwd(H, n, s, pooled, argmax, stream=None):
    call("gs_segment_max_fwd", H.ptr, H.ld, n, s, H.d, pooled.ptr, pooled.ld, ptr(argmax), argmax.stride(0),
         _s(stream))


def segment_max_bwd(d_pooled, pooled, argmax, n, s, dH, stream=None):
    call("gs_segment_max_bwd", d_pooled.ptr, d_pooled.ld, pooled.ptr, pooled.ld, ptr(argmax), argmax.stride(0), n, s,
         pooled.d, dH.ptr, dH.ld, _s(stream))


def maxpool_sparse_wgrad(X, ids, n, s, argmax, dpm, hidden, n_slabs, slabs_ptr, ld_slab, stream=None):
    """dW slabs [n_slabs, X.d, ld_slab] of the max-pool MLP from the arg-max rows only (dH never materialised)."""
    call("gs_maxpool_sparse_wgrad", X.ptr, X.ld, ptr(ids), n, s, X.d, ptr(argmax), argmax.stride(0), dpm.ptr, dpm.ld,
         hidden, n_slabs, slabs_ptr, ld_slab, _s(stream))


def scatter_add_rows(d, n, s, cols, scale, ids, table, stream=None):
    """table[ids[i*s + j], :cols] += scale * d[i, :cols]  (gradient of a row gather w.r.t. the gathered table)."""
    call("gs_scatter_add_rows", d.ptr, d.ld, n, s, cols, scale, ptr(ids), table.ptr, table.ld, _s(stream))


def copy_cols(src, dst, rows, cols, stream=None):
    call("gs_copy_cols", src.ptr, src.ld, dst.ptr, dst.ld, rows, cols, _s(stream))


# ------------------------------------------------------------------------------------------ K5
def l2norm_fwd(x, n, y, inv_norm, stream=None):
    call("gs_l2norm_fwd", x.ptr, x.ld, n, x.d, y.ptr, y.ld, ptr(inv_norm), _s(stream))


def l2norm_bwd(dy, y, inv_norm, n, dx, stream=None):
    call("gs_l2norm_bwd", dy.ptr, dy.ld, y.ptr, y.ld, ptr(inv_norm), n, y.d, dx.ptr, dx.ld, _s(stream))


def class_loss(logits, labels, n, C, sigmoid_loss, loss_rows, preds=None, dlogits=None, stream=None):
    call("gs_class_loss", logits.ptr, logits.ld, labels.ptr, labels.ld, n, C, 1 if sigmoid_loss else 0,
         ptr(loss_rows), preds.ptr if preds is not None else None, preds.ld if preds is not None else 0,
         dlogits.ptr if dlogits is not None else None, dlogits.ld if dlogits is not None else 0, _s(stream))


def head_fwd_bwd(x, n, W, bias, labels, C, sigmoid_loss, y, logits, preds, dlogits, loss_rows, dx, stream=None):
    call("gs_head_fwd_bwd", x.ptr, x.ld, n, x.d, W.ptr, W.ld, ptr(bias), labels.ptr, labels.ld, C,
         1 if sigmoid_loss else 0, y.ptr, y.ld, logits.ptr if logits is not None else None,
         logits.ld if logits is not None else 0, preds.ptr if preds is not None else None,
         preds.ld if preds is not None else 0, dlogits.ptr, dlogits.ld, ptr(loss_rows),
         dx.ptr if dx is not None else None, dx.ld if dx is not None else 0, _s(stream))


def sage_tail_supported(d_in, out_dim, C):
    return bool(_lib.load().gs_sage_tail_supported(int(d_in), int(out_dim), int(C)))


def tail_sync_words(n, out_dim=128):
    """Size (uint32 words) of the hand-over buffer gs_sage_tail_fwd_bwd needs for n batch rows: epochs, the error word and
    16 x 2*out_dim eight-byte granules per 16-row group (gs_tail_desc.sync).  A buffer serves ONE n."""
    G = (n + 15) // 16
    return 2 * G + 2 + 64 * G * out_dim


def tail_sync_error(sync, n):
    """The error word of a tail hand-over buffer (0 = fine); synchronises."""
    return int(sync[2 * ((n + 15) // 16)].item())


def sage_tail_fwd_bwd(h0, n, s, W_self, W_neigh, out_dim, W_head, b_head, labels, C, sigmoid_loss, means, z, y, logits,
                      preds, dlogits, loss_rows, dz=None, d_h0=None, counters=(), jobs=(), stream=None, sync=None,
                      split=False, jobs_z=(), gcn=False, ids_copy=None):
    """gs_sage_tail_fwd_bwd: layer 1 + head (+ their input gradients when dz / d_h0 are given) in ONE launch.
    counters: up to three (device int64 tensor, delta) pairs advanced at the end of the launch.
    sync: int32 device tensor of tail_sync_words(n) words, zero-initialised once and owned by ONE caller / stream
    (kernel-internal hand-over state + an error word, see tail_sync_error); a fresh one is allocated if not given.
    split: two launches instead -- gs_sage_tail_z (lean z-helper kernel carrying the gather jobs `jobs_z`) and then this
    entry with z_ready (no helpers, no hand-over state) carrying `jobs`; same results bit for bit.
    ids_copy: (src int32 tensor, dst int32 tensor, count) -- the launch's helper workgroups copy the step's node ids into the
    private buffer the weight gradients gather through (gs_tail_desc.ids_copy_*)."""
    if sync is None and not split:
        import torch
        sync = torch.zeros(tail_sync_words(n, out_dim), dtype=torch.int32, device=h0.buf.device)
        torch.cuda.synchronize()
    assert split or sync.numel() >= tail_sync_words(n, out_dim)
    q = _lib.TailDesc()
    q._keep = sync
    q.sync = ptr(sync) if not split else None
    q.z_ready = 1 if split else 0
    q.h0, q.ldh, q.n = h0.ptr, h0.ld, n
    q.W_self, q.ldws, q.W_neigh, q.ldwn = W_self.ptr, W_self.ld, W_neigh.ptr, W_neigh.ld
    q.W_head, q.ldwh, q.b_head = W_head.ptr, W_head.ld, ptr(b_head)
    q.labels, q.ldlab = labels.ptr, labels.ld
    q.means, q.ldm, q.z, q.ldz, q.y, q.ldy = means.ptr, means.ld, z.ptr, z.ld, y.ptr, y.ld
    q.logits, q.ldlo = (logits.ptr, logits.ld) if logits is not None else (None, 0)
    q.preds, q.ldp = (preds.ptr, preds.ld) if preds is not None else (None, 0)
    q.dlogits, q.lddl, q.loss_rows = dlogits.ptr, dlogits.ld, ptr(loss_rows)
    train = dz is not None and d_h0 is not None
    q.dz, q.lddz = (dz.ptr, dz.ld) if train else (None, 0)
    q.d_h0, q.lddh = (d_h0.ptr, d_h0.ld) if train else (None, 0)
    cs = [(ptr(c), int(d)) for c, d in counters if c is not None and d]
    cs += [(None, 0)] * (3 - len(cs))
    (q.c0, q.d0), (q.c1, q.d1), (q.c2, q.d2) = cs[:3]
    q.s, q.d_in, q.out_dim, q.C, q.sigmoid, q.train = s, h0.d, out_dim, C, 1 if sigmoid_loss else 0, 1 if train else 0
    q.gcn = 1 if gcn else 0          # GCNAggregator layer 1: W_self / W_neigh are the two column halves of ONE weight matrix
    if ids_copy is not None:
        q.ids_copy_src, q.ids_copy_dst, q.ids_copy_n = ptr(ids_copy[0]), ptr(ids_copy[1]), int(ids_copy[2])
    if split:
        jz = list(jobs_z or ())
        jzarr = (_lib.GatherDesc * max(len(jz), 1))(*jz)
        call("gs_sage_tail_z", ctypes.addressof(q), ctypes.addressof(jzarr), len(jz), _s(stream))
    jobs = list(jobs or ())
    jarr = (_lib.GatherDesc * max(len(jobs), 1))(*jobs)
    call("gs_sage_tail_fwd_bwd", ctypes.addressof(q), ctypes.addressof(jarr), len(jobs), _s(stream))


def sage_tail_z(h0, n, s, W_self, W_neigh, out_dim, means, z, jobs=(), stream=None):
    """gs_sage_tail_z: z = [h0[:n] . W_self | mean_j(h0[n + i s + j]) . W_neigh] and the neighbor means of a LAST
    mean-aggregator layer (concat, identity act, no bias) in one lean launch (+ gather jobs riding at the full HBM rate)."""
    q = _lib.TailDesc()
    q.h0, q.ldh, q.n = h0.ptr, h0.ld, n
    q.W_self, q.ldws, q.W_neigh, q.ldwn = W_self.ptr, W_self.ld, W_neigh.ptr, W_neigh.ld
    q.means, q.ldm, q.z, q.ldz = means.ptr, means.ld, z.ptr, z.ld
    q.s, q.d_in, q.out_dim, q.C = s, h0.d, out_dim, 1
    q.z_ready = 1
    jobs = list(jobs or ())
    jarr = (_lib.GatherDesc * max(len(jobs), 1))(*jobs)
    call("gs_sage_tail_z", ctypes.addressof(q), ctypes.addressof(jarr), len(jobs), _s(stream))
    return z


def sage_tail_dh0(h0, n, s, W_self, W_neigh, out_dim, dz, d_h0, jobs=(), stream=None):
    """gs_sage_tail_dh0: d_h0 = relu'(h0) * ([dz[:, :O] . W_self^T on the self rows | dz[:, O:] . W_neigh^T / s on the
    neighbor rows]) of a last mean layer, one launch (+ gather jobs)."""
    q = _lib.TailDesc()
    q.h0, q.ldh, q.n = h0.ptr, h0.ld, n
    q.W_self, q.ldws, q.W_neigh, q.ldwn = W_self.ptr, W_self.ld, W_neigh.ptr, W_neigh.ld
    q.dz, q.lddz, q.d_h0, q.lddh = dz.ptr, dz.ld, d_h0.ptr, d_h0.ld
    q.s, q.d_in, q.out_dim, q.C = s, h0.d, out_dim, 1
    q.z_ready = 1
    jobs = list(jobs or ())
    jarr = (_lib.GatherDesc * max(len(jobs), 1))(*jobs)
    call("gs_sage_tail_dh0", ctypes.addressof(q), ctypes.addressof(jarr), len(jobs), _s(stream))
    return d_h0


def linkpred_tail_supported(d_in, out_dim, n_neg):
    return bool(_lib.load().gs_linkpred_tail_supported(int(d_in), int(out_dim), int(n_neg)))


def lp_tail_sync_words(B, n_neg):
    """Size (uint32 words) of the hand-over buffer of linkpred_tail for B pairs and n_neg negatives."""
    return 2 * ((B + 7) // 8 + (n_neg + 15) // 16) + 2


def lp_tail_sync_error(sync, B, n_neg):
    return int(sync[2 * ((B + 7) // 8 + (n_neg + 15) // 16)].item())


def linkpred_tail_desc(h0, B, n_neg, s, W_self, W_neigh, out_dim, means, z, y, loss_rows, rr_rows, aff_all, neg_weight, scale,
                       sync, dz=None, d_h0=None, neg_slabs=None):
    """struct gs_lp_tail_desc of the unsupervised fused tail (see include/graphsage_amd.h); dz / d_h0 / neg_slabs given =
    training (forward + backward), else forward only."""
    assert sync.numel() >= lp_tail_sync_words(B, n_neg)
    q = _lib.LpTailDesc()
    q._keep = (sync, neg_slabs)
    q.h0, q.ldh, q.B, q.n_neg, q.s, q.d_in, q.out_dim = h0.ptr, h0.ld, B, n_neg, s, h0.d, out_dim
    q.W_self, q.ldws, q.W_neigh, q.ldwn = W_self.ptr, W_self.ld, W_neigh.ptr, W_neigh.ld
    q.means, q.ldm, q.z, q.ldz, q.y, q.ldy = means.ptr, means.ld, z.ptr, z.ld, y.ptr, y.ld
    train = dz is not None and d_h0 is not None and neg_slabs is not None
    q.train = 1 if train else 0
    if train:
        q.dz, q.lddz, q.d_h0, q.lddh, q.neg_slabs = dz.ptr, dz.ld, d_h0.ptr, d_h0.ld, ptr(neg_slabs)
    q.loss_rows, q.rr_rows = ptr(loss_rows), ptr(rr_rows)
    q.aff_all, q.ld_aff = (aff_all.ptr, aff_all.ld) if aff_all is not None else (None, 0)
    q.neg_weight, q.scale, q.sync = float(neg_weight), float(scale), ptr(sync)
    return q


def linkpred_tail(desc, jobs=(), stream=None):
    """gs_linkpred_tail: launch 1 of the unsupervised fused tail (+ gather jobs riding)."""
    jobs = list(jobs or ())
    jarr = (_lib.GatherDesc * max(len(jobs), 1))(*jobs)
    call("gs_linkpred_tail", ctypes.addressof(desc), ctypes.addressof(jarr), len(jobs), _s(stream))


def linkpred_tail_neg(desc, loss_out=None, accumulate=False, mrr_out=None, counters=(), jobs=(), stream=None):
    """gs_linkpred_tail_neg: launch 2 (the negatives' rows, the step epilogue, the commit of the hand-over state) + gather jobs."""
    cs = [(ptr(c), int(d)) for c, d in counters if c is not None and d]
    cs += [(None, 0)] * (3 - len(cs))
    jobs = list(jobs or ())
    jarr = (_lib.GatherDesc * max(len(jobs), 1))(*jobs)
    call("gs_linkpred_tail_neg", ctypes.addressof(desc), ptr(loss_out), 1 if accumulate else 0, ptr(mrr_out),
         cs[0][0], cs[0][1], cs[1][0], cs[1][1], cs[2][0], cs[2][1], ctypes.addressof(jarr), len(jobs), _s(stream))


# ------------------------------------------------------------------------------------------ K6
def reduce_slabs(slabs, n_slabs, slab_stride, rows, cols, ld_slab, weight_decay, w_ptr, ldw, grad_ptr, ldg,
                 accumulate=False, stream=None):
    call("gs_reduce_slabs", ptr(slabs) if hasattr(slabs, "is_cuda") else slabs, n_slabs, slab_stride, rows, cols,
         ld_slab, weight_decay, w_ptr, ldw, grad_ptr, ldg, 1 if accumulate else 0, _s(stream))


def adam_step(p, grad, m, v, count, lr, step_dev, beta1=0.9, beta2=0.999, eps=1e-8, clip=5.0, grad_scale=1.0,
              step_offset=1, stream=None):
    """TF Adam with t = *step_dev + step_offset (0 when the step counter was already advanced this step)."""
    call("gs_adam_step", ptr(p), ptr(grad), ptr(m), ptr(v), count, lr, beta1, beta2, eps, clip, grad_scale,
         ptr(step_dev), int(step_offset), _s(stream))


def sum_scaled(x, count, scale, out, accumulate=False, stream=None):
    call("gs_sum_scaled", ptr(x), count, scale, ptr(out), 1 if accumulate else 0, _s(stream))


def sumsq_scaled(x, count, scale, out, accumulate=False, stream=None):
    call("gs_sumsq_scaled", ptr(x), count, scale, ptr(out), 1 if accumulate else 0, _s(stream))


# ------------------------------------------------------------------------------------------ graphs / events
class Stream(object):
    """A non-blocking HIP stream created by the library (capturable into a hipGraph)."""

    def __init__(self):
        import ctypes
        h = ctypes.c_void_p()
        call("gs_stream_create", ctypes.byref(h))
        self.handle = h.value

    def sync(self):
        call("gs_stream_sync", self.handle)

    def __del__(self):
        try:
            _lib.load().gs_stream_destroy(self.handle)
        except Exception:
            pass


class Graph(object):
    """hipGraph captured from the kernel chain enqueued between begin() and end()."""

    def __init__(self, stream):
        self.stream = stream
        self.exec_ = None

    def begin(self):
        call("gs_capture_begin", self.stream)

    def end(self):
        import ctypes
        h = ctypes.c_void_p()
        call("gs_capture_end", self.stream, ctypes.byref(h))
        self.exec_ = h.value

    def launch(self):
        call("gs_graph_launch", self.exec_, self.stream)

    def __del__(self):
        try:
            if self.exec_:
                _lib.load().gs_graph_destroy(self.exec_)
        except Exception:
            pass


class Event(object):
    def __init__(self):
        import ctypes
        h = ctypes.c_void_p()
        call("gs_event_create", ctypes.byref(h))
        self.handle = h.value

    def record(self, stream):
        call("gs_event_record", self.handle, stream)

    def wait(self, stream):
        """Make `stream` wait for this event (fork/join; a graph edge while capturing)."""
        call("gs_stream_wait_event", stream, self.handle)

    def elapsed_ms(self, stop):
        import ctypes
        ms = ctypes.c_float()
        call("gs_event_elapsed_ms", self.handle, stop.handle, ctypes.byref(ms))
        return ms.value

    def __del__(self):
        try:
            _lib.load().gs_event_destroy(self.handle)
        except Exception:
            pass


def build_csr_host(src, dst, n_nodes, symmetrize=True, keep_mask=None):
    """Host-side C++ CSR builder (gs_build_csr_host).  numpy in, numpy out."""
    import ctypes
    import numpy as np
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    if keep_mask is not None:
        keep_mask = np.ascontiguousarray(keep_mask, dtype=np.uint8)
    cap = int(src.size) * (2 if symmetrize else 1)
    rowptr = np.zeros((n_nodes + 1,), dtype=np.int64)
    col = np.zeros((max(cap, 1),), dtype=np.int32)
    nnz = ctypes.c_int64()
    call("gs_build_csr_host", _lib.host_ptr(src), _lib.host_ptr(dst), _lib.host_ptr(keep_mask), src.size, n_nodes,
         1 if symmetrize else 0, _lib.host_ptr(rowptr), _lib.host_ptr(col), cap, ctypes.byref(nnz))
    return rowptr, col[: nnz.value].copy()
