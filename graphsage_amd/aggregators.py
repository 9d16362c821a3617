"""Mean / GCN / MaxPooling / MeanPooling aggregators with the constructor and call signatures of
graphsage/aggregators.py, executing on the gfx950 kernels.

    agg = MeanAggregator(input_dim, output_dim, act=..., dropout=..., name=..., concat=..., model_size=...)
    out = agg((self_vecs, neigh_vecs))        # models.py:326-327

`self_vecs` is a `Rows` [n, d]; `neigh_vecs` is a `Rows` reshaped to [n, s, d] (both may be lazy row gathers of
the feature table, so the [n*s, d] tensor of models.py:299 is never materialised).  `.vars` holds exactly the
variables the reference's weight-decay loop sees (supervised_models.py:104-106).

MI355X-first addition: the reference calls the SAME aggregator once per hop of a layer (models.py:321-328);
`call_hops(self_all, [neigh_0, neigh_1, ...])` runs all hops of a layer in one dense launch (the rows of all
hops are contiguous), and `backward_hops` is its hand-written reverse.  `_call` is `call_hops` with one hop.
"""
import os

from . import ops
from .inits import glorot, zeros
from .layers import SITE_MLP, SITE_NEIGH, SITE_SELF, Dense, Layer, Rows, _act_code, _rate, relu
from .ops import ACT_IDENTITY, ACT_RELU


def _scope(self_name, name):
    # variable scope naming of aggregators.py:24-29
    return self_name + ('/' + name if name is not None else '') + '_vars'


def _contiguous(rows_list):
    """If the Rows views are adjacent slices of one buffer, return the single Rows covering all of them."""
    first = rows_list[0]
    total = first.n
    for prev, cur in zip(rows_list[:-1], rows_list[1:]):
        if (prev.ids is None) != (cur.ids is None):
            return None
        if cur.ids is not None:
            if cur.src is not prev.src or cur.ids.data_ptr() != prev.ids.data_ptr() + 4 * prev.n:
                return None
        else:
            if cur.src.ld != prev.src.ld or cur.src.d != prev.src.d or \
                    cur.src.buf.data_ptr() != prev.src.buf.data_ptr() + 4 * prev.n * prev.src.ld:
                return None
        total += cur.n
    if len(rows_list) == 1:
        return Rows(first.src, first.ids, first.n, first.requires_grad)
    if first.ids is not None:
        import torch
        ids = torch.as_strided(first.ids, (total,), (1,))
        return Rows(first.src, ids, total, first.requires_grad)
    import torch
    from .ops import Mat
    buf = torch.as_strided(first.src.buf, (total, first.src.buf.shape[1]), first.src.buf.stride())
    return Rows(Mat(buf, first.src.d), None, total, first.requires_grad)


def _run_jobs(e, jobs):
    """Issue gather+mean job descriptors as standalone launches (aggregators without a fused variant)."""
    for j in jobs or ():
        ops.call("gs_gather_mean_fwd", j.X, j.ldx, j.idx, j.n, j.s, j.d, j.self_src, j.ld_self, j.self_idx, j.out, j.ldo,
                 e.stream)


class _SageBase(Layer):
    """Shared plumbing: saved-activation stack, activation backward, masked scatter of input gradients."""

    def _push(self, rec):
        self._saved.append(rec)

    def _dz(self, d_out, out, n, n_cols, pre_masked):
        e = self.engine
        if self.act_code == ACT_RELU and not pre_masked:
            dz = e.ws_mat((self.name, "dz", len(self._saved)), n, n_cols)
            ops.act_bwd(d_out, out, n, n_cols, ACT_RELU, dz, stream=e.stream)
            return dz
        return d_out

    def reset(self):
        del self._saved[:]

    # ---- dropout (tf.nn.dropout on the aggregator inputs, aggregators.py:46-47,104-105; layers.py:107) ----
    def _drop(self, rate, role, k, row0=0):
        return self.engine.dropout(rate, self.site + role + 4 * k, row0)

    def _no_dropout_here(self, what):
        if _rate(self.dropout) > 0:
            raise NotImplementedError("dropout > 0 with %s: use the sequential schedule (model.pipeline = False)" % what)

    def _drop_self(self, self_all, rate, k):
        """dropout(self_vecs), materialised once ([n_total, d]; small) so that the GEMMs read it as a dense operand."""
        e = self.engine
        sd = e.ws_mat((self.name, "self_drop", k), self_all.n, self_all.src.d)
        ops.dropout_rows(self_all.src, self_all.ids, self_all.n, self._drop(rate, SITE_SELF, k), sd, stream=e.stream)
        return Rows(sd, None, self_all.n, self_all.requires_grad)

    def _sink(self, var, d_rows, ids, n, s, scale, rate, role, k, row0, tag):
        """Scatter scale * d_rows[i] to the s sampled ids of row i of a trainable table (identity features); with
        dropout the per-sampled-row mask is applied first (the table rows went through `dropout` before the mean)."""
        e = self.engine
        if rate == 0 or role is None:
            e.scatter_grad(var, d_rows, ids, n, s, scale)
            return
        tmp = e.ws_mat((self.name, "d_sink", k, tag), n * s, var.cols)
        ops.mean_bwd(d_rows, n, s, scale, tmp, stream=e.stream)
        ops.dropout_rows(tmp, None, n * s, self._drop(rate, role, k, row0), tmp, stream=e.stream)
        e.scatter_grad(var, tmp, ids, n * s, 1, 1.0)

    def _bwd_dropped(self, d_rows, n, s, scale, rate, role, k, row0, dst, relu_mask, accumulate, tag):
        """dst (+)= relu'(relu_mask) * dropout_mask * broadcast_s(scale * d_rows): the reverse of `dropout` followed by
        a (segmented) mean, with the mask regenerated from the counter hash."""
        e = self.engine
        tmp = e.ws_mat((self.name, "d_bcast", k, tag), n * s, d_rows.d)
        ops.mean_bwd(d_rows, n, s, scale, tmp, stream=e.stream)
        ops.dropout_rows(tmp, None, n * s, self._drop(rate, role, k, row0), tmp, stream=e.stream)
        ops.mean_bwd(tmp, n * s, 1, 1.0, dst, mask_y=relu_mask, accumulate=accumulate, stream=e.stream)

    def _call(self, inputs):
        self_vecs, neigh_vecs = inputs
        return self.call_hops(self_vecs, [neigh_vecs])

    def backward(self, d_out, pre_masked=False):
        """Single-hop reverse of `_call`: returns raw (d_self [n, d], d_neigh [n*s, d]) when the inputs
        require gradients, else (None, None)."""
        self_all, neighs = self._saved[-1][0], self._saved[-1][1]
        need = self_all.requires_grad or neighs[0].requires_grad
        if not need:
            self.backward_hops(d_out, pre_masked)
            return None, None
        n, s, d = neighs[0].shape3
        d_prev = self.engine.ws_mat((self.name, "d_prev1"), n + n * s, self.input_dim)
        self.backward_hops(d_out, pre_masked, d_prev=d_prev, prev_mask=None, prev_offsets=[0, n, n + n * s])
        return d_prev.rows_slice(0, n), d_prev.rows_slice(n, n + n * s)

    def _scatter_self(self, d_self_all, n_total, d_prev, prev_mask):
        """d_prev[0:n_total] = mask * d_self_all  (self rows of hop h are rows h of the previous layer)."""
        e = self.engine
        act = ACT_RELU if prev_mask is not None else ACT_IDENTITY
        ops.act_bwd(d_self_all, prev_mask.rows_slice(0, n_total) if prev_mask is not None else None, n_total,
                    d_self_all.d, act, d_prev.rows_slice(0, n_total), stream=e.stream)


class MeanAggregator(_SageBase):
    """Aggregates via mean followed by matmul and non-linearity (aggregators.py:6-64)."""

    def __init__(self, input_dim, output_dim, neigh_input_dim=None, dropout=0., bias=False, act=relu,
                 name=None, concat=False, **kwargs):
        super(MeanAggregator, self).__init__(**kwargs)
        self.dropout = dropout
        self.bias = bias
        self.act = act
        self.act_code = _act_code(act)
        self.concat = concat
        if neigh_input_dim is None:
            neigh_input_dim = input_dim
        scope = _scope(self.name, name)
        e = self.engine
        self.vars['neigh_weights'] = e.add_variable(scope + '/neigh_weights', glorot((neigh_input_dim, output_dim)), decay=True)
        self.vars['self_weights'] = e.add_variable(scope + '/self_weights', glorot((input_dim, output_dim)), decay=True)
        if self.bias:
            self.vars['bias'] = e.add_variable(scope + '/bias', zeros(((2 if concat else 1) * output_dim,)), decay=True)
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.neigh_input_dim = neigh_input_dim
        self._saved = []

    def prefetch(self, self_all, neighs, tag=0):
        """The weight-free half of the call: reduce_mean(neigh_vecs, axis=1) (aggregators.py:48) fused with the row
        gather, one launch per hop.  Because it needs no weights it can run ahead of time (next step's data chain)."""
        e = self.engine
        n_total = self_all.n
        d = neighs[0].shape3[2]
        k = len(self._saved)
        rate = _rate(self.dropout)
        means = e.ws_mat((self.name, "mean", k, tag), n_total, d, ld_multiple=32)      # whole 128-byte lines per row
        r = row0 = 0
        for nv in neighs:
            n, s, _ = nv.shape3
            ops.gather_mean_fwd(nv.src, nv.ids, n, s, out=means.rows_slice(r, r + n),
                                drop=self._drop(rate, SITE_NEIGH, k, row0), stream=e.stream)   # dropout(neigh_vecs) (:46)
            r += n
            row0 += n * s
        assert r == n_total
        return means

    def prefetch_jobs(self, self_all, neighs, tag=0):
        """Like prefetch(), but only DESCRIBES the gather+mean launches (one per hop) so that they can be issued
        inside another kernel's launch (horizontal fusion).  Returns (means, jobs)."""
        self._no_dropout_here("the prefetch pipeline")
        e = self.engine
        n_total = self_all.n
        d = neighs[0].shape3[2]
        means = e.ws_mat((self.name, "mean", len(self._saved), tag), n_total, d, ld_multiple=32)
        jobs, r = [], 0
        for nv in neighs:
            n, s, _ = nv.shape3
            jobs.append(ops.gather_job(nv.src, nv.ids, n, s, means.rows_slice(r, r + n)))
            r += n
        return means, jobs

    def _last_layer_z(self, self_all, neighs, rate, means):
        """Can this call be ONE gs_sage_tail_z launch?  A last layer (identity act, concat, no bias, no dropout) over ONE hop
        whose inputs are the dense rows [self (n) | neighbors (n s)] of one buffer -- the layer-1 call of every two-layer
        mean model (models.py:321-328).  Returns that buffer as a Mat, or None."""
        if (means is not None or rate > 0 or len(neighs) != 1 or not self.concat or self.bias or self.act_code != ACT_IDENTITY
                or self_all.ids is not None or neighs[0].ids is not None or not getattr(self, "layer1_z", True)):
            return None
        n, s, d = neighs[0].shape3
        a, b = self_all.src, neighs[0].src
        if (n != self_all.n or s > 11 or a.ld != b.ld or a.d != d or b.d != d or (n + n * s) * a.ld >= (1 << 31)
                or b.buf.data_ptr() != a.buf.data_ptr() + 4 * n * a.ld or not ops.sage_tail_supported(d, self.output_dim, 1)):
            return None
        import torch
        return ops.Mat(torch.as_strided(a.buf, (n + n * s, a.buf.shape[1]), a.buf.stride()), d)

    def call_hops(self, self_all, neighs, means=None, side_jobs=None):
        e = self.engine
        n_total = self_all.n
        k = len(self._saved)
        rate = _rate(self.dropout)
        h0 = self._last_layer_z(self_all, neighs, rate, means)
        if h0 is not None:
            # reduce_mean + both matmuls + concat (aggregators.py:48-58) of the last layer: ONE lean launch instead of a
            # gather-mean launch and a small GEMM (13 -> 8 us at 1044 rows), and one in which gather jobs ride at the full rate
            s = neighs[0].shape3[1]
            means = e.ws_mat((self.name, "mean", k, 0), n_total, h0.d, ld_multiple=32)
            out = e.ws_mat((self.name, "out", k), n_total, 2 * self.output_dim)
            ops.sage_tail_z(h0, n_total, s, self.vars['self_weights'].value, self.vars['neigh_weights'].value, self.output_dim,
                            means, out, jobs=side_jobs, stream=e.stream)
            self._push((self_all, neighs, means, out, rate, self_all, h0))
            return out
        if means is None:
            means = self.prefetch(self_all, neighs)
        self_in = self._drop_self(self_all, rate, k) if rate > 0 else self_all           # dropout(self_vecs) (:47)
        # from_neighs / from_self matmuls + concat|add + bias + act   (:51-64): ONE launch for all hops
        n_out = self.output_dim * (2 if self.concat else 1)
        out = e.ws_mat((self.name, "out", k), n_total, n_out)
        b = self.vars['bias'].value.buf if self.bias else None
        stream_fwd = e.stream_gemm and self.concat and rate == 0 and n_total > 2048 and self.output_dim % 2 == 0
        if side_jobs or stream_fwd:
            # horizontally fused launch: the contraction workgroups + the NEXT step's gather-mean waves share the CUs.
            # Stream form (gs_stream.hip): split-K workgroups without LDS staging, the self rows gathered in the A loads.
            tiled3 = stream_fwd and e.tiled3_fwd and self.output_dim % 4 == 0

            def launch(jobs=list(side_jobs or ())):
                if tiled3:
                    # LDS-tiled, on the bf16 matrix pipe in the three-piece arithmetic (fp32 in and out, cut inside the kernel)
                    ops.sage_dense_fwd_tiled3(self_all.src, self_all.ids, means, n_total, self.vars['self_weights'].value,
                                              self.vars['neigh_weights'].value, self.output_dim, self.act_code, b, out, jobs,
                                              stream=e.stream)
                elif stream_fwd:
                    ops.sage_dense_fwd_stream(self_all.src, self_all.ids, means, n_total, self.vars['self_weights'].value,
                                              self.vars['neigh_weights'].value, self.output_dim, self.act_code, b, out, jobs,
                                              stream=e.stream)
                else:
                    ops.sage_dense_fwd_cogather(self_all.src, self_all.ids, means, None, n_total,
                                                self.vars['self_weights'].value, self.vars['neigh_weights'].value,
                                                self.output_dim, self.concat, self.act_code, b, out, jobs, stream=e.stream)
            launch()
            # bench.py re-issues exactly this launch between HIP events (roofline of the step's dominant kernel)
            d_in = self_all.src.d
            jobs_ = list(side_jobs or ())
            self.last_fused_launch = (launch, {
                "kernel": "%s: [%d x %d|%d] . [%d x %d] x2 (%s) + %d co-scheduled "
                          "gather+mean jobs of the next step" % ("sage_tiled3_fwd_kernel" if tiled3 else
                                                                 "sage_stream_fwd_kernel" if stream_fwd else "sage_dense_cogather_kernel",
                                                                 n_total, d_in, means.d, d_in, self.output_dim,
                                                                 "fp32 as 3 bf16 pieces, 6 bf16 MFMAs per product" if tiled3 else "fp32 MFMA",
                                                                 len(jobs_)),
                "gather_bytes": sum(j.n * j.s * j.d * 4 + j.n * j.s * 4 + j.n * j.d * 4 for j in jobs_),
                "gemm_bytes": n_total * (d_in + means.d) * 4 + (d_in + means.d) * self.output_dim * 4 + n_total * n_out * 4,
                "flops": 2.0 * n_total * (d_in + means.d) * self.output_dim,
                "piece_products": 6 if tiled3 else 1,       # MFMAs issued per fp32 product tile (bf16 pipe) | fp32 pipe
                "gather_share": sum(j.n * j.s for j in jobs_) / float(max(1, sum(nv.shape3[0] * nv.shape3[1] for nv in neighs)))})
        else:
            ops.sage_dense_fwd(self_in.src, self_in.ids, means, None, n_total, self.vars['self_weights'].value,
                               self.vars['neigh_weights'].value, self.output_dim, self.concat, self.act_code, b, out,
                               stream=e.stream)
        self._push((self_all, neighs, means, out, rate, self_in, None))
        return out

    def backward_hops(self, d_out, pre_masked=False, d_prev=None, prev_mask=None, prev_offsets=None, embed_sink=None):
        e = self.engine
        self_all, neighs, means, out, rate, self_in, h0 = self._saved.pop()
        n_total = self_all.n
        k = len(self._saved)
        o = self.output_dim
        n_out = o * (2 if self.concat else 1)
        dz = self._dz(d_out, out, n_total, n_out, pre_masked)
        col_n = o if self.concat else 0
        # (wgrad_ids: the step's private copy of these ids, made by the fused tail launch when a later step's sampler rides in
        #  the weight-gradient launch and refills the id buffer meanwhile -- SupervisedGraphsage._forward)
        wg_ids = getattr(self, "wgrad_ids", None)
        self.wgrad_ids = None
        e.wgrad(self.vars['self_weights'], self_in.src, wg_ids if (wg_ids is not None and self_in.ids is not None) else self_in.ids,
                dz, 0, n_total)
        e.wgrad(self.vars['neigh_weights'], means, None, dz, col_n, n_total)
        if self.bias:
            e.bgrad(self.vars['bias'], dz, n_total, n_out)
        if embed_sink is not None:
            # layer 0 over a table whose leading c columns are trainable (identity features): only those columns of
            # the input gradients are formed ([n, c] = dz . W[:c]^T) and scattered per sampled id, 1/s per neighbor
            var, c = embed_sink
            d_self_e = e.ws_mat((self.name, "d_self_e", k), n_total, c)
            ops.dense_dgrad(dz, 0, o, n_total, self.vars['self_weights'].value.rows_slice(0, c), d_self_e, stream=e.stream)
            d_means_e = e.ws_mat((self.name, "d_means_e", k), n_total, c)
            ops.dense_dgrad(dz, col_n, o, n_total, self.vars['neigh_weights'].value.rows_slice(0, c), d_means_e,
                            stream=e.stream)
            self._sink(var, d_self_e, self_all.ids, n_total, 1, 1.0, rate, SITE_SELF, k, 0, "s")
            r = row0 = 0
            for h, nv in enumerate(neighs):
                n, s, _ = nv.shape3
                self._sink(var, d_means_e.rows_slice(r, r + n), nv.ids, n, s, 1.0 / s, rate, SITE_NEIGH, k, row0, ("n", h))
                r += n
                row0 += n * s
        if d_prev is None:
            return
        d_in = self.input_dim
        if (h0 is not None and rate == 0 and embed_sink is None and prev_mask is not None and prev_mask.ptr == h0.ptr
                and prev_mask.ld == h0.ld and d_prev.rows == h0.rows and d_prev.d == h0.d
                and list(prev_offsets[:3]) == [0, n_total, h0.rows]):
            # the forward went through gs_sage_tail_z: the input gradients are its backward twin, ONE launch
            # (dz . W^T for both terms + relu mask + 1/s broadcast) instead of a small GEMM and the pull
            ops.sage_tail_dh0(h0, n_total, neighs[0].shape3[1], self.vars['self_weights'].value,
                              self.vars['neigh_weights'].value, o, dz, d_prev, jobs=None, stream=e.stream)
            return
        if self.neigh_input_dim == d_in and d_in % 4 == 0 and (not self.concat or o % 4 == 0):
            t2 = e.ws_mat((self.name, "dgrad2", k), n_total, 2 * d_in)       # [d_self | d_means] in one launch
            ops.sage_dense_dgrad(dz, n_total, o, self.concat, self.vars['self_weights'].value,
                                 self.vars['neigh_weights'].value, d_in, t2, stream=e.stream)
            d_self_all, d_means_all = t2.cols_slice(0, d_in), t2.cols_slice(d_in, 2 * d_in)
        else:
            d_self_all = e.ws_mat((self.name, "d_self", k), n_total, d_in)
            ops.dense_dgrad(dz, 0, o, n_total, self.vars['self_weights'].value, d_self_all, stream=e.stream)
            d_means_all = e.ws_mat((self.name, "d_means", k), n_total, self.neigh_input_dim)
            ops.dense_dgrad(dz, col_n, o, n_total, self.vars['neigh_weights'].value, d_means_all, stream=e.stream)
        if rate == 0:
            # ONE launch: d_prev = relu'(prev) * (d_self on the self rows + d_means / s broadcast over each hop's samples)
            segs, r = [], 0
            for h, nv in enumerate(neighs):
                n, s, _ = nv.shape3
                segs.append((d_means_all.rows_slice(r, r + n), prev_offsets[h + 1], n, s, 1.0 / s))
                r += n
            ops.input_grad_pull(d_prev, d_prev.rows, d_prev.d, d_self=d_self_all, n_self=n_total, segments=segs,
                                mask_y=prev_mask, stream=e.stream)
            return
        ops.dropout_rows(d_self_all, None, n_total, self._drop(rate, SITE_SELF, k), d_self_all, stream=e.stream)
        self._scatter_self(d_self_all, n_total, d_prev, prev_mask)
        r = row0 = 0
        for h, nv in enumerate(neighs):
            n, s, _ = nv.shape3
            r0 = prev_offsets[h + 1]
            dst = d_prev.rows_slice(r0, r0 + n * s)
            mask = prev_mask.rows_slice(r0, r0 + n * s) if prev_mask is not None else None
            self._bwd_dropped(d_means_all.rows_slice(r, r + n), n, s, 1.0 / s, rate, SITE_NEIGH, k, row0, dst, mask,
                              (h + 1 < len(neighs)), ("n", h))
            r += n
            row0 += n * s


class GCNAggregator(_SageBase):
    """Same matmul parameters for self and neighbor vectors (aggregators.py:66-116).
    `concat` is stored but ignored, as in the reference (:79)."""

    def __init__(self, input_dim, output_dim, neigh_input_dim=None, dropout=0., bias=False, act=relu, name=None,
                 concat=False, **kwargs):
        super(GCNAggregator, self).__init__(**kwargs)
        self.dropout = dropout
        self.bias = bias
        self.act = act
        self.act_code = _act_code(act)
        self.concat = concat
        if neigh_input_dim is None:
            neigh_input_dim = input_dim
        scope = _scope(self.name, name)
        e = self.engine
        self.vars['weights'] = e.add_variable(scope + '/neigh_weights', glorot((neigh_input_dim, output_dim)), decay=True)
        if self.bias:
            self.vars['bias'] = e.add_variable(scope + '/bias', zeros((output_dim,)), decay=True)
        self.input_dim = input_dim
        self.output_dim = output_dim
        self._saved = []

    def prefetch_jobs(self, self_all, neighs, tag=0):
        self._no_dropout_here("the prefetch pipeline")
        e = self.engine
        d = neighs[0].shape3[2]
        means = e.ws_mat((self.name, "mean", len(self._saved), tag), self_all.n, d, ld_multiple=32)
        jobs, r = [], 0
        for nv in neighs:
            n, s, _ = nv.shape3
            sv = self_all.slice(r, r + n)
            jobs.append(ops.gather_job(nv.src, nv.ids, n, s, means.rows_slice(r, r + n), self_src=sv.src, self_idx=sv.ids))
            r += n
        return means, jobs

    def prefetch(self, self_all, neighs, tag=0):
        """mean over {neighbors} U {self}  (aggregators.py:106-107); weight-free, so it can run ahead of time."""
        e = self.engine
        n_total = self_all.n
        d = neighs[0].shape3[2]
        k = len(self._saved)
        rate = _rate(self.dropout)
        means = e.ws_mat((self.name, "mean", k, tag), n_total, d, ld_multiple=32)      # whole 128-byte lines per row
        self_in = self._drop_self(self_all, rate, k) if rate > 0 else self_all            # dropout(self_vecs) (:105)
        r = row0 = 0
        for nv in neighs:
            n, s, _ = nv.shape3
            sv = self_in.slice(r, r + n)
            ops.gather_mean_fwd(nv.src, nv.ids, n, s, out=means.rows_slice(r, r + n), self_src=sv.src,
                                self_idx=sv.ids, drop=self._drop(rate, SITE_NEIGH, k, row0), stream=e.stream)
            r += n
            row0 += n * s
        return means

    def call_hops(self, self_all, neighs, means=None, side_jobs=None):
        e = self.engine
        n_total = self_all.n
        k = len(self._saved)
        rate = _rate(self.dropout)
        if means is None:
            means = self.prefetch(self_all, neighs)
        out = e.ws_mat((self.name, "out", k), n_total, self.output_dim)
        b = self.vars['bias'].value.buf if self.bias else None
        if e.stream_gemm and e.tiled3_fwd and n_total > 2048 and rate == 0 and self.output_dim % 4 == 0:
            ops.sage_dense_fwd_tiled3(None, None, means, n_total, None, self.vars['weights'].value, self.output_dim, self.act_code,
                                      b, out, side_jobs, stream=e.stream)
        elif e.stream_gemm and n_total > 2048 and rate == 0 and self.output_dim % 2 == 0:
            # stream form: LDS-free contraction waves (+ the next step's gather jobs) in one launch
            ops.sage_dense_fwd_stream(None, None, means, n_total, None, self.vars['weights'].value, self.output_dim, self.act_code,
                                      b, out, side_jobs, stream=e.stream)
        elif side_jobs:
            # horizontally fused launch: these GEMM tiles + the NEXT step's gather-mean waves share the CUs
            ops.sage_dense_fwd_cogather(None, None, means, None, n_total, None, self.vars['weights'].value, self.output_dim,
                                        False, self.act_code, b, out, side_jobs, stream=e.stream)
        else:
            ops.sage_dense_fwd(None, None, means, None, n_total, None, self.vars['weights'].value, self.output_dim, False,
                               self.act_code, b, out, stream=e.stream)
        self._push((self_all, neighs, means, out, rate))
        return out

    def backward_hops(self, d_out, pre_masked=False, d_prev=None, prev_mask=None, prev_offsets=None, embed_sink=None):
        e = self.engine
        self_all, neighs, means, out, rate = self._saved.pop()
        n_total = self_all.n
        k = len(self._saved)
        dz = self._dz(d_out, out, n_total, self.output_dim, pre_masked)
        e.wgrad(self.vars['weights'], means, None, dz, 0, n_total)
        if self.bias:
            e.bgrad(self.vars['bias'], dz, n_total, self.output_dim)
        if embed_sink is not None:
            var, c = embed_sink                    # see MeanAggregator.backward_hops; self counts as one more neighbor
            d_means_e = e.ws_mat((self.name, "d_means_e", k), n_total, c)
            ops.dense_dgrad(dz, 0, self.output_dim, n_total, self.vars['weights'].value.rows_slice(0, c), d_means_e,
                            stream=e.stream)
            r = row0 = 0
            for h, nv in enumerate(neighs):
                n, s, _ = nv.shape3
                dm = d_means_e.rows_slice(r, r + n)
                self._sink(var, dm, self_all.slice(r, r + n).ids, n, 1, 1.0 / (s + 1), rate, SITE_SELF, k, r, ("s", h))
                self._sink(var, dm, nv.ids, n, s, 1.0 / (s + 1), rate, SITE_NEIGH, k, row0, ("n", h))
                r += n
                row0 += n * s
        if d_prev is None:
            return
        d = means.d
        d_means = e.ws_mat((self.name, "d_means", k), n_total, d)
        ops.dense_dgrad(dz, 0, self.output_dim, n_total, self.vars['weights'].value, d_means, stream=e.stream)
        if rate == 0:
            # ONE launch; the self term of hop h is one more "neighbor" of weight 1/(s+1)
            segs, r = [], 0
            for h, nv in enumerate(neighs):
                n, s, _ = nv.shape3
                dm = d_means.rows_slice(r, r + n)
                segs.append((dm, r, n, 1, 1.0 / (s + 1)))
                segs.append((dm, prev_offsets[h + 1], n, s, 1.0 / (s + 1)))
                r += n
            ops.input_grad_pull(d_prev, d_prev.rows, d_prev.d, segments=segs, mask_y=prev_mask, stream=e.stream)
            return
        r = 0
        for h, nv in enumerate(neighs):           # self parts first: d_self = d_means / (s + 1)
            n, s, _ = nv.shape3
            mask = prev_mask.rows_slice(r, r + n) if prev_mask is not None else None
            self._bwd_dropped(d_means.rows_slice(r, r + n), n, 1, 1.0 / (s + 1), rate, SITE_SELF, k, r,
                              d_prev.rows_slice(r, r + n), mask, False, ("s", h))
            r += n
        r = row0 = 0
        for h, nv in enumerate(neighs):
            n, s, _ = nv.shape3
            r0 = prev_offsets[h + 1]
            dst = d_prev.rows_slice(r0, r0 + n * s)
            mask = prev_mask.rows_slice(r0, r0 + n * s) if prev_mask is not None else None
            self._bwd_dropped(d_means.rows_slice(r, r + n), n, s, 1.0 / (s + 1), rate, SITE_NEIGH, k, row0, dst, mask,
                              (h + 1 < len(neighs)), ("n", h))
            r += n
            row0 += n * s


class _PoolingAggregator(_SageBase):
    """relu-MLP over every neighbor row, pooled over the s samples, then the SAGE matmuls
    (aggregators.py:119-195 for max, :197-273 for mean)."""
    POOL = "max"

    def __init__(self, input_dim, output_dim, model_size="small", neigh_input_dim=None, dropout=0., bias=False,
                 act=relu, name=None, concat=False, **kwargs):
        super(_PoolingAggregator, self).__init__(**kwargs)
        self.dropout = dropout
        self.bias = bias
        self.act = act
        self.act_code = _act_code(act)
        self.concat = concat
        if neigh_input_dim is None:
            neigh_input_dim = input_dim
        if model_size == "small":
            hidden_dim = self.hidden_dim = 512
        elif model_size == "big":
            hidden_dim = self.hidden_dim = 1024
        else:
            raise ops._lib.GraphsageAmdError("model_size must be 'small' or 'big'")
        self.mlp_layers = []
        self.mlp_layers.append(Dense(input_dim=neigh_input_dim, output_dim=hidden_dim, act=relu, dropout=dropout,
                                     sparse_inputs=False, logging=self.logging))
        # the MLP weights are NOT part of aggregator.vars (aggregators.py:144-159) -> no weight decay
        for v in self.mlp_layers[0].vars.values():
            v.decay = False
        scope = _scope(self.name, name)
        e = self.engine
        self.vars['neigh_weights'] = e.add_variable(scope + '/neigh_weights', glorot((hidden_dim, output_dim)), decay=True)
        self.vars['self_weights'] = e.add_variable(scope + '/self_weights', glorot((input_dim, output_dim)), decay=True)
        if self.bias:
            self.vars['bias'] = e.add_variable(scope + '/bias', zeros(((2 if concat else 1) * output_dim,)), decay=True)
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.neigh_input_dim = neigh_input_dim
        self._saved = []

    def prefetch(self, self_all, neighs, tag=0):
        return None   # the pooling MLP needs the weights: nothing can run ahead

    def prefetch_jobs(self, self_all, neighs, tag=0):
        return None, []

    def call_hops(self, self_all, neighs, means=None, side_jobs=None):
        e = self.engine
        _run_jobs(e, side_jobs)
        n_total = self_all.n
        k = len(self._saved)
        rate = _rate(self.dropout)
        mlp = self.mlp_layers[0]
        rows_total = sum(nv.shape3[0] * nv.shape3[1] for nv in neighs)
        flat = [Rows(nv.src, nv.ids, nv.shape3[0] * nv.shape3[1], nv.requires_grad) for nv in neighs]
        x_all = _contiguous(flat)
        pieces = [x_all] if x_all is not None else flat
        pooled = e.ws_mat((self.name, "pooled", k), n_total, self.hidden_dim)
        argmax = None
        if self.POOL == "max":
            argmax = e.ws_i32((self.name, "argmax", k), n_total * self.hidden_dim).view(n_total, self.hidden_dim)
        fused_pool = (self.POOL == "max" and rate == 0 and getattr(self, "fuse_pool", True)
                      and all(nv.shape3[1] <= 64 for nv in neighs))
        # layer 0 (rows gathered from the feature table through the model's contiguous id buffer): the MLP of a node does
        # not depend on who sampled it -- run it once per DISTINCT id of the step and let the reduce_max pick rows
        # through an index (37 % fewer GEMM rows at Reddit's degree)
        # gs_unique_ids makes three passes over a flag word per TABLE row (independent of the batch): worth it while the table
        # is within a small multiple of the step's sampled rows (Reddit: 233 k rows for 133 k ids), not for 10^7-node graphs
        dedup_min = getattr(self, "dedup_min_rows", None)
        if dedup_min is None:
            dedup_min = int(os.environ.get("GS_POOL_DEDUP_MIN_ROWS", "2048"))
        dedup = (fused_pool and x_all is not None and x_all.ids is not None and rows_total > dedup_min
                 and x_all.src.rows < (1 << 31)
                 and x_all.src.rows <= int(os.environ.get("GS_POOL_DEDUP_MAX_RATIO", "16")) * rows_total
                 and getattr(self, "dedup_pool", True))
        H = None
        if dedup:
            X, ids, nv_rows = x_all.src, x_all.ids, x_all.src.rows
            rank_ws = e.ws_i32((self.name, "dd_rank", k), 2 * nv_rows)       # [flags | ranks]: zero-initialised, self-cleaning
            sums_ws = e.ws_i32((self.name, "dd_sums", k), 256)
            uniq = e.ws_i32((self.name, "dd_uniq", k), rows_total)
            inv = e.ws_i32((self.name, "dd_inv", k), rows_total)
            cnt = e.ws_i32((self.name, "dd_count", k), 1)
            ops.call("gs_unique_ids", ops.ptr(ids), rows_total, nv_rows, ops.ptr(rank_ws), ops.ptr(sums_ws), ops.ptr(uniq),
                     ops.ptr(inv), ops.ptr(cnt), e.stream)
            Hu = e.ws_mat((self.name, "H_unique", k), rows_total, self.hidden_dim)
            W, bmlp = mlp.vars['weights'].value, mlp.vars['bias'].value.buf
            self.last_pool_kernel = None
            if (e.split_pool and e.pool_f16 and not x_all.requires_grad and e.is_constant_table(X) and e.table16_fits(X)):
                # ... on the fp16 matrix pipe, operands as two fp16 pieces each (fp32 accuracy class, half the matrix-pipe work of
                # the three-piece form below, which is bound by the chip's POWER cap): the constant feature table is cut once.
                # A table with trainable leading columns (identity features, rewritten behind every optimizer launch) is NOT
                # constant -- its cut-once copy would be stale from the second step on -- and takes the three-piece kernel below,
                # which cuts the rows it reads in registers.
                self.last_pool_kernel = "split16"
                X2, rexp = e.table16_of(X)
                ws = e.ws_f32((self.name, "split_ws"), ops.split_tiled_ws_words())
                ops.call("gs_dense_fwd_rows_split16", ops.ptr(X2), ops.ptr(rexp), ops.ptr(uniq), X.d, rows_total, ops.ptr(cnt),
                         ops.ptr(e.split_of(mlp.vars['weights'], form="f16x2")), self.hidden_dim, ACT_RELU, ops.ptr(bmlp),
                         Hu.ptr, Hu.ld, ops.ptr(ws), 4 * ws.numel(), e.stream)
            elif e.split_pool:
                # the 51 GF of the pooling MLP on the bf16 matrix pipe, operands as three bf16 pieces (fp32 accuracy)
                # (+ a workspace: the last, nearly empty round of its one-per-CU workgroups is cut along K, gs_split.hip)
                self.last_pool_kernel = "split_bf16x3"
                ws = e.ws_f32((self.name, "split_ws"), ops.split_tiled_ws_words())
                ops.call("gs_dense_fwd_rows_split_ws", X.ptr, X.ld, ops.ptr(uniq), X.d, rows_total, ops.ptr(cnt),
                         ops.ptr(e.split_of(mlp.vars['weights'])), self.hidden_dim, ACT_RELU, ops.ptr(bmlp), Hu.ptr, Hu.ld,
                         ops.ptr(ws), 4 * ws.numel(), e.stream)
            else:
                self.last_pool_kernel = "fp32_mfma"
                ops.call("gs_dense_fwd_rows_dev", X.ptr, X.ld, ops.ptr(uniq), X.d, rows_total, ops.ptr(cnt), W.ptr, W.ld,
                         self.hidden_dim, ACT_RELU, ops.ptr(bmlp), Hu.ptr, Hu.ld, e.stream)
            r = hr = 0
            for nv in neighs:
                n, s, _ = nv.shape3
                pr, ar = pooled.rows_slice(r, r + n), argmax[r:r + n]
                ops.call("gs_segment_max_gather_fwd", Hu.ptr, Hu.ld, inv.data_ptr() + 4 * hr, n, s, self.hidden_dim, pr.ptr, pr.ld,
                         ar.data_ptr(), argmax.stride(0), e.stream)
                r += n
                hr += n * s
            self.last_unique = (cnt, rows_total)
        elif fused_pool:
            # Dense (:176-179) + reduce_max (:181) in ONE launch per hop: the GEMM tiles hold whole neighbor groups and
            # reduce them in the epilogue, so the [n*s, hidden] activations never exist in HBM
            r = 0
            for nv, x in zip(neighs, flat):
                n, s, _ = nv.shape3
                ops.dense_pool_max_fwd(x.src, x.ids, n, s, mlp.vars['weights'].value, mlp.vars['bias'].value.buf,
                                       pooled.rows_slice(r, r + n), argmax[r:r + n], stream=e.stream)
                r += n
        else:
            # h_reshaped = Dense(reshape(neigh, [n*s, d]))   (aggregators.py:176-179): one GEMM over every neighbor row
            H = e.ws_mat((self.name, "H", k), rows_total, self.hidden_dim)
            if rate > 0:
                # the Dense's x = tf.nn.dropout(x, 1 - dropout) (layers.py:107) over every gathered neighbor row: the
                # dropped rows are materialised ([n*s, d]) and feed the MLP GEMM and its weight gradient as a dense operand
                dropped, r = [], 0
                for i, x in enumerate(pieces):
                    xd = e.ws_mat((self.name, "x_drop", k, i), x.n, x.src.d)
                    ops.dropout_rows(x.src, x.ids, x.n, self._drop(rate, SITE_MLP, k, r), xd, stream=e.stream)
                    dropped.append(Rows(xd, None, x.n, x.requires_grad))
                    r += x.n
                pieces = dropped
            r = 0
            for x in pieces:
                ops.sage_dense_fwd(None, None, x.src, x.ids, x.n, None, mlp.vars['weights'].value, self.hidden_dim, False,
                                   ACT_RELU, mlp.vars['bias'].value.buf, H.rows_slice(r, r + x.n), stream=e.stream)
                r += x.n
            r = hr = 0
            for nv in neighs:
                n, s, _ = nv.shape3
                if self.POOL == "max":
                    ops.segment_max_fwd(H.rows_slice(hr, hr + n * s), n, s, pooled.rows_slice(r, r + n), argmax[r:r + n],
                                        stream=e.stream)                                             # reduce_max (:181)
                else:
                    ops.gather_mean_fwd(H.rows_slice(hr, hr + n * s), None, n, s, out=pooled.rows_slice(r, r + n),
                                        stream=e.stream)                                             # reduce_mean (:259)
                r += n
                hr += n * s
        n_out = self.output_dim * (2 if self.concat else 1)
        out = e.ws_mat((self.name, "out", k), n_total, n_out)
        b = self.vars['bias'].value.buf if self.bias else None
        if e.stream_gemm and self.concat and n_total > 2048 and self.output_dim % 2 == 0:
            # the stream form of the two contractions (split-K workgroups, no LDS staging, the self rows gathered in the A loads),
            # with each term's own reduction length: 23 instead of 33 us for the Reddit step's layer 0
            (ops.sage_dense_fwd_tiled3 if (e.tiled3_fwd and self.output_dim % 4 == 0) else ops.sage_dense_fwd_stream2)(
                self_all.src, self_all.ids, pooled, n_total, self.vars['self_weights'].value, self.vars['neigh_weights'].value,
                self.output_dim, self.act_code, b, out, stream=e.stream)
        else:
            ops.sage_dense_fwd(self_all.src, self_all.ids, pooled, None, n_total, self.vars['self_weights'].value,
                               self.vars['neigh_weights'].value, self.output_dim, self.concat, self.act_code, b, out,
                               stream=e.stream)
        self._push((self_all, neighs, pieces, (H, rows_total), pooled, argmax, out, rate))
        return out

    def backward_hops(self, d_out, pre_masked=False, d_prev=None, prev_mask=None, prev_offsets=None, embed_sink=None):
        e = self.engine
        self_all, neighs, pieces, (H, rows_total), pooled, argmax, out, rate = self._saved.pop()
        n_total = self_all.n
        k = len(self._saved)

        o = self.output_dim
        n_out = o * (2 if self.concat else 1)
        mlp = self.mlp_layers[0]
        dz = self._dz(d_out, out, n_total, n_out, pre_masked)
        col_n = o if self.concat else 0
        e.wgrad(self.vars['self_weights'], self_all.src, self_all.ids, dz, 0, n_total)
        e.wgrad(self.vars['neigh_weights'], pooled, None, dz, col_n, n_total)
        if self.bias:
            e.bgrad(self.vars['bias'], dz, n_total, n_out)
        d_pooled = e.ws_mat((self.name, "d_pooled", k), n_total, self.hidden_dim)
        ops.dense_dgrad(dz, col_n, o, n_total, self.vars['neigh_weights'].value, d_pooled, stream=e.stream)
        dpm = None
        if self.POOL == "max":
            # reduce_max grad then the Dense's relu grad: only the arg-max row of each (group, column) receives
            # gradient, and only where the pooled activation is > 0.
            dpm = e.ws_mat((self.name, "d_pooled_masked", k), n_total, self.hidden_dim)
            ops.act_bwd(d_pooled, pooled, n_total, self.hidden_dim, ACT_RELU, dpm, stream=e.stream)
            e.bgrad(mlp.vars['bias'], dpm, n_total, self.hidden_dim)   # column sums of dH == column sums of dpm
        threads = min(512, (self.hidden_dim + 63) // 64 * 64)
        sparse = (self.POOL == "max" and d_prev is None and embed_sink is None and rate == 0
                  and all(nv.ids is not None for nv in neighs)
                  and getattr(self, "sparse_wgrad", True)
                  and 16 * max(nv.shape3[1] for nv in neighs) <= 4 * threads)
        if sparse:
            # layer 0: the gathered feature rows need no gradient, so dH = [n*s, hidden] is never materialised; the
            # MLP weight gradient is accumulated straight from the arg-max rows (gs_maxpool_sparse_wgrad)
            r = 0
            for nv in neighs:
                n, s, _ = nv.shape3
                e.sparse_pool_wgrad(mlp.vars['weights'], nv.src, nv.ids, n, s, argmax[r:r + n], dpm.rows_slice(r, r + n))
                r += n
            return
        dH = e.ws_mat((self.name, "dH", k), rows_total, self.hidden_dim)
        r = hr = 0
        for nv in neighs:
            n, s, _ = nv.shape3
            if self.POOL == "max":
                ops.segment_max_bwd(dpm.rows_slice(r, r + n), pooled.rows_slice(r, r + n), argmax[r:r + n], n, s,
                                    dH.rows_slice(hr, hr + n * s), stream=e.stream)
            else:
                ops.mean_bwd(d_pooled.rows_slice(r, r + n), n, s, 1.0 / s, dH.rows_slice(hr, hr + n * s),
                             mask_y=H.rows_slice(hr, hr + n * s), stream=e.stream)
            r += n
            hr += n * s
        if self.POOL != "max":
            e.bgrad(mlp.vars['bias'], dH, rows_total, self.hidden_dim)
        r = 0
        for x in pieces:
            e.wgrad(mlp.vars['weights'], x.src, x.ids, dH.rows_slice(r, r + x.n), 0, x.n)
            r += x.n
        if embed_sink is not None:
            var, c = embed_sink                    # see MeanAggregator.backward_hops; every neighbor row has its own dH
            d_self_e = e.ws_mat((self.name, "d_self_e", k), n_total, c)
            ops.dense_dgrad(dz, 0, o, n_total, self.vars['self_weights'].value.rows_slice(0, c), d_self_e, stream=e.stream)
            e.scatter_grad(var, d_self_e, self_all.ids, n_total, 1, 1.0)      # the pooling aggregators do not drop self
            d_neigh_e = e.ws_mat((self.name, "d_neigh_e", k), rows_total, c)
            ops.dense_dgrad(dH, 0, self.hidden_dim, rows_total, mlp.vars['weights'].value.rows_slice(0, c), d_neigh_e,
                            stream=e.stream)
            if rate > 0:
                ops.dropout_rows(d_neigh_e, None, rows_total, self._drop(rate, SITE_MLP, k), d_neigh_e, stream=e.stream)
            hr = 0
            for nv in neighs:
                n, s, _ = nv.shape3
                e.scatter_grad(var, d_neigh_e.rows_slice(hr, hr + n * s), nv.ids, n * s, 1, 1.0)
                hr += n * s
        if d_prev is None:
            return
        d_self_all = e.ws_mat((self.name, "d_self", k), n_total, self.input_dim)
        ops.dense_dgrad(dz, 0, o, n_total, self.vars['self_weights'].value, d_self_all, stream=e.stream)
        d_neigh = e.ws_mat((self.name, "d_neigh", k), rows_total, self.neigh_input_dim)
        ops.dense_dgrad(dH, 0, self.hidden_dim, rows_total, mlp.vars['weights'].value, d_neigh, stream=e.stream)
        if rate > 0:
            ops.dropout_rows(d_neigh, None, rows_total, self._drop(rate, SITE_MLP, k), d_neigh, stream=e.stream)
        segs, hr = [], 0
        for h, nv in enumerate(neighs):         # every neighbor row has its own gradient row (s = 1)
            n, s, _ = nv.shape3
            segs.append((d_neigh.rows_slice(hr, hr + n * s), prev_offsets[h + 1], n * s, 1, 1.0))
            hr += n * s
        ops.input_grad_pull(d_prev, d_prev.rows, d_prev.d, d_self=d_self_all, n_self=n_total, segments=segs,
                            mask_y=prev_mask, stream=e.stream)


class MaxPoolingAggregator(_PoolingAggregator):
    """Aggregates via max-pooling over MLP functions (aggregators.py:119-195)."""
    POOL = "max"


class MeanPoolingAggregator(_PoolingAggregator):
    """Aggregates via mean-pooling over MLP functions (aggregators.py:197-273)."""
    POOL = "mean"
