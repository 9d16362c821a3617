"""BipartiteEdgePredLayer -- constructor surface of graphsage/prediction.py:12-66; the xent skip-gram loss
(:102-110), the affinities (:68-92) and the MRR ranks of models.py:393-405 run as ONE fused gfx950 kernel
(gs_linkpred_fwd_bwd) that also produces the gradients w.r.t. the three groups of normalised embeddings.
Only loss_fn='xent' with bilinear_weights=False is on the hot path (the only configuration the reference
drivers use, models.py:363-366); other settings fail loudly."""
from . import ops
from .layers import Layer


class BipartiteEdgePredLayer(Layer):
    def __init__(self, input_dim1, input_dim2, placeholders, dropout=False, act="sigmoid", loss_fn='xent',
                 neg_sample_weights=1.0, bias=False, bilinear_weights=False, **kwargs):
        super(BipartiteEdgePredLayer, self).__init__(**kwargs)
        if loss_fn != 'xent' or bilinear_weights or bias:
            raise NotImplementedError("only loss_fn='xent', bilinear_weights=False, bias=False is implemented "
                                      "(what models.py:363-366 instantiates)")
        self.input_dim1 = input_dim1
        self.input_dim2 = input_dim2
        self.act = act
        self.bias = bias
        self.eps = 1e-7
        self.margin = 0.1
        self.neg_sample_weights = neg_sample_weights
        self.bilinear_weights = bilinear_weights
        self.dropout = 0.
        self.output_dim = 1

    def loss_and_grads(self, outputs_all, batch_size, n_neg, scale, loss_rows, rr_rows, aff_all, d_outputs_all):
        """outputs_all: Mat [2B + n_neg, d] = [outputs1 | outputs2 | neg_outputs] (l2-normalised).
        Fills loss_rows (per-pair xent, :102-110), rr_rows (1/(rank+1), models.py:399-404), aff_all ([neg_aff | aff],
        models.py:395-400) and d_outputs_all = scale * dLoss/d(outputs_all)."""
        import ctypes
        e = self.engine
        d = outputs_all.d
        B = batch_size
        n_slabs = (B + 3) // 4
        slabs = e.ws_f32((self.name, "neg_slabs", B, n_neg, d), n_slabs * n_neg * d)
        out_n = ctypes.c_int32()
        ops.call("gs_linkpred_fwd_bwd", outputs_all.ptr, outputs_all.ld, B, d, n_neg, float(self.neg_sample_weights),
                 float(scale), ops.ptr(loss_rows), ops.ptr(rr_rows), aff_all.ptr if aff_all is not None else None,
                 aff_all.ld if aff_all is not None else 0, d_outputs_all.ptr, d_outputs_all.ld, ops.ptr(slabs),
                 ctypes.byref(out_n), e.stream)
        assert out_n.value == n_slabs
        dneg = d_outputs_all.rows_slice(2 * B, 2 * B + n_neg)
        ops.call("gs_reduce_slabs", ops.ptr(slabs), n_slabs, n_neg * d, n_neg, d, d, 0.0, None, 0, dneg.ptr, dneg.ld, 0,
                 e.stream)

    def loss_and_grads_fused(self, z_all, outputs_all, batch_size, n_neg, scale, loss_rows, rr_rows, aff_all, d_z_all,
                             epilogue=None):
        """z_all: Mat [2B + n_neg, d] RAW aggregator outputs.  One launch (+ a small one for the negatives' rows):
        outputs_all = l2_normalize(z_all) (models.py:368-370), loss_rows / rr_rows / aff_all as loss_and_grads, and
        d_z_all = scale * dLoss/d(z_all) (the gradient carried back through the normalisation)."""
        e = self.engine
        d = z_all.d
        B = batch_size
        n_slabs = (B + 3) // 4
        slabs = e.ws_f32((self.name, "neg_slabs", B, n_neg, d), n_slabs * n_neg * d)
        args = (z_all.ptr, z_all.ld, B, d, n_neg, float(self.neg_sample_weights), float(scale),
                outputs_all.ptr, outputs_all.ld, ops.ptr(loss_rows), ops.ptr(rr_rows),
                aff_all.ptr if aff_all is not None else None, aff_all.ld if aff_all is not None else 0,
                d_z_all.ptr, d_z_all.ld, ops.ptr(slabs))
        if epilogue is None:
            ops.call("gs_linkpred_norm_fwd_bwd", *args, e.stream)
        else:
            # `epilogue`: (loss_out, accumulate, mrr_out, [(counter, delta)] * 3) -- the step's loss / mrr means and device
            # counters ride in the second launch
            loss_out, accumulate, mrr_out, counters = epilogue
            cargs = []
            for c, dlt in counters:
                cargs += [ops.ptr(c) if (c is not None and dlt) else None, int(dlt) if c is not None else 0]
            ops.call("gs_linkpred_norm_fwd_bwd_step", *args, ops.ptr(loss_out), 1 if accumulate else 0, ops.ptr(mrr_out), *cargs,
                     e.stream)
