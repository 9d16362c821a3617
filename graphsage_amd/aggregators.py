"""Mean / GCN / MaxPooling / MeanPooling aggregators with the constructor and call signatures of
graphsage/aggregators.py, executing on the gfx950 kernels.

    agg = MeanAggregator(input_dim, output_dim, act=..., dropout=..., name=..., concat=..., model_size=...)
    out = agg((self_vecs, neigh_vecs))        # models.py:326-327

`self_vecs` is a `Rows` [n, d]; `neigh_vecs` is a `Rows` reshaped to [n, s, d] (both may be lazy
row gathers of the feature table, so the [n*s, d] tensor of models.py:299 is never materialised).
`.vars` holds exactly the variables the reference's weight-decay loop sees (supervised_models.py:104-106).
Every aggregator also implements `backward(d_out, ...)`.
"""
from . import ops
from .layers import Dense, Layer, Rows, _act_code, _check_dropout, relu
from .inits import glorot, zeros
from .ops import ACT_IDENTITY, ACT_RELU


def _scope(self_name, name):
    # variable scope naming of aggregators.py:24-29
    return self_name + ('/' + name if name is not None else '') + '_vars'


class _SageBase(Layer):
    """Shared plumbing: saved-activation stack, activation backward."""

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

    def _call(self, inputs):
        self_vecs, neigh_vecs = inputs
        _check_dropout(self.dropout)
        e = self.engine
        n, s, d = neigh_vecs.shape3
        k = len(self._saved)
        # reduce_mean(neigh_vecs, axis=1)   (aggregators.py:48) fused with the row gather
        means = e.ws_mat((self.name, "mean", k), n, d)
        ops.gather_mean_fwd(neigh_vecs.src, neigh_vecs.ids, n, s, out=means, stream=e.stream)
        # from_neighs / from_self matmuls + concat|add + bias + act   (:51-64) in one launch
        n_out = self.output_dim * (2 if self.concat else 1)
        out = e.ws_mat((self.name, "out", k), n, n_out)
        b = self.vars['bias'].value.buf if self.bias else None
        ops.sage_dense_fwd(self_vecs.src, self_vecs.ids, means, None, n, self.vars['self_weights'].value,
                           self.vars['neigh_weights'].value, self.output_dim, self.concat, self.act_code, b, out,
                           stream=e.stream)
        self._push((self_vecs, neigh_vecs, means, out))
        return out

    def backward(self, d_out, pre_masked=False, neigh_mask=None):
        """Returns (d_self, d_neigh): Mats or None when the corresponding input needs no gradient.
        `neigh_mask` (the relu output that produced the neighbor rows) fuses that layer's relu
        gradient into d_neigh."""
        e = self.engine
        self_vecs, neigh_vecs, means, out = self._saved.pop()
        n, s, d = neigh_vecs.shape3
        k = len(self._saved)
        o = self.output_dim
        n_out = o * (2 if self.concat else 1)
        dz = self._dz(d_out, out, n, n_out, pre_masked)
        col_n = o if self.concat else 0
        e.wgrad(self.vars['self_weights'], self_vecs.src, self_vecs.ids, dz, 0, n)
        e.wgrad(self.vars['neigh_weights'], means, None, dz, col_n, n)
        if self.bias:
            e.bgrad(self.vars['bias'], dz, n, n_out)
        d_self = d_neigh = None
        if self_vecs.requires_grad:
            d_self = e.ws_mat((self.name, "d_self", k), n, self.input_dim)
            ops.dense_dgrad(dz, 0, o, n, self.vars['self_weights'].value, d_self, stream=e.stream)
        if neigh_vecs.requires_grad:
            d_means = e.ws_mat((self.name, "d_means", k), n, d)
            ops.dense_dgrad(dz, col_n, o, n, self.vars['neigh_weights'].value, d_means, stream=e.stream)
            d_neigh = e.ws_mat((self.name, "d_neigh", k), n * s, d)
            ops.mean_bwd(d_means, n, s, 1.0 / s, d_neigh, mask_y=neigh_mask, stream=e.stream)
        return d_self, d_neigh


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

    def _call(self, inputs):
        self_vecs, neigh_vecs = inputs
        _check_dropout(self.dropout)
        e = self.engine
        n, s, d = neigh_vecs.shape3
        k = len(self._saved)
        # mean over {neighbors} U {self}  (aggregators.py:106-107)
        means = e.ws_mat((self.name, "mean", k), n, d)
        ops.gather_mean_fwd(neigh_vecs.src, neigh_vecs.ids, n, s, out=means, self_src=self_vecs.src,
                            self_idx=self_vecs.ids, stream=e.stream)
        out = e.ws_mat((self.name, "out", k), n, self.output_dim)
        b = self.vars['bias'].value.buf if self.bias else None
        ops.sage_dense_fwd(None, None, means, None, n, None, self.vars['weights'].value, self.output_dim, False,
                           self.act_code, b, out, stream=e.stream)
        self._push((self_vecs, neigh_vecs, means, out))
        return out

    def backward(self, d_out, pre_masked=False, neigh_mask=None):
        e = self.engine
        self_vecs, neigh_vecs, means, out = self._saved.pop()
        n, s, d = neigh_vecs.shape3
        k = len(self._saved)
        dz = self._dz(d_out, out, n, self.output_dim, pre_masked)
        e.wgrad(self.vars['weights'], means, None, dz, 0, n)
        if self.bias:
            e.bgrad(self.vars['bias'], dz, n, self.output_dim)
        d_self = d_neigh = None
        if self_vecs.requires_grad or neigh_vecs.requires_grad:
            d_means = e.ws_mat((self.name, "d_means", k), n, d)
            ops.dense_dgrad(dz, 0, self.output_dim, n, self.vars['weights'].value, d_means, stream=e.stream)
            if self_vecs.requires_grad:
                d_self = e.ws_mat((self.name, "d_self", k), n, d)
                ops.mean_bwd(d_means, n, 1, 1.0 / (s + 1), d_self, stream=e.stream)
            if neigh_vecs.requires_grad:
                d_neigh = e.ws_mat((self.name, "d_neigh", k), n * s, d)
                ops.mean_bwd(d_means, n, s, 1.0 / (s + 1), d_neigh, mask_y=neigh_mask, stream=e.stream)
        return d_self, d_neigh


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

    def reset(self):
        del self._saved[:]
        for l in self.mlp_layers:
            del l._saved[:]

    def _call(self, inputs):
        self_vecs, neigh_vecs = inputs
        e = self.engine
        n, s, d = neigh_vecs.shape3
        k = len(self._saved)
        # h_reshaped = Dense(reshape(neigh, [n*s, d]))   (aggregators.py:176-179)
        h = Rows(neigh_vecs.src, neigh_vecs.ids, n * s, neigh_vecs.requires_grad)
        for l in self.mlp_layers:
            h = l(h)
        H = h  # Mat [n*s, hidden]
        pooled = e.ws_mat((self.name, "pooled", k), n, self.hidden_dim)
        argmax = None
        if self.POOL == "max":
            argmax = e.ws_i32((self.name, "argmax", k), n * self.hidden_dim).view(n, self.hidden_dim)
            ops.segment_max_fwd(H, n, s, pooled, argmax, stream=e.stream)                 # reduce_max (:181)
        else:
            ops.gather_mean_fwd(H, None, n, s, out=pooled, stream=e.stream)              # reduce_mean (:259)
        n_out = self.output_dim * (2 if self.concat else 1)
        out = e.ws_mat((self.name, "out", k), n, n_out)
        b = self.vars['bias'].value.buf if self.bias else None
        ops.sage_dense_fwd(self_vecs.src, self_vecs.ids, pooled, None, n, self.vars['self_weights'].value,
                           self.vars['neigh_weights'].value, self.output_dim, self.concat, self.act_code, b, out,
                           stream=e.stream)
        self._push((self_vecs, neigh_vecs, H, pooled, argmax, out))
        return out

    def backward(self, d_out, pre_masked=False, neigh_mask=None):
        e = self.engine
        self_vecs, neigh_vecs, H, pooled, argmax, out = self._saved.pop()
        n, s, d = neigh_vecs.shape3
        k = len(self._saved)
        o = self.output_dim
        n_out = o * (2 if self.concat else 1)
        dz = self._dz(d_out, out, n, n_out, pre_masked)
        col_n = o if self.concat else 0
        e.wgrad(self.vars['self_weights'], self_vecs.src, self_vecs.ids, dz, 0, n)
        e.wgrad(self.vars['neigh_weights'], pooled, None, dz, col_n, n)
        if self.bias:
            e.bgrad(self.vars['bias'], dz, n, n_out)
        d_self = None
        if self_vecs.requires_grad:
            d_self = e.ws_mat((self.name, "d_self", k), n, self.input_dim)
            ops.dense_dgrad(dz, 0, o, n, self.vars['self_weights'].value, d_self, stream=e.stream)
        d_pooled = e.ws_mat((self.name, "d_pooled", k), n, self.hidden_dim)
        ops.dense_dgrad(dz, col_n, o, n, self.vars['neigh_weights'].value, d_pooled, stream=e.stream)
        dH = e.ws_mat((self.name, "dH", k), n * s, self.hidden_dim)
        mlp = self.mlp_layers[0]
        x, _ = mlp._saved.pop()
        if self.POOL == "max":
            # reduce_max grad then the Dense's relu grad: only the arg-max row of each (group, column)
            # receives gradient, and only where the pooled activation is > 0.
            dpm = e.ws_mat((self.name, "d_pooled_masked", k), n, self.hidden_dim)
            ops.act_bwd(d_pooled, pooled, n, self.hidden_dim, ACT_RELU, dpm, stream=e.stream)
            ops.segment_max_bwd(dpm, pooled, argmax, n, s, dH, stream=e.stream)
            e.bgrad(mlp.vars['bias'], dpm, n, self.hidden_dim)  # column sums of dH == column sums of dpm
        else:
            ops.mean_bwd(d_pooled, n, s, 1.0 / s, dH, mask_y=H, stream=e.stream)
            e.bgrad(mlp.vars['bias'], dH, n * s, self.hidden_dim)
        e.wgrad(mlp.vars['weights'], x.src, x.ids, dH, 0, n * s)
        d_neigh = None
        if neigh_vecs.requires_grad:
            d_neigh = e.ws_mat((self.name, "d_neigh", k), n * s, d)
            ops.dense_dgrad(dH, 0, self.hidden_dim, n * s, mlp.vars['weights'].value, d_neigh, stream=e.stream)
            if neigh_mask is not None:
                ops.act_bwd(d_neigh, neigh_mask, n * s, d, ACT_RELU, d_neigh, stream=e.stream)
        return d_self, d_neigh


class MaxPoolingAggregator(_PoolingAggregator):
    """Aggregates via max-pooling over MLP functions (aggregators.py:119-195)."""
    POOL = "max"


class MeanPoolingAggregator(_PoolingAggregator):
    """Aggregates via mean-pooling over MLP functions (aggregators.py:197-273)."""
    POOL = "mean"
