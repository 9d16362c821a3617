"""Layer (the operator/plugin base class) and Dense -- same call surface as graphsage/layers.py.

`Layer.__call__(inputs)` wraps `_call(inputs)` (layers.py:59-66); kwargs are restricted to
{'name', 'logging', 'model_size'} (layers.py:43-45); `.vars` is the dict of trainable variables
that the model iterates for weight decay (supervised_models.py:104-108).

Differences by design: the TF graph is replaced by eager kernel launches on the engine's HIP
stream, and every layer also implements `backward()` (TF derived it via compute_gradients).
Inputs/outputs are `Rows` views (a device matrix + optional row-index vector = a lazy
tf.nn.embedding_lookup, models.py:299) instead of tf.Tensors.
"""
from . import ops
from .engine import get_engine
from .inits import glorot, zeros
from .ops import ACT_IDENTITY, ACT_RELU

# global unique layer ID dictionary for layer name assignment (layers.py:16-26)
_LAYER_UIDS = {}


def get_layer_uid(layer_name=''):
    if layer_name not in _LAYER_UIDS:
        _LAYER_UIDS[layer_name] = 1
        return 1
    _LAYER_UIDS[layer_name] += 1
    return _LAYER_UIDS[layer_name]


def relu(x):
    """Marker for the reference's `act=tf.nn.relu` default (aggregators.py:12)."""
    return x


def identity(x):
    """Marker for `act=lambda x: x` (models.py:308, supervised_models.py:90)."""
    return x


def _act_code(act):
    if act is relu or act == "relu":
        return ACT_RELU
    if act is None or act is identity or act == "identity":
        return ACT_IDENTITY
    # the reference passes `lambda x: x` for the last layer (models.py:308): probe it on a sentinel
    try:
        probe = object()
        if act(probe) is probe:
            return ACT_IDENTITY
    except Exception:
        pass
    raise ops._lib.GraphsageAmdError("only relu and identity activations have gfx950 epilogues")


class Rows(object):
    """rows `ids` of matrix `src` ([n, d]); ids None -> the matrix itself.  `requires_grad` is False
    for the non-trainable feature table (models.py:238)."""

    __slots__ = ("src", "ids", "n", "requires_grad", "shape3")

    def __init__(self, src, ids=None, n=None, requires_grad=False, shape3=None):
        self.src = src
        self.ids = ids
        self.n = n if n is not None else (ids.numel() if ids is not None else src.rows)
        self.requires_grad = requires_grad
        self.shape3 = shape3  # (n, s, d) after reshape (models.py:327)

    @property
    def d(self):
        return self.src.d

    def slice(self, r0, r1):
        """Rows [r0, r1) of this view."""
        if self.ids is not None:
            return Rows(self.src, self.ids[r0:r1], r1 - r0, self.requires_grad)
        return Rows(self.src.rows_slice(r0, r1), None, r1 - r0, self.requires_grad)

    def reshape(self, dims):
        n, s, d = dims
        assert n * s == self.n and d == self.src.d, (dims, self.n, self.src.d)
        return Rows(self.src, self.ids, self.n, self.requires_grad, (n, s, d))


class Layer(object):
    """Base layer class; see module docstring (graphsage/layers.py:28-70)."""

    def __init__(self, **kwargs):
        allowed_kwargs = {'name', 'logging', 'model_size'}
        for kwarg in kwargs.keys():
            assert kwarg in allowed_kwargs, 'Invalid keyword argument: ' + kwarg
        name = kwargs.get('name')
        if not name:
            layer = self.__class__.__name__.lower()
            name = layer + '_' + str(get_layer_uid(layer))
        self.name = name
        self.vars = {}
        self.logging = kwargs.get('logging', False)
        self.sparse_inputs = False
        self.engine = get_engine()
        self.site = self.engine.new_site()

    def _call(self, inputs):
        return inputs

    def __call__(self, inputs):
        return self._call(inputs)

    def _log_vars(self):
        pass  # TF histogram summaries (layers.py:68-70) have no equivalent here


def _rate(dropout):
    """Current dropout rate of a layer: a number, or the model's `dropout` placeholder (fed per step,
    supervised_train.py:117; 0 for validation feeds, minibatch.py:269 / supervised_train.py:58-61)."""
    p = dropout.value if hasattr(dropout, "value") else dropout
    p = float(p) if p is not None else 0.0
    if not 0.0 <= p < 1.0:
        raise ops._lib.GraphsageAmdError("dropout rate must be in [0, 1), got %r" % p)
    return p


# dropout call-site roles inside one layer object (site id = layer.site + role + 4 * call index)
SITE_SELF, SITE_NEIGH, SITE_MLP, SITE_DENSE = 0, 1, 2, 3


class Dense(Layer):
    """act(x @ W + b)  -- graphsage/layers.py:73-116.  W Xavier-uniform, b zeros (:94-99)."""

    def __init__(self, input_dim, output_dim, dropout=0., act=relu, placeholders=None, bias=True,
                 featureless=False, sparse_inputs=False, **kwargs):
        super(Dense, self).__init__(**kwargs)
        self.dropout = dropout
        self.act = act
        self.act_code = _act_code(act)
        self.featureless = featureless
        self.bias = bias
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.sparse_inputs = sparse_inputs
        e = self.engine
        self.vars['weights'] = e.add_variable(self.name + '_vars/weights', glorot((input_dim, output_dim)), decay=True)
        if self.bias:
            self.vars['bias'] = e.add_variable(self.name + '_vars/bias', zeros((output_dim,)), decay=True)
        self._saved = []

    def _call(self, inputs):
        """inputs: Rows [n, input_dim] (possibly a lazy row gather).  Returns a Mat [n, output_dim]."""
        x = inputs if isinstance(inputs, Rows) else Rows(inputs)
        e = self.engine
        k = len(self._saved)
        rate = _rate(self.dropout)
        if rate > 0:                                                     # x = tf.nn.dropout(x, 1 - dropout)  (:107)
            xd = e.ws_mat((self.name, "x_drop", k), x.n, self.input_dim)
            ops.dropout_rows(x.src, x.ids, x.n, e.dropout(rate, self.site + SITE_DENSE + 4 * k), xd, stream=e.stream)
            x = Rows(xd, None, x.n, x.requires_grad)
        out = e.ws_mat((self.name, "out", k), x.n, self.output_dim)
        b = self.vars['bias'].value.buf if self.bias else None
        ops.sage_dense_fwd(None, None, x.src, x.ids, x.n, None, self.vars['weights'].value, self.output_dim, False,
                           self.act_code, b, out, stream=e.stream)
        self._saved.append((x, out, rate))
        return out

    def backward(self, d_out, need_input_grad=True, pre_masked=False):
        """d_out: Mat [n, output_dim] = dLoss/d(output).  Returns dLoss/d(input) (Mat) or None."""
        e = self.engine
        x, out, rate = self._saved.pop()
        dz = d_out
        if self.act_code == ACT_RELU and not pre_masked:
            dz = e.ws_mat((self.name, "dz", len(self._saved)), x.n, self.output_dim)
            ops.act_bwd(d_out, out, x.n, self.output_dim, ACT_RELU, dz, stream=e.stream)
        e.wgrad(self.vars['weights'], x.src, x.ids, dz, 0, x.n)
        if self.bias:
            e.bgrad(self.vars['bias'], dz, x.n, self.output_dim)
        if not need_input_grad:
            return None
        dx = e.ws_mat((self.name, "dx", len(self._saved)), x.n, self.input_dim)
        ops.dense_dgrad(dz, 0, self.output_dim, x.n, self.vars['weights'].value, dx, stream=e.stream)
        if rate > 0:
            ops.dropout_rows(dx, None, x.n, e.dropout(rate, self.site + SITE_DENSE + 4 * len(self._saved)), dx,
                             stream=e.stream)
        return dx
