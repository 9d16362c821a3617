"""A minimal, torch-backed stand-in for the TensorFlow 1.x graph-mode API.  TEST INFRASTRUCTURE ONLY.

Why it exists: the reference (williamleif/GraphSAGE) is Python on `tensorflow==1.8.0` (`/root/reference/requirements.txt:23`),
which is neither installed nor installable here.  With THIS directory's parent (`tests/tf1_shim`) on `sys.path`,
`import tensorflow as tf` resolves to this module and the reference's own modules
(`graphsage/{inits,layers,neigh_samplers,aggregators,prediction,models,supervised_models,metrics,minibatch}.py`) can be
imported UNMODIFIED from `/root/reference` and executed: `tests/golden/make_ref_fixtures.py` does exactly that and writes
`tests/golden/ref_*.npz`, the fixtures that pin `oracle/` (CPU suite) and the HIP path (`-m gpu` suite).

What it is: a lazy dataflow graph (every `tf.*` call returns a `Tensor` node holding a closure; `Session.run` evaluates the
fetched nodes once per call with memoisation, like one TF1 step), `torch` for the arithmetic and `torch.autograd` for
`tf.gradients` / `Optimizer.compute_gradients`.  Only the ~70 entry points the reference touches exist.  The semantics of
each op follow TensorFlow 1.x's published documentation (the arithmetic lives in that third-party dependency, not in
the reference): `nn.sigmoid_cross_entropy_with_logits` = max(x,0) - x*z + log(1+exp(-|x|)); `nn.l2_normalize` =
x * rsqrt(max(sum(x^2), eps)); `nn.l2_loss` = sum(x^2)/2; `nn.dropout(x, keep)` = x/keep * floor(keep + U[0,1));
`reduce_max` splits its gradient evenly among tied maxima; `nn.top_k` orders ties by lower index first;
`train.AdamOptimizer`: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), m,v moving averages, var -= lr_t*m/(sqrt(v)+eps).
Random ops (`random_uniform`, `random_shuffle`, `fixed_unigram_candidate_sampler`, dropout masks) draw from a NumPy
stream seeded by `tf.set_random_seed` and are LOGGED (`tf.shim.log`) so that a fixture can carry the permutations /
negatives / masks the run used -- TF's own streams are not reproducible.

Nothing under graphsage_amd/, bench.py or oracle/ imports this package.
"""
import builtins as _bi
import contextlib
import types

import numpy as np
import torch

__version__ = "1.8.0-shim"


# ----------------------------------------------------------------------------------------------------------------
# dtypes, global state
# ----------------------------------------------------------------------------------------------------------------
class DType(object):
    def __init__(self, name, kind):
        self.name, self.kind = name, kind

    def torch(self):
        if self.kind == "f":
            return shim.real
        return {"int32": torch.int64, "int64": torch.int64, "bool": torch.bool}[self.name]

    def __repr__(self):
        return "tf." + self.name


float32 = DType("float32", "f")
float64 = DType("float64", "f")
int32 = DType("int32", "i")
int64 = DType("int64", "i")
bool = DType("bool", "b")  # noqa: A001  (TF exports tf.bool)


class _Shim(object):
    """Process-wide state of the stand-in (the 'default graph' + the knobs a fixture generator needs)."""

    def __init__(self):
        self.real = torch.float32        # what tf.float32 computes in; torch.float64 gives a high-precision twin run
        self.reset()

    def reset(self, seed=0):
        self.rng = np.random.RandomState(seed)
        self.variables = []
        self.scope = []
        self.log = {"shuffle": [], "unigram": [], "dropout": []}
        self.inject_shuffle = []         # permutations consumed (FIFO) by random_shuffle before the RNG is used

    def set_real(self, name):
        self.real = {"float32": torch.float32, "float64": torch.float64}[name]


shim = _Shim()


def set_random_seed(seed):
    shim.rng = np.random.RandomState(seed)


def reset_default_graph():
    shim.reset()


class MissingFeed(Exception):
    pass


# ----------------------------------------------------------------------------------------------------------------
# graph nodes
# ----------------------------------------------------------------------------------------------------------------
def _flat_tensors(x, out):
    if isinstance(x, Tensor):
        out.append(x)
    elif isinstance(x, (list, tuple)):
        for i in x:
            _flat_tensors(i, out)
    return out


class _Ctx(object):
    def __init__(self, feed):
        self.feed = feed or {}
        self.cache = {}
        self.post = []


def _ev(x, ctx):
    if isinstance(x, Tensor):
        return x._eval(ctx)
    if isinstance(x, (list, tuple)):
        return type(x)(_ev(i, ctx) for i in x)
    return x


def _as_torch(v, dtype=None):
    if isinstance(v, torch.Tensor):
        return v if dtype is None else v.to(dtype)
    a = np.asarray(v)
    if dtype is None:
        dtype = shim.real if a.dtype.kind == "f" else (torch.bool if a.dtype.kind == "b" else torch.int64)
    return torch.as_tensor(a.astype(np.float64) if a.dtype.kind == "f" else a).to(dtype)


class TensorShape(object):
    def __init__(self, dims):
        self.dims = list(dims)

    def as_list(self):
        return list(self.dims)

    def __getitem__(self, i):
        return self.dims[i]

    def __len__(self):
        return len(self.dims)


class Tensor(object):
    """A node of the lazy graph.  `fn(*evaluated_args)` is run at most once per Session.run."""

    def __init__(self, fn, args=(), name=None, static_shape=None):
        self.fn, self.args, self.name = fn, args, name
        self.inputs = _flat_tensors(args, [])
        self.static_shape = static_shape
        self.op_type = None

    def _eval(self, ctx):
        k = id(self)
        if k not in ctx.cache:
            ctx.cache[k] = self.fn(*[_ev(a, ctx) for a in self.args])
        return ctx.cache[k]

    def get_shape(self):
        if self.static_shape is None:
            raise ValueError("static shape unknown for %r" % (self.name,))
        return TensorShape(self.static_shape)

    # ---- operators (python numbers / numpy arrays on the other side are fine)
    def __add__(self, o): return _op(lambda a, b: a + b, self, o)
    def __radd__(self, o): return _op(lambda a, b: b + a, self, o)
    def __sub__(self, o): return _op(lambda a, b: a - b, self, o)
    def __rsub__(self, o): return _op(lambda a, b: b - a, self, o)
    def __mul__(self, o): return _op(lambda a, b: a * b, self, o)
    def __rmul__(self, o): return _op(lambda a, b: b * a, self, o)
    def __truediv__(self, o): return _op(lambda a, b: a / b, self, o)
    def __rtruediv__(self, o): return _op(lambda a, b: b / a, self, o)
    __div__, __rdiv__ = __truediv__, __rtruediv__
    def __neg__(self): return _op(lambda a: -a, self)
    def __getitem__(self, idx): return _op(lambda a: a[idx], self)
    __hash__ = object.__hash__


def _op(fn, *args, **kw):
    def run(*vals):
        vals = [(_as_torch(v) if isinstance(v, np.ndarray) else v) for v in vals]
        return fn(*vals)
    return Tensor(run, args, name=kw.get("name"))


class Operation(Tensor):
    """A node whose value is None (assign / apply_gradients / initializers / summaries)."""


class Placeholder(Tensor):
    def __init__(self, dtype, shape=None, name=None, default=None):
        Tensor.__init__(self, None, (), name=name, static_shape=None if shape is None else list(
            shape if isinstance(shape, (list, tuple)) else [shape]))
        self.dtype, self.default = dtype, default

    def _eval(self, ctx):
        k = id(self)
        if k not in ctx.cache:
            if self in ctx.feed:
                v = ctx.feed[self]
            elif self.default is not None:
                v = self.default
            else:
                raise MissingFeed("placeholder %r was not fed" % (self.name,))
            ctx.cache[k] = _as_torch(v, self.dtype.torch())
        return ctx.cache[k]


def placeholder(dtype, shape=None, name=None):
    return Placeholder(dtype, shape, name)


def placeholder_with_default(input, shape=None, name=None):  # noqa: A002
    return Placeholder(float32 if isinstance(input, float) else int32, shape, name, default=input)


class Variable(Tensor):
    def __init__(self, initial_value=None, trainable=True, name=None, dtype=None):
        Tensor.__init__(self, None, (), name="/".join(shim.scope + [name or "Variable"]) + ":0")
        self.trainable = trainable
        self.value = None
        self._init = initial_value
        shim.variables.append(self)
        try:
            self._initialize(_Ctx(None))
        except MissingFeed:
            if isinstance(initial_value, Tensor) and initial_value.static_shape is not None:
                self.static_shape = initial_value.static_shape

    def _initialize(self, ctx):
        v = _ev(self._init, ctx)
        v = _as_torch(v).detach().clone()
        if self.trainable and v.dtype.is_floating_point:
            v.requires_grad_(True)
        self.value = v
        self.static_shape = list(v.shape)
        self._init = None

    def _eval(self, ctx):
        if self.value is None:
            raise RuntimeError("variable %s used before tf.global_variables_initializer()" % self.name)
        return self.value

    def _set(self, new):
        with torch.no_grad():
            if self.value.shape == new.shape:
                self.value.copy_(new.to(self.value.dtype))
            else:
                self.value = new.detach().clone()


class GraphKeys(object):
    GLOBAL_VARIABLES = "variables"
    TRAINABLE_VARIABLES = "trainable_variables"


def get_collection(key, scope=None):
    vs = [v for v in shim.variables if key == GraphKeys.GLOBAL_VARIABLES or v.trainable]
    return [v for v in vs if scope is None or v.name.startswith(scope)]


def global_variables():
    return list(shim.variables)


def trainable_variables():
    return [v for v in shim.variables if v.trainable]


def global_variables_initializer():
    def run_init(ctx):
        for v in shim.variables:
            if v.value is None:
                v._initialize(ctx)
    return _CtxOp(run_init)


class _CtxOp(Operation):
    """Operation whose body needs the run context (feeds) or must run AFTER every fetch of the same run was evaluated."""

    def __init__(self, body, args=(), deferred=False):
        Tensor.__init__(self, None, args)
        self.body, self.deferred = body, deferred

    def _eval(self, ctx):
        k = id(self)
        if k not in ctx.cache:
            ctx.cache[k] = None
            if self.deferred:
                vals = [_ev(a, ctx) for a in self.args]
                ctx.post.append(lambda: self.body(*vals))
            else:
                self.body(ctx)
        return None


def assign(ref, value, name=None):
    return _CtxOp(lambda v: ref._set(_as_torch(v)), (value,), deferred=True)


class _VarScope(object):
    def reuse_variables(self):
        pass


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    shim.scope.append(name)
    try:
        yield _VarScope()
    finally:
        shim.scope.pop()


@contextlib.contextmanager
def name_scope(name):
    yield name


def _glorot_uniform(shape):
    fan_in, fan_out = shape[0], shape[-1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return shim.rng.uniform(-lim, lim, size=shape).astype(np.float32)


def get_variable(name, shape=None, dtype=None, initializer=None, regularizer=None, trainable=True):
    init = (initializer or _glorot_uniform)(list(shape))       # TF1 default initializer: glorot_uniform
    return Variable(init, trainable=trainable, name=name)


# ----------------------------------------------------------------------------------------------------------------
# sessions
# ----------------------------------------------------------------------------------------------------------------
class ConfigProto(object):
    def __init__(self, **kw):
        self.gpu_options = types.SimpleNamespace(allow_growth=False, per_process_gpu_memory_fraction=1.0)
        self.allow_soft_placement = False
        self.__dict__.update(kw)


class Session(object):
    def __init__(self, target="", graph=None, config=None):
        self.graph = None

    def run(self, fetches, feed_dict=None):
        ctx = _Ctx(feed_dict)
        single = not isinstance(fetches, (list, tuple))
        vals = [_ev(f, ctx) for f in ([fetches] if single else fetches)]
        out = [_to_numpy(v) for v in vals]
        for fn in ctx.post:             # variable updates happen after every fetch read the pre-update values
            fn()
        return out[0] if single else out

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _to_numpy(v):
    if isinstance(v, torch.Tensor):
        a = v.detach().cpu().numpy()
        return a.copy() if a.ndim else a[()]
    if isinstance(v, (list, tuple)):
        return type(v)(_to_numpy(i) for i in v)
    return v


# ----------------------------------------------------------------------------------------------------------------
# array / math ops
# ----------------------------------------------------------------------------------------------------------------
def _int(x):
    return int(x.item()) if isinstance(x, torch.Tensor) else int(x)


def _shape_list(s):
    if isinstance(s, torch.Tensor):
        return [int(i) for i in s.reshape(-1).tolist()]
    return [_int(i) for i in s]


def constant(value, dtype=None, shape=None, name=None):
    t = _as_torch(value, None if dtype is None else dtype.torch())
    if shape is not None:
        t = t.expand(*shape).clone() if t.numel() == 1 else t.reshape(*shape)
    return Tensor(lambda: t, (), static_shape=list(t.shape))


def zeros(shape, dtype=float32, name=None):
    return Tensor(lambda s: torch.zeros(_shape_list(s), dtype=dtype.torch()), (shape,))


def ones(shape, dtype=float32, name=None):
    return Tensor(lambda s: torch.ones(_shape_list(s), dtype=dtype.torch()), (shape,))


def zeros_like(x): return _op(torch.zeros_like, x)
def ones_like(x): return _op(torch.ones_like, x)


def random_uniform(shape, minval=0, maxval=None, dtype=float32, seed=None, name=None):
    hi = 1.0 if maxval is None else maxval

    def draw(s):
        # float32-rounded draws also in the float64 twin run, so both runs start from identical weights
        return _as_torch(shim.rng.uniform(minval, hi, size=_shape_list(s)).astype(np.float32), dtype.torch())
    return Tensor(draw, (shape,))


def random_shuffle(value, seed=None, name=None):
    """Shuffles along dimension 0.  The permutation is injected (shim.inject_shuffle FIFO) or drawn, and logged."""
    def run(v):
        n = v.shape[0]
        if shim.inject_shuffle:
            perm = np.asarray(shim.inject_shuffle.pop(0), dtype=np.int64)
            assert sorted(perm.tolist()) == list(_bi.range(n))
        else:
            perm = shim.rng.permutation(n)
        shim.log["shuffle"].append((id(node), perm.copy()))
        return v[torch.as_tensor(perm)]
    node = Tensor(run, (value,))
    node.op_type = "random_shuffle"
    return node


def transpose(a, perm=None, name=None):
    return _op(lambda v: v.permute(*(perm if perm is not None else reversed(_bi.range(v.dim())))), a)


def slice(input_, begin, size, name=None):  # noqa: A001
    def run(v):
        idx = tuple(np.s_[b:(None if s == -1 else b + s)] for b, s in zip(_shape_list(begin), _shape_list(size)))
        return v[idx]
    return _op(run, input_)


def reshape(tensor, shape, name=None):
    return Tensor(lambda v, s: _as_torch(v).reshape(_shape_list(s)), (tensor, shape))


def shape(input, name=None):  # noqa: A002
    return _op(lambda v: torch.tensor(list(v.shape), dtype=torch.int64), input)


def expand_dims(input, axis=None, name=None, dim=None):  # noqa: A002
    return _op(lambda v: v.unsqueeze(axis if axis is not None else dim), input)


def squeeze(input, axis=None, name=None):  # noqa: A002
    return _op(lambda v: v.squeeze() if axis is None else v.squeeze(axis), input)


def concat(values=None, axis=None, name=None, **kw):
    if isinstance(values, int) and not isinstance(axis, int):      # pre-1.0 argument order
        values, axis = axis, values
    return Tensor(lambda vs: torch.cat([_as_torch(v) for v in vs], dim=axis), (list(values),))


def add_n(inputs, name=None):
    def run(vs):
        acc = vs[0]
        for v in vs[1:]:
            acc = acc + v
        return acc
    return Tensor(run, (list(inputs),))


def cast(x, dtype, name=None):
    return _op(lambda v: _as_torch(v).to(dtype.torch()), x)


def _reduce(fn):
    def red(input_tensor, axis=None, keep_dims=False, name=None, keepdims=None, reduction_indices=None):
        ax = axis if axis is not None else reduction_indices
        kd = keep_dims if keepdims is None else keepdims
        return _op(lambda v: fn(v, ax, kd), input_tensor)
    return red


reduce_sum = _reduce(lambda v, ax, kd: v.sum() if ax is None else v.sum(dim=ax, keepdim=kd))
reduce_mean = _reduce(lambda v, ax, kd: v.mean() if ax is None else v.mean(dim=ax, keepdim=kd))
reduce_max = _reduce(lambda v, ax, kd: v.amax() if ax is None else v.amax(dim=ax, keepdim=kd))   # ties share the gradient


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    return _op(lambda x, y: (x.t() if transpose_a else x) @ (y.t() if transpose_b else y), a, b)


def multiply(x, y, name=None): return _op(lambda a, b: a * b, x, y)
def subtract(x, y, name=None): return _op(lambda a, b: a - b, x, y)
def add(x, y, name=None): return _op(lambda a, b: a + b, x, y)
def div(x, y, name=None): return _op(lambda a, b: a / b, x, y)
def maximum(x, y, name=None): return _op(lambda a, b: torch.maximum(_as_torch(a), _as_torch(b).to(_as_torch(a).dtype)), x, y)
def sqrt(x, name=None): return _op(torch.sqrt, x)
def exp(x, name=None): return _op(torch.exp, x)
def log(x, name=None): return _op(torch.log, x)
def abs(x, name=None): return _op(torch.abs, x)  # noqa: A001
def sign(x, name=None): return _op(torch.sign, x)
def negative(x, name=None): return _op(lambda a: -a, x)
def equal(x, y, name=None): return _op(lambda a, b: a == b, x, y)
def argmax(input, axis=None, name=None): return _op(lambda v: v.argmax(dim=axis), input)  # noqa: A002
def stop_gradient(input, name=None): return _op(lambda v: v.detach(), input)  # noqa: A002
def gather(params, indices, name=None): return _op(lambda p, i: p[i], params, indices)


def range(start, limit=None, delta=1, name=None):  # noqa: A001
    return Tensor(lambda a, b: torch.arange(_int(a), _int(b), delta) if b is not None else torch.arange(_int(a)),
                  (start, limit))


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    return _op(lambda v: v.clamp(clip_value_min, clip_value_max), t)


def gradients(ys, xs, name=None):
    single = not isinstance(xs, (list, tuple))
    g = _grad_nodes(ys, [xs] if single else list(xs))
    return g[0] if single else g


def _reachable_variables(t):
    seen, stack, out = set(), [t], set()
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        if isinstance(n, Variable):
            out.add(id(n))
        stack.extend(n.inputs)
    return out


def _grad_nodes(loss, xs):
    """d loss / d x for every x (None where the graph holds no path, as TF does); one shared autograd call per run."""
    reach = _reachable_variables(loss)
    live = [x for x in xs if not isinstance(x, Variable) or id(x) in reach]

    def all_grads(lv, *xv):
        return torch.autograd.grad(lv, list(xv), allow_unused=True, retain_graph=True)
    shared = Tensor(all_grads, (loss,) + tuple(live))
    out = []
    for x in xs:
        if x not in live:
            out.append(None)
            continue
        i = live.index(x)
        out.append(_op(lambda gs, xv, i=i: gs[i] if gs[i] is not None else torch.zeros_like(xv), shared, x))
    return out


# ----------------------------------------------------------------------------------------------------------------
# tf.nn
# ----------------------------------------------------------------------------------------------------------------
def _embedding_lookup(params, ids, name=None):
    if isinstance(params, (list, tuple)):
        assert len(params) == 1, "partitioned embedding_lookup is not needed by the reference"
        params = params[0]
    return _op(lambda p, i: p[_as_torch(i).long()], params, ids)


def _dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    def run(v, keep):
        keep = float(keep)
        if keep == 1.0:
            return v                       # floor(1 + u) == 1: exact identity
        u = shim.rng.uniform(size=tuple(v.shape))
        mask = np.floor(keep + u)
        shim.log["dropout"].append((id(node), mask.astype(np.uint8)))     # the keep bits; scale = 1/keep
        return v / keep * _as_torch(mask, v.dtype)
    node = _op(run, x, keep_prob)
    node.op_type = "dropout"
    return node


def _l2_normalize(x, dim=None, epsilon=1e-12, name=None, axis=None):
    d = dim if dim is not None else axis
    return _op(lambda v: v * torch.rsqrt(torch.clamp((v * v).sum(dim=d, keepdim=True), min=epsilon)), x)


def _sigmoid_xent(_sentinel=None, labels=None, logits=None, name=None):
    return _op(lambda z, x: torch.clamp(x, min=0) - x * _as_torch(z).to(x.dtype) + torch.log1p(torch.exp(-x.abs())),
               labels, logits)


def _softmax_xent(_sentinel=None, labels=None, logits=None, dim=-1, name=None):
    return _op(lambda z, x: -(_as_torch(z).to(x.dtype) * torch.log_softmax(x, dim=dim)).sum(dim=dim), labels, logits)


def _top_k(input, k=1, sorted=True, name=None):  # noqa: A002
    def run(v, kk):
        vals, idx = torch.sort(v, dim=-1, descending=True, stable=True)     # equal values: lower index first
        return vals[..., :_int(kk)], idx[..., :_int(kk)]
    both = Tensor(run, (input, k))
    return _op(lambda b: b[0], both), _op(lambda b: b[1], both)


def _fixed_unigram_candidate_sampler(true_classes, num_true, num_sampled, unique, range_max, vocab_file="",
                                     distortion=1.0, num_reserved_ids=0, num_shards=1, shard=0, unigrams=(), seed=None,
                                     name=None):
    """P(class) proportional to unigrams[class]**distortion; unique=False: draws with replacement."""
    assert not unique and num_true == 1
    w = np.asarray(unigrams, dtype=np.float64) ** distortion
    p = w / w.sum()

    def run(_true):
        s = shim.rng.choice(range_max, size=num_sampled, replace=True, p=p)
        shim.log["unigram"].append((id(sampled), s.copy()))
        return torch.as_tensor(s, dtype=torch.int64)
    sampled = Tensor(run, (true_classes,))
    true_exp = _op(lambda t: _as_torch(p[t.numpy()] * num_sampled), true_classes)
    samp_exp = _op(lambda s: _as_torch(p[s.numpy()] * num_sampled), sampled)
    return sampled, true_exp, samp_exp


def _unsupported(what):
    def f(*a, **k):
        raise NotImplementedError("tf1_shim: %s is outside the hot path and not emulated" % what)
    return f


nn = types.SimpleNamespace(
    embedding_lookup=_embedding_lookup, dropout=_dropout, l2_normalize=_l2_normalize,
    relu=lambda x, name=None: _op(torch.relu, x), sigmoid=lambda x, name=None: _op(torch.sigmoid, x),
    softmax=lambda x, dim=-1, name=None: _op(lambda v: torch.softmax(v, dim=dim), x),
    tanh=lambda x, name=None: _op(torch.tanh, x),
    l2_loss=lambda t, name=None: _op(lambda v: (v * v).sum() / 2, t),
    sigmoid_cross_entropy_with_logits=_sigmoid_xent, softmax_cross_entropy_with_logits=_softmax_xent,
    top_k=_top_k, fixed_unigram_candidate_sampler=_fixed_unigram_candidate_sampler,
    dynamic_rnn=_unsupported("nn.dynamic_rnn (SeqAggregator)"))
sigmoid = nn.sigmoid
tanh = nn.tanh


# ----------------------------------------------------------------------------------------------------------------
# tf.train
# ----------------------------------------------------------------------------------------------------------------
class _Optimizer(object):
    def compute_gradients(self, loss, var_list=None):
        vs = var_list if var_list is not None else trainable_variables()
        return list(zip(_grad_nodes(loss, list(vs)), vs))

    def minimize(self, loss, var_list=None):
        return self.apply_gradients(self.compute_gradients(loss, var_list))


class AdamOptimizer(_Optimizer):
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, name="Adam"):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.t, self.slots = 0, {}

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        gv = [(g, v) for g, v in grads_and_vars if g is not None]

        def update(*gvals):
            self.t += 1
            with torch.no_grad():
                for g, (_, var) in zip(gvals, gv):
                    dt = var.value.dtype
                    m, v = self.slots.setdefault(id(var), (torch.zeros_like(var.value), torch.zeros_like(var.value)))
                    g = g.to(dt)
                    lr_t = torch.tensor(self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t), dtype=dt)
                    m.mul_(self.b1).add_((1 - self.b1) * g)
                    v.mul_(self.b2).add_((1 - self.b2) * g * g)
                    var.value.sub_(lr_t * m / (torch.sqrt(v) + self.eps))
        return _CtxOp(update, tuple(g for g, _ in gv), deferred=True)


class GradientDescentOptimizer(_Optimizer):
    def __init__(self, learning_rate, name="GradientDescent"):
        self.lr = learning_rate

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        gv = [(g, v) for g, v in grads_and_vars if g is not None]

        def update(*gvals):
            with torch.no_grad():
                for g, (_, var) in zip(gvals, gv):
                    var.value.sub_(self.lr * g.to(var.value.dtype))
        return _CtxOp(update, tuple(g for g, _ in gv), deferred=True)


class _Saver(object):
    def __init__(self, *a, **k):
        pass

    save = restore = _unsupported("train.Saver")


train = types.SimpleNamespace(AdamOptimizer=AdamOptimizer, GradientDescentOptimizer=GradientDescentOptimizer, Saver=_Saver)


# ----------------------------------------------------------------------------------------------------------------
# tf.contrib, tf.summary, tf.app
# ----------------------------------------------------------------------------------------------------------------
class _LSTMStub(object):
    def __init__(self, *a, **k):
        pass

    zero_state = _unsupported("contrib.rnn.BasicLSTMCell (SeqAggregator)")


contrib = types.SimpleNamespace(
    layers=types.SimpleNamespace(xavier_initializer=lambda uniform=True, seed=None, dtype=None: _glorot_uniform,
                                 l2_regularizer=lambda scale, scope=None: (lambda w: None)),
    rnn=types.SimpleNamespace(BasicLSTMCell=_LSTMStub))


class _FileWriter(object):
    def __init__(self, *a, **k):
        pass

    def add_summary(self, *a, **k):
        pass

    def close(self):
        pass


summary = types.SimpleNamespace(scalar=lambda *a, **k: None, histogram=lambda *a, **k: None,
                                merge_all=lambda *a, **k: Operation(lambda: None, ()), FileWriter=_FileWriter)


class _Flags(object):
    def __getattr__(self, k):
        raise AttributeError("flag %r was never defined" % k)


class _FlagsModule(object):
    def __init__(self):
        self.FLAGS = _Flags()

    def _define(self, name, default, doc=None):
        self.FLAGS.__dict__.setdefault(name, default)

    DEFINE_string = DEFINE_integer = DEFINE_float = DEFINE_boolean = DEFINE_bool = _define


flags = _FlagsModule()
app = types.SimpleNamespace(flags=flags, run=lambda main=None, argv=None: main(argv or []))
