"""SampleAndAggregate: the sample -> gather -> aggregate schedule of graphsage/models.py:187-330 on
the gfx950 engine, keeping the reference's method names and argument meaning:

    samples, support_sizes = model.sample(inputs, layer_infos, batch_size)
    out, aggregators       = model.aggregate(samples, [features], dims, num_samples, support_sizes,
                                             batch_size, aggregators, name, concat, model_size)

TF built a static graph once and ran it with sess.run; here the same Python runs eagerly on the
engine's HIP stream, is captured into a hipGraph on its second execution for a given batch size,
and replayed afterwards.  `aggregate_backward` is the hand-written reverse schedule (TF:
optimizer.compute_gradients, models.py:379).
"""
from collections import namedtuple

import numpy as np
import torch

from . import ops
from .aggregators import GCNAggregator, MaxPoolingAggregator, MeanAggregator, MeanPoolingAggregator
from .engine import get_engine
from .layers import Rows, identity, relu
from .ops import Mat

# SAGEInfo is a namedtuple that specifies the parameters of the recursive GraphSAGE layers
# (graphsage/models.py:180-185)
SAGEInfo = namedtuple("SAGEInfo",
                      ['layer_name',     # name of the layer (to get feature embedding etc.)
                       'neigh_sampler',  # callable neigh_sampler constructor
                       'num_samples',
                       'output_dim'])    # the output (i.e., hidden) dimension

_AGGREGATORS = {
    "mean": MeanAggregator,
    "maxpool": MaxPoolingAggregator,
    "meanpool": MeanPoolingAggregator,
    "gcn": GCNAggregator,
}


def _aggregator_cls(aggregator_type):
    """Dispatch of models.py:211-222 / supervised_models.py:34-45 ('seq' is out of scope)."""
    if aggregator_type == "seq":
        raise NotImplementedError("SeqAggregator (LSTM) is outside the MI355X hot-path scope")
    if aggregator_type not in _AGGREGATORS:
        raise Exception("Unknown aggregator: ", aggregator_type)
    return _AGGREGATORS[aggregator_type]


class Placeholder(object):
    """Host-side feed slot standing in for tf.placeholder (supervised_train.py:112-120)."""

    def __init__(self, name, default=None):
        self.name = name
        self.value = default

    def __repr__(self):
        return "<Placeholder %s>" % self.name


def device_features(features, engine=None):
    """Upload the [N+1, F] feature table (row N = zero pad row, supervised_train.py:133-135) with a
    leading dimension of whole 128-byte lines (F=602 -> ld=608)."""
    e = engine or get_engine()
    if isinstance(features, Mat):
        return features
    return Mat.from_numpy(np.asarray(features, dtype=np.float32), e.device, ld_multiple=32)


class SampleAndAggregate(object):
    """Base implementation of GraphSAGE (graphsage/models.py:187-405); the supervised subclass is in
    supervised_models.py.  The unsupervised objective (_build/_loss/_accuracy, :332-405) is a
    "next" row."""

    def __init__(self, placeholders, features, adj, degrees, layer_infos, concat=True, aggregator_type="mean",
                 model_size="small", identity_dim=0, **kwargs):
        allowed_kwargs = {'name', 'logging', 'model_size'}
        for kwarg in kwargs.keys():
            assert kwarg in allowed_kwargs, 'Invalid keyword argument: ' + kwarg
        self.name = kwargs.get('name') or self.__class__.__name__.lower()
        self.engine = get_engine()
        self.aggregator_cls = _aggregator_cls(aggregator_type)
        self.aggregator_type = aggregator_type
        self.model_size = model_size
        self.adj_info = adj
        if identity_dim > 0:
            raise NotImplementedError("identity_dim > 0 (trainable node embeddings) is a 'next' row (SURVEY §8f N4)")
        if features is None:
            raise Exception("Must have a positive value for identity feature dimension if no input features given.")
        self.features = device_features(features, self.engine)
        self.degrees = degrees
        self.concat = concat
        self.dims = [self.features.d + identity_dim]
        self.dims.extend([layer_infos[i].output_dim for i in range(len(layer_infos))])
        self.placeholders = placeholders
        self.batch_size = placeholders.get("batch_size") if placeholders else None
        self.layer_infos = layer_infos
        self.aggregators = None
        self._tape = None

    # ------------------------------------------------------------------------------ sample (S2)
    def ids_buffer(self, batch_size, layer_infos=None, parity=None):
        """One contiguous int32 buffer [batch | hop-1 samples | hop-2 samples | ...] so that the rows of all hops
        of a layer are adjacent (lets `aggregate` run every hop of a layer in one launch).  Returns
        (buffer, offsets) with offsets[k] = start of samples[k]."""
        layer_infos = layer_infos or self.layer_infos
        sizes = [batch_size]
        support = 1
        for k in range(len(layer_infos)):
            support *= layer_infos[len(layer_infos) - k - 1].num_samples
            sizes.append(batch_size * support)
        offsets = [0]
        for sz in sizes:
            offsets.append(offsets[-1] + sz)
        if parity is None:
            parity = getattr(self, "_parity", 0)
        buf = self.engine.ws_i32(("ids_all", tuple(sizes), parity), offsets[-1])
        return buf, offsets

    def sample(self, inputs, layer_infos, batch_size=None):
        """Sample neighbors to be the supportive fields for multi-layer convolutions
        (models.py:254-275).  `inputs`: int32 device vector of batch node ids.  When `inputs` is the head of
        the model's contiguous id buffer, the sampled hops are written right behind it."""
        if batch_size is None:
            batch_size = inputs.numel()
        samples = [inputs]
        support_size = 1
        support_sizes = [support_size]
        buf, offsets = self.ids_buffer(batch_size, layer_infos)
        contiguous = inputs.data_ptr() == buf.data_ptr() and inputs.numel() == batch_size
        K = len(layer_infos)
        sampler0 = layer_infos[0].neigh_sampler
        from .neigh_samplers import CSRAdjacency
        fused = (contiguous and K <= 3 and all(li.neigh_sampler is sampler0 for li in layer_infos)
                 and isinstance(sampler0.adj_info.current, CSRAdjacency) and getattr(self, "fuse_sampler", True))
        stage = getattr(self, "_pending_stage", None)
        self._pending_stage = None
        if fused:
            # every hop (and, on the device-epoch path, batch + label staging) in ONE launch with an LDS fan-out buffer
            fans = [layer_infos[K - k - 1].num_samples for k in range(K)]
            per_root = 1
            for f in fans[:-1]:
                per_root *= f
            if per_root <= 8192:
                sampler0.fanout(buf, offsets, fans, batch_size, root_offset=getattr(self, "row_offset", 0), stage=stage)
                for k in range(K):
                    support_size *= fans[k]
                    samples.append(buf[offsets[k + 1]: offsets[k + 2]])
                    support_sizes.append(support_size)
                return samples, support_sizes
        if stage is not None:
            order, cursor, table, labels_out = stage
            ops.stage_batch(order, cursor, batch_size, inputs, table, labels_out, stream=self.engine.stream)
        for k in range(len(layer_infos)):
            t = len(layer_infos) - k - 1
            sampler = layer_infos[t].neigh_sampler
            # this rank's first global row at this hop (keeps draws independent of the DP sharding)
            sampler.global_row_offset = getattr(self, "row_offset", 0) * support_size
            support_size *= layer_infos[t].num_samples
            if contiguous:
                sampler.next_out = buf[offsets[k + 1]: offsets[k + 2]]
            node = sampler((samples[k], layer_infos[t].num_samples))
            samples.append(node.reshape(support_size * batch_size))
            support_sizes.append(support_size)
        return samples, support_sizes

    # ------------------------------------------------------------------------------ aggregate (A0/A1)
    def make_aggregators(self, dims, num_samples, concat, model_size, name=None):
        """Aggregator construction of models.py:303-315 (one per layer, last layer identity act)."""
        aggregators = []
        for layer in range(len(num_samples)):
            dim_mult = 2 if concat and (layer != 0) else 1
            if layer == len(num_samples) - 1:
                aggregator = self.aggregator_cls(dim_mult * dims[layer], dims[layer + 1], act=identity,
                                                 dropout=self.placeholders['dropout'], name=name, concat=concat,
                                                 model_size=model_size)
            else:
                aggregator = self.aggregator_cls(dim_mult * dims[layer], dims[layer + 1],
                                                 dropout=self.placeholders['dropout'], name=name, concat=concat,
                                                 model_size=model_size)
            aggregators.append(aggregator)
        return aggregators

    def layer_inputs(self, hidden, layer, batch_size, num_samples, support_sizes, dims, concat):
        """(self_all, neighs, rows, offsets) of one layer: the contiguous self rows of all hops and the per-hop
        neighbor views reshaped as models.py:323-327."""
        from .aggregators import _contiguous
        K = len(num_samples)
        n_hops = K - layer
        dim_mult = 2 if concat and (layer != 0) else 1
        neighs = []
        for hop in range(n_hops):
            neigh_dims = [batch_size * support_sizes[hop], num_samples[K - hop - 1], dim_mult * dims[layer]]
            neighs.append(hidden[hop + 1].reshape(neigh_dims))
        self_all = _contiguous(hidden[:n_hops])
        rows = [hidden[h].n for h in range(n_hops + 1)]
        offsets = [0]
        for r in rows:
            offsets.append(offsets[-1] + r)
        return self_all, neighs, rows, offsets

    def aggregate(self, samples, input_features, dims, num_samples, support_sizes, batch_size=None,
                  aggregators=None, name=None, concat=False, model_size="small", layer0_means=None,
                  layer0_side_jobs=None):
        from .aggregators import _contiguous
        if batch_size is None:
            batch_size = samples[0].numel()
        features = input_features[0] if isinstance(input_features, (list, tuple)) else input_features
        # hidden[h] = embedding_lookup(features, samples[h]) -- kept LAZY (models.py:299)
        hidden = [Rows(features, node_samples, requires_grad=False) for node_samples in samples]
        new_agg = aggregators is None
        if new_agg:
            aggregators = self.make_aggregators(dims, num_samples, concat, model_size, name)
            self.engine.finalize()
        tape = []
        K = len(num_samples)
        for layer in range(K):
            aggregator = aggregators[layer]
            n_hops = K - layer
            self_all, neighs, rows, offsets = self.layer_inputs(hidden, layer, batch_size, num_samples, support_sizes,
                                                               dims, concat)
            if self_all is not None:
                means = layer0_means if layer == 0 else None
                jobs = layer0_side_jobs if layer == 0 else None
                h_all = aggregator.call_hops(self_all, neighs, means=means, side_jobs=jobs)   # all hops, one launch
                outs = [h_all.rows_slice(offsets[h], offsets[h + 1]) for h in range(n_hops)]
                tape.append(("batched", aggregator, rows, offsets, h_all))
            else:                                                           # non-adjacent inputs: hop by hop (:326)
                outs = [aggregator((hidden[hop], neighs[hop])) for hop in range(n_hops)]
                tape.append(("per_hop", aggregator, rows, offsets, outs))
            hidden = [Rows(o, None, requires_grad=True) for o in outs]
        self._tape = tape
        return hidden[0].src, aggregators

    def aggregate_backward(self, d_out):
        """Reverse schedule of `aggregate`.  d_out: Mat = dLoss/d(hidden[0] of the last layer).
        No gradient flows into the feature table (models.py:238: trainable=False)."""
        e = self.engine
        tape = self._tape
        d_cur, pre_masked = d_out, False
        for layer in range(len(tape) - 1, -1, -1):
            mode, agg, rows, offsets, outs = tape[layer]
            if mode != "batched":
                raise NotImplementedError("backward through non-contiguous hop inputs (use the model's id buffer)")
            if layer == 0:
                agg.backward_hops(d_cur, pre_masked)                         # features need no gradient
                break
            prev_mode, prev_agg, prev_rows, prev_offsets, prev_out = tape[layer - 1]
            d_prev = e.ws_mat((self.name, "d_hidden", layer - 1), prev_out.rows, prev_out.d)
            # the previous layer is never the last one, so its activation is relu (models.py:307-314): its
            # relu gradient is fused into the scatter of this layer's input gradients
            mask = prev_out if prev_agg.act_code == ops.ACT_RELU else None
            agg.backward_hops(d_cur, pre_masked, d_prev=d_prev, prev_mask=mask, prev_offsets=prev_offsets)
            d_cur, pre_masked = d_prev, mask is not None

    def reset_tapes(self):
        if self.aggregators:
            for a in self.aggregators:
                a.reset()
        self._tape = None
