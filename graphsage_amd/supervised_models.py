"""SupervisedGraphsage on the gfx950 engine -- constructor/attribute surface of
graphsage/supervised_models.py:10-126.

    model = SupervisedGraphsage(num_classes, placeholders, features, adj_info, degrees, layer_infos,
                                concat=True, aggregator_type="mean", model_size="small",
                                sigmoid_loss=False, identity_dim=0)
    loss, preds = model.train_step(feed_dict)      # sess.run([opt_op, loss, preds], feed_dict)   (supervised_train.py:275)
    loss, preds = model.eval_step(feed_dict)       # sess.run([preds, loss], feed_dict)           (:76-77, :101-102)

`FLAGS.learning_rate` / `FLAGS.weight_decay` (read from module-level flags in the reference,
supervised_models.py:73,106-108) are explicit keyword arguments here.
"""
import numpy as np
import torch

from . import ops
from .layers import Dense, Rows, identity
from .models import SampleAndAggregate
from .ops import Mat


class SupervisedGraphsage(SampleAndAggregate):
    """Implementation of supervised GraphSAGE."""

    def __init__(self, num_classes, placeholders, features, adj, degrees, layer_infos, concat=True,
                 aggregator_type="mean", model_size="small", sigmoid_loss=False, identity_dim=0,
                 learning_rate=0.01, weight_decay=0.0, world_size=1, rank=0, **kwargs):
        super(SupervisedGraphsage, self).__init__(placeholders, features, adj, degrees, layer_infos, concat=concat,
                                                  aggregator_type=aggregator_type, model_size=model_size,
                                                  identity_dim=identity_dim, **kwargs)
        self.inputs1 = placeholders["batch"]
        self.num_classes = num_classes
        self.sigmoid_loss = sigmoid_loss
        self.learning_rate = float(learning_rate)
        self.weight_decay = float(weight_decay)
        self.world_size = int(world_size)
        self.rank = int(rank)
        self.row_offset = 0
        self.label_table = None   # optional device-resident [N+1, C] label matrix (device fast path)
        self._graphs = {}
        self._warm = set()
        self.use_graphs = True
        self.grad_hook = None     # called between backward and the optimizer (RCCL all-reduce for DP)
        self.build()

    # ------------------------------------------------------------------------------ build (:78-100)
    def build(self):
        e = self.engine
        self.num_samples = [layer_info.num_samples for layer_info in self.layer_infos]
        self.aggregators = self.make_aggregators(self.dims, self.num_samples, self.concat, self.model_size)
        dim_mult = 2 if self.concat else 1
        self.node_pred = Dense(dim_mult * self.dims[-1], self.num_classes, dropout=self.placeholders['dropout'],
                               act=identity)
        e.finalize()
        self.loss_dev = torch.zeros(1, dtype=torch.float32, device=e.device)

    # ------------------------------------------------------------------------------ one step
    def _samplers(self):
        seen = []
        for li in self.layer_infos:
            if li.neigh_sampler not in seen:
                seen.append(li.neigh_sampler)
        return seen

    def _fused_head_ok(self, d):
        C = self.num_classes
        return (getattr(self, "fuse_head", True) and d <= 1024 and C <= 256
                and (d * (((C + 3) & ~3) | 1) + 4 * d) * 4 <= 160 * 1024)

    def _forward(self, batch, labels, n, train=False):
        """sample -> aggregate -> l2_normalize -> node_pred -> loss/preds  (supervised_models.py:79-92,102-126)."""
        e = self.engine
        self.reset_tapes()
        del self.node_pred._saved[:]
        for s in self._samplers():
            s.new_step()
        samples1, support_sizes1 = self.sample(batch, self.layer_infos, n)
        out, _ = self.aggregate(samples1, [self.features], self.dims, self.num_samples, support_sizes1, batch_size=n,
                                aggregators=self.aggregators, concat=self.concat, model_size=self.model_size)
        self.samples1 = samples1
        self.agg_out = out
        C = self.num_classes
        self.outputs1 = e.ws_mat("outputs1", n, out.d)
        self._loss_rows = e.ws_f32("loss_rows", n)
        self.preds = e.ws_mat("preds", n, C)
        self._dlogits = e.ws_mat("dlogits", n, C)
        self._head_fused = self._fused_head_ok(out.d)
        if self._head_fused:
            # l2_normalize (:85) + Dense head (:88-92) + loss/preds (:111-126) + their gradients: ONE launch
            self.node_preds = e.ws_mat("node_preds", n, C)
            self._d_agg_out = e.ws_mat("d_agg_out", n, out.d) if train else None
            ops.head_fwd_bwd(out, n, self.node_pred.vars['weights'].value, self.node_pred.vars['bias'].value.buf, labels,
                             C, self.sigmoid_loss, self.outputs1, self.node_preds, self.preds, self._dlogits,
                             self._loss_rows, self._d_agg_out, stream=e.stream)
        else:
            self._inv_norm = e.ws_f32("inv_norm", n)
            ops.l2norm_fwd(out, n, self.outputs1, self._inv_norm, stream=e.stream)                      # :85
            self.node_preds = self.node_pred(Rows(self.outputs1, None, requires_grad=True))             # :88-92
            ops.class_loss(self.node_preds, labels, n, C, self.sigmoid_loss, self._loss_rows, self.preds,
                           self._dlogits, stream=e.stream)                                               # :111-126
        # loss = weight decay terms (:104-108) + mean classification loss (the mean is added by the step epilogue)
        self._loss_accumulate = False
        if self.weight_decay != 0.0:
            first = True
            for v in e.variables:
                if v.decay:
                    ops.call("gs_sumsq_scaled", v.value.ptr, v.size, 0.5 * self.weight_decay,
                             self.loss_dev.data_ptr(), 0 if first else 1, e.stream)
                    first = False
            self._loss_accumulate = not first

    def _backward(self, n, fuse_adam):
        """Reverse of _forward.  Every weight gradient of the pass is ONE grouped launch; the slab reduction
        (+ weight decay, :104-108) and -- on a single GPU -- clip + Adam (:96-99) are ONE more launch."""
        e = self.engine
        e.begin_backward()
        if self._head_fused:
            e.wgrad(self.node_pred.vars['weights'], self.outputs1, None, self._dlogits, 0, n)
            e.bgrad(self.node_pred.vars['bias'], self._dlogits, n, self.num_classes)
            d_out = self._d_agg_out
        else:
            d_outputs1 = self.node_pred.backward(self._dlogits, need_input_grad=True)
            d_out = e.ws_mat("d_agg_out", n, self.agg_out.d)
            ops.l2norm_bwd(d_outputs1, self.outputs1, self._inv_norm, n, d_out, stream=e.stream)
        self.aggregate_backward(d_out)
        e.finish_backward(self.weight_decay, fuse_adam=fuse_adam, lr=self.learning_rate, clip=5.0, grad_scale=1.0)

    def _epilogue(self, n, **counters):
        self.engine.advance(loss_rows=self._loss_rows, n=n, loss_out=self.loss_dev, accumulate=self._loss_accumulate,
                            **counters)

    def _optimize(self):
        """Data-parallel path: clip_by_value(+-5) + Adam (:96-99) after the RCCL all-reduce.  The local gradient
        is that of the local batch mean; the hook sums over ranks and grad_scale divides by world_size."""
        e = self.engine
        e.adam(self.learning_rate, clip=5.0, grad_scale=1.0 / self.world_size)
        e.advance(step=1)

    # ------------------------------------------------------------------------------ feeds
    def _stage_feed(self, feed_dict):
        """Copy the host feed (batch ids + label matrix) into persistent device buffers."""
        e = self.engine
        ph = self.placeholders
        batch = np.ascontiguousarray(np.asarray(feed_dict[ph['batch']]), dtype=np.int32)
        n = int(batch.shape[0])
        bs = feed_dict.get(ph['batch_size'], n)
        assert int(bs) == n, "batch_size feed (%s) != len(batch) (%d)" % (bs, n)
        drop = feed_dict.get(ph['dropout'], 0.0)
        if float(drop) != 0.0:
            raise NotImplementedError("dropout > 0 is not implemented in the gfx950 kernels yet")
        batch_dev = self.ids_buffer(n)[0][:n]     # head of the contiguous id buffer (see models.sample)
        batch_dev.copy_(torch.from_numpy(batch))
        labels = np.ascontiguousarray(np.asarray(feed_dict[ph['labels']]), dtype=np.float32)
        labels_dev = e.ws_mat("labels", n, self.num_classes)
        labels_dev.buf[:, : self.num_classes].copy_(torch.from_numpy(labels.reshape(n, self.num_classes)))
        torch.cuda.current_stream().synchronize()
        return batch_dev, labels_dev, n

    def _run(self, key, fn):
        """Eager on first use, captured into a hipGraph on the second, replayed afterwards."""
        e = self.engine
        g = self._graphs.get(key)
        if g is not None:
            g.launch()
            return
        if not self.use_graphs or key not in self._warm or self._needs_host_rng():
            fn()
            self._warm.add(key)
            return
        g = ops.Graph(e.stream)
        g.begin()
        try:
            fn()
        finally:
            g.end()
        self._graphs[key] = g
        g.launch()

    def _needs_host_rng(self):
        from .neigh_samplers import PaddedAdjacency
        return any(isinstance(s.adj_info.current, PaddedAdjacency) for s in self._samplers())

    def _adj_version(self):
        return tuple(id(s.adj_info.current) for s in self._samplers())

    # ------------------------------------------------------------------------------ public steps
    def train_step(self, feed_dict, fetch=True):
        batch_dev, labels_dev, n = self._stage_feed(feed_dict)
        return self._train_on_device(batch_dev, labels_dev, n, fetch)

    def eval_step(self, feed_dict, fetch=True):
        batch_dev, labels_dev, n = self._stage_feed(feed_dict)
        e = self.engine
        self._run(("eval", n, self._adj_version()), lambda: (self._forward(batch_dev, labels_dev, n), self._epilogue(n, clock=1)))
        return self._fetch(n) if fetch else None

    def _train_on_device(self, batch_dev, labels_dev, n, fetch=True, prologue=None, cursor=None, key="train"):
        e = self.engine
        fused = self.grad_hook is None

        def fwd_bwd():
            if prologue is not None:
                prologue()
            self._forward(batch_dev, labels_dev, n, train=True)
            self._backward(n, fuse_adam=fused)
            self._epilogue(n, step=1 if fused else 0, clock=1, cursor=cursor, cursor_delta=n if cursor is not None else 0)

        if fused:
            self._run((key, n, self._adj_version()), fwd_bwd)     # the whole step: one hipGraph
        else:
            self._run((key + "_fb", n, self._adj_version()), fwd_bwd)
            self.grad_hook(self)              # RCCL all-reduce of engine.grads (ordered by stream events)
            self._run(("opt",), self._optimize)
        return self._fetch(n) if fetch else None

    def _fetch(self, n):
        self.engine.sync()
        loss = float(self.loss_dev.item())
        preds = self.preds.view()[:n].detach().cpu().numpy()
        return loss, preds

    # ------------------------------------------------------------------------------ device-resident epoch
    def attach_device_epoch(self, order, label_table):
        """Device fast path: epoch order + label table live in HBM; a step is one hipGraph replay with
        no host->device traffic.  `order`: int32 node ids (this rank's shard); `label_table`: [N+1, C]."""
        e = self.engine
        self._order = torch.from_numpy(np.ascontiguousarray(order, dtype=np.int32)).to(e.device)
        self._cursor = torch.zeros(1, dtype=torch.int64, device=e.device)
        if not isinstance(label_table, Mat):
            label_table = Mat.from_numpy(np.asarray(label_table, dtype=np.float32), e.device)
        self.label_table = label_table
        torch.cuda.synchronize()

    def set_epoch_order(self, order):
        self.engine.sync()  # steps still queued on the engine stream read the old order / cursor
        self._order.copy_(torch.from_numpy(np.ascontiguousarray(order, dtype=np.int32)))
        self._cursor.zero_()
        torch.cuda.synchronize()

    def train_step_device(self, n, fetch=False):
        """One training step on the next n ids of the device-resident epoch order."""
        e = self.engine
        batch_dev = self.ids_buffer(n)[0][:n]
        labels_dev = e.ws_mat("labels", n, self.num_classes)

        def stage():   # batch selection + label gather (minibatch.py:264-274, 302-307) ride along with the sampler launch
            self._pending_stage = (self._order, self._cursor, self.label_table, labels_dev)

        return self._train_on_device(batch_dev, labels_dev, n, fetch, prologue=stage, cursor=self._cursor, key="dtrain")

    def predict(self):
        """sigmoid / softmax of the logits (supervised_models.py:122-126); filled by the last step."""
        return self.preds
