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
                                                  identity_dim=identity_dim, learning_rate=learning_rate,
                                                  weight_decay=weight_decay, world_size=world_size, rank=rank,
                                                  _defer_build=True, **kwargs)
        self.inputs1 = placeholders["batch"]
        self.num_classes = num_classes
        self.sigmoid_loss = sigmoid_loss
        self.label_table = None   # optional device-resident [N+1, C] label matrix (device fast path)
        self.build()

    _OUT_ATTRS = ("preds", "samples1", "outputs1", "node_preds", "agg_out", "_loss_rows", "_dlogits", "_loss_accumulate",
                  "_tail_sync", "_tail_sync_n",
                  "_head_fused", "_d_agg_out", "_tape", "_tail_used", "_tail_means", "_tail_dz", "_tail_dh0", "_tail_h0",
                  "_tail_step_advanced")

    # ------------------------------------------------------------------------------ build (:78-100)
    def _root_segments(self, n_roots):
        return None           # one sample() call per step (supervised_models.py:79)

    def build(self):
        e = self.engine
        self.num_samples = [layer_info.num_samples for layer_info in self.layer_infos]
        self.aggregators = self.make_aggregators(self.dims, self.num_samples, self.concat, self.model_size)
        dim_mult = 2 if self.concat else 1
        self.node_pred = Dense(dim_mult * self.dims[-1], self.num_classes, dropout=self.placeholders['dropout'],
                               act=identity)
        e.finalize()
        if self.embeds is not None:
            self._refresh_embeds()
        self.loss_dev = torch.zeros(1, dtype=torch.float32, device=e.device)

    # ------------------------------------------------------------------------------ one step
    def _fused_head_ok(self, d):
        C = self.num_classes
        return (getattr(self, "fuse_head", True) and self._dropout_rate() == 0 and d in (64, 128, 256, 512) and C <= 128
                and (d * (((C + 3) & ~3) | 1) + 4 + 4 * d) * 4 <= 160 * 1024)

    def _tail_ok(self):
        """The fused layer-1 + head launch (gs_sage_tail_fwd_bwd) applies to the supervised two-layer mean model with
        concat -- and, since round 5, to the two-layer GCN model (one weight matrix, the mean over {neighbors} U {self},
        aggregators.py:101-116) --, no dropout, no trainable identity features and shapes the kernel supports."""
        if not getattr(self, "fuse_tail", True) or len(self.layer_infos) != 2 or self.aggregator_type not in ("mean", "gcn"):
            return False
        a1 = self.aggregators[1]
        if a1.bias or self._dropout_rate() != 0 or self.embeds is not None or self.num_samples[-1] > 11:
            return False                                 # (s <= 11: the neighbor rows of a batch node are held in registers)
        if self.aggregator_type == "gcn":
            # layer-1 input = layer-0 output width (GCN ignores concat, the drivers pass 2 * dim: supervised_train.py:175-185)
            return self.dims[2] % 2 == 0 and ops.sage_tail_supported(self.dims[1], self.dims[2] // 2, self.num_classes)
        return self.concat and ops.sage_tail_supported(2 * self.dims[1], self.dims[2], self.num_classes)

    def _tail_weights(self):
        """(W_self, W_neigh, out_dim O, is_gcn) of the fused tail's layer-1 contraction z = [. W_self | . W_neigh] (z has 2 O
        columns): the mean aggregator's two matrices, or the two column halves of the GCN aggregator's one."""
        a1 = self.aggregators[1]
        if self.aggregator_type == "gcn":
            O = self.dims[2] // 2
            W = a1.vars['weights'].value
            return W.cols_slice(0, O), W.cols_slice(O, 2 * O), O, True
        return a1.vars['self_weights'].value, a1.vars['neigh_weights'].value, self.dims[2], False

    def _forward(self, batch, labels, n, train=False, prefetched=None, side_jobs=None, epilogue=None, tail_jobs=None):
        """sample -> aggregate -> l2_normalize -> node_pred -> loss/preds  (supervised_models.py:79-92,102-126).
        `epilogue`: the step's device-counter increments; when the fused tail launch runs it advances them itself
        (and `_backward` then skips the separate epilogue launch)."""
        e = self.engine
        self.reset_tapes()
        del self.node_pred._saved[:]
        if prefetched is None:
            prefetched = self._data_phase(batch, n, getattr(self, "_parity", 0), stage=getattr(self, "_pending_stage", None))
        samples1, support_sizes1, means0 = prefetched
        C = self.num_classes
        # the fused tail needs the hops of layer 0 in ONE contiguous buffer (model.sample on the model's id buffer);
        # anything else takes the per-operator schedule
        contiguous = all(b.data_ptr() == a.data_ptr() + 4 * a.numel() for a, b in zip(samples1[:-1], samples1[1:]))
        self._tail_used = bool(train and contiguous and self._tail_ok())
        out, _ = self.aggregate(samples1, [self.features], self.dims, self.num_samples, support_sizes1, batch_size=n,
                                aggregators=self.aggregators, concat=self.concat, model_size=self.model_size,
                                layer0_means=means0, layer0_side_jobs=side_jobs,
                                _stop_after_layer=0 if self._tail_used else None)
        self.samples1 = samples1
        if self._tail_used and self._tape[0][0] != "batched":
            raise ops._lib.GraphsageAmdError("fused tail needs the contiguous id buffer (model.sample on ids_buffer)")
        self._loss_rows = e.ws_f32("loss_rows", n)
        self.preds = e.ws_mat("preds", n, C)
        self._dlogits = e.ws_mat("dlogits", n, C)
        if self._tail_used:
            # layer 1 + l2_normalize + head + loss + every input gradient down to layer 0's pre-activations: ONE launch
            h0 = self._tape[0][4]                       # [n + n*s, 2*dim_1]: layer-0 outputs of both hops
            W_self1, W_neigh1, O1, gcn1 = self._tail_weights()
            Z = 2 * O1
            s = self.num_samples[len(self.num_samples) - 1]
            self._tail_means = e.ws_mat("tail_means", n, h0.d)
            self.agg_out = e.ws_mat("tail_z", n, Z)
            self.outputs1 = e.ws_mat("outputs1", n, Z)
            self.node_preds = e.ws_mat("node_preds", n, C)
            self._tail_dz = e.ws_mat("tail_dz", n, Z)
            self._tail_dh0 = e.ws_mat((self.name, "d_hidden", 0), h0.rows, h0.d)
            self._tail_h0 = h0
            self._head_fused = True
            counters = []
            self._tail_step_advanced = False
            # hand-over state of the launch's helper workgroups: private to this model (its engine stream)
            # (one buffer per batch size: its layout depends on n, and stale granules of another layout must never be met)
            self._tail_sync = e.ws_i32(("tail_sync", self.name, n, O1), ops.tail_sync_words(n, O1))
            self._tail_sync_n = n
            if epilogue:
                if epilogue.get("step"):
                    counters.append((e.step_dev, epilogue["step"]))
                    self._tail_step_advanced = True
                if epilogue.get("clock"):
                    counters.append((e.sample_clock_dev, epilogue["clock"]))
                if epilogue.get("cursor") is not None and epilogue.get("cursor_delta"):
                    counters.append((epilogue["cursor"], epilogue["cursor_delta"]))
            # split form: the z helpers as their own lean launch (its riders stream at the full HBM rate), then the
            # row-group workgroups; the tail's gather share is divided between the two launches
            jobs_z, jobs_m = [], tail_jobs
            if self.tail_split and tail_jobs:
                jobs_z, jobs_m = ops.split_gather_jobs(tail_jobs, self.cogather_tail_z)
            ids_copy = None
            agg0 = self.aggregators[0]
            agg0.wgrad_ids = None
            if getattr(e, "_sampler_to_wgrad", False) and self.aggregator_type == "mean" and agg0._saved and agg0._saved[-1][5] is not None \
                    and agg0._saved[-1][5].ids is not None:
                src = agg0._saved[-1][5].ids                     # the rows layer 0's self term gathered: [roots | hop-1 ids]
                dst = e.ws_i32(("wgrad_ids", self.name, n), src.numel())
                ids_copy = (src, dst, src.numel())
                agg0.wgrad_ids = dst[:src.numel()]
            ops.sage_tail_fwd_bwd(h0, n, s, W_self1, W_neigh1, O1,
                                  self.node_pred.vars['weights'].value, self.node_pred.vars['bias'].value.buf, labels, C,
                                  self.sigmoid_loss, self._tail_means, self.agg_out, self.outputs1, self.node_preds,
                                  self.preds, self._dlogits, self._loss_rows, dz=self._tail_dz, d_h0=self._tail_dh0,
                                  counters=counters, jobs=jobs_m, stream=e.stream, sync=self._tail_sync,
                                  split=self.tail_split, jobs_z=jobs_z, gcn=gcn1, ids_copy=ids_copy)
        else:
            self.agg_out = out
            self.outputs1 = e.ws_mat("outputs1", n, out.d)
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

    def _backward(self, n, fuse_adam, wgrad_jobs=None, epilogue=None):
        """Reverse of _forward.  Every weight gradient of the pass is ONE grouped launch; the slab reduction
        (+ weight decay, :104-108) and -- on a single GPU -- clip + Adam (:96-99) are ONE more launch."""
        e = self.engine
        e.begin_backward()
        if getattr(self, "_tail_used", False):
            # the fused tail launch already produced every input gradient; queue the weight gradients it feeds
            a1 = self.aggregators[1]
            o = self.dims[2]
            h0 = self._tail_h0
            e.wgrad(self.node_pred.vars['weights'], self.outputs1, None, self._dlogits, 0, n)
            e.bgrad(self.node_pred.vars['bias'], self._dlogits, n, self.num_classes)
            if self.aggregator_type == "gcn":
                e.wgrad(a1.vars['weights'], self._tail_means, None, self._tail_dz, 0, n)   # means = (sum neigh + self) / (s + 1)
            else:
                e.wgrad(a1.vars['self_weights'], h0.rows_slice(0, n), None, self._tail_dz, 0, n)
                e.wgrad(a1.vars['neigh_weights'], self._tail_means, None, self._tail_dz, o, n)
            mode, agg0, rows, offsets, outs = self._tape[0]
            agg0.backward_hops(self._tail_dh0, True, embed_sink=None)
            e.finish_backward(self.weight_decay, fuse_adam=fuse_adam, lr=self.learning_rate, clip=5.0, side_jobs=wgrad_jobs,
                              loss=(self._loss_rows, n, 1.0 / n, self.loss_dev, self._loss_accumulate) if epilogue is not None else None,
                              step_offset=0 if self._tail_step_advanced else 1)
            return
        if self._head_fused:
            e.wgrad(self.node_pred.vars['weights'], self.outputs1, None, self._dlogits, 0, n)
            e.bgrad(self.node_pred.vars['bias'], self._dlogits, n, self.num_classes)
            d_out = self._d_agg_out
        else:
            d_outputs1 = self.node_pred.backward(self._dlogits, need_input_grad=True)
            d_out = e.ws_mat("d_agg_out", n, self.agg_out.d)
            ops.l2norm_bwd(d_outputs1, self.outputs1, self._inv_norm, n, d_out, stream=e.stream)
        # The epilogue (loss mean + device counters) runs BEFORE the backward pass when there is no dropout (its masks are a
        # function of the device clock and the backward pass regenerates them): a later mini-batch's fan-out sampler can
        # then ride in this pass's optimizer launch -- it must see the advanced sampler clock and epoch cursor -- as it does
        # behind the fused tail launch.  (Folding the epilogue into the optimizer launch's last workgroup measured 9 us SLOWER.)
        early = epilogue is not None and self._dropout_rate() == 0
        if early:
            self._epilogue(n, **epilogue)
        self._early_epilogue = early
        advanced = early and bool(epilogue.get("step"))
        self.aggregate_backward(d_out)
        e.finish_backward(self.weight_decay, fuse_adam=fuse_adam, lr=self.learning_rate, clip=5.0, side_jobs=wgrad_jobs,
                          step_offset=0 if advanced else 1)
        if epilogue is not None and not early:
            self._epilogue(n, **epilogue)

    def _epilogue(self, n, **counters):
        self.engine.advance(loss_rows=self._loss_rows, n=n, loss_out=self.loss_dev, accumulate=self._loss_accumulate,
                            **counters)

    def _optimize(self, advanced=False):
        """Data-parallel path: clip_by_value(+-5) + Adam (:96-99) after the RCCL all-reduce.  The local gradient
        is that of the local batch mean; the hook sums over ranks and grad_scale divides by world_size.
        advanced: the step counter was already incremented by an earlier launch of this step."""
        e = self.engine
        e.adam(self.learning_rate, clip=5.0, grad_scale=1.0 / self.world_size, step_offset=0 if advanced else 1)
        if not advanced:
            e.advance(step=1)

    # ------------------------------------------------------------------------------ feeds
    def _stage_feed(self, feed_dict):
        """Copy the host feed (batch ids + label matrix) into persistent device buffers."""
        e = self.engine
        ph = self.placeholders
        # host-fed batches own their buffers (key "h"): a device-epoch batch prefetched into the parity-0/1 buffers
        # must survive an interleaved eval_step / train_step(feed) of the same size
        self._parity = "h"
        self._pending_stage = None
        e.sync()          # earlier steps still queued on the engine stream read these persistent buffers
        batch = np.ascontiguousarray(np.asarray(feed_dict[ph['batch']]), dtype=np.int32)
        n = int(batch.shape[0])
        bs = feed_dict.get(ph['batch_size'], n)
        assert int(bs) == n, "batch_size feed (%s) != len(batch) (%d)" % (bs, n)
        self._feed_dropout(feed_dict)
        batch_dev = self.ids_buffer(n)[0][:n]     # head of the contiguous id buffer (see models.sample)
        batch_dev.copy_(torch.from_numpy(batch))
        labels = np.ascontiguousarray(np.asarray(feed_dict[ph['labels']]), dtype=np.float32)
        labels_dev = e.ws_mat(("labels", "h"), n, self.num_classes)
        labels_dev.buf[:, : self.num_classes].copy_(torch.from_numpy(labels.reshape(n, self.num_classes)))
        torch.cuda.current_stream().synchronize()
        return batch_dev, labels_dev, n

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
        in_graph = self._dp_in_graph()

        def fwd_bwd():
            if prologue is not None:
                prologue()
            ep = dict(step=1 if (fused or in_graph) else 0, clock=1, cursor=cursor, cursor_delta=n if cursor is not None else 0)
            self._forward(batch_dev, labels_dev, n, train=True, epilogue=ep)
            self._backward(n, fuse_adam=fused, epilogue=ep)
            if in_graph:
                self.grad_hook(self)          # ncclAllReduce on the engine stream, recorded in the graph
                self._optimize(advanced=True)

        if fused or in_graph:
            self._run((key if fused else key + "_dp", n, self._adj_version()), fwd_bwd)     # the whole step: one hipGraph
        else:
            self._run((key + "_fb", n, self._adj_version()), fwd_bwd)
            self.grad_hook(self)              # RCCL all-reduce of engine.grads (ordered by stream events)
            self._run(("opt",), self._optimize)
        return self._fetch(n) if fetch else None

    def _fetch(self, n):
        self.engine.sync()
        if getattr(self, "_tail_sync", None) is not None:
            err = ops.tail_sync_error(self._tail_sync, self._tail_sync_n)
            if err:
                raise ops._lib.GraphsageAmdError(
                    "fused tail launch: hand-over between its workgroups failed (flags %d: 1 = a row-group workgroup "
                    "gave up waiting for its helpers, 2 = unexpected arrival count); results since the last fetch are "
                    "invalid -- set model.fuse_tail = False to use the per-operator schedule" % err)
        if hasattr(self.grad_hook, "check"):
            self.grad_hook.check()            # peer-store exchange: a bounded device-side wait that tripped
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
        self._primed = None
        torch.cuda.synchronize()

    def set_epoch_order(self, order):
        self.engine.sync()  # steps still queued on the engine stream read the old order / cursor
        self._order.copy_(torch.from_numpy(np.ascontiguousarray(order, dtype=np.int32)))
        self._cursor.zero_()
        if self._primed is not None:
            # a prefetched-but-unused batch of the old order is dropped: give its sampler-clock tick back so the
            # pipelined schedule draws exactly the samples of the sequential one
            self.engine.sample_clock_dev -= 1
        self._primed = None
        torch.cuda.synchronize()

    def _drop_prefetched(self):
        """Forget the batch the pipeline has already staged (a different batch size follows, e.g. the short last batch
        of an epoch): hand its ids back to the epoch cursor and its tick back to the sampler clock, so the schedule
        draws exactly what the sequential one would."""
        if self._primed is not None:
            self.engine.sync()
            self._cursor -= int(self._primed)
            self.engine.sample_clock_dev -= 1
            self._primed = None
            torch.cuda.synchronize()

    def train_step_device(self, n, fetch=False):
        """One training step on the next n ids of the device-resident epoch order.

        Pipelined (default): the data chain of step t+1 (batch/label staging + fan-out sampling + layer-0
        gather-means; HBM-bound, needs no weights) runs on a second HIP stream concurrently with the compute chain
        of step t (MFMA-bound), as the two branches of ONE fork/join hipGraph.  Buffers alternate by parity; the
        batches, samples and updates are exactly those of the sequential schedule."""
        e = self.engine
        if not getattr(self, "pipeline", True) or self._dropout_rate() > 0:
            self._parity = 0
            batch_dev = self.ids_buffer(n)[0][:n]
            labels_dev = e.ws_mat(("labels", 0), n, self.num_classes)

            def stage():   # batch selection + label gather ride along with the fused sampler launch
                self._pending_stage = (self._order, self._cursor, self.label_table, labels_dev)

            return self._train_on_device(batch_dev, labels_dev, n, fetch, prologue=stage, cursor=self._cursor,
                                         key="dtrain")
        fused = self.grad_hook is None or self._dp_in_graph()
        data = self._data_fn(n)
        if self._primed != n:                       # fill the pipeline: data chain of the first step
            self._drop_prefetched()
            self._pipe_parity = 0
            data(0)
            e.sync()
            self._primed = n
        self._pipelined_steps(n, 1, data, fused)
        return self._fetch(n) if fetch else None

    def _data_fn(self, n):
        e = self.engine

        def data(parity):
            batch_dev = self.ids_buffer(n, parity=parity)[0][:n]
            labels_dev = e.ws_mat(("labels", parity), n, self.num_classes)
            pre = self._data_phase(batch_dev, n, parity, stage=(self._order, self._cursor, self.label_table, labels_dev))
            e.advance(clock=1, cursor=self._cursor, cursor_delta=n)
            self._prefetched[(n, parity)] = (batch_dev, labels_dev, pre)   # static views of persistent buffers
        return data

    def _pipelined_steps(self, n, k, data, fused):
        """k consecutive pipelined steps as ONE hipGraph launch (k even, or 1).

        Single stream: step t first samples step t+1 (one small launch, or riding in an optimizer launch), then its
        launches carry step t+1's gather+mean waves along (horizontal fusion): no cross-stream dependency.  (A second-stream
        fork/join pipeline and a forked gather branch beside the all-reduce were built and measured in rounds 2-3 -- 152-162 us
        against 118, 149.9 against 147.9 -- and removed in round 4: benchmarks/variants/README.md.)"""
        e = self.engine
        p0 = self._pipe_parity
        mode = "fused"
        in_graph = self._dp_in_graph()          # `fused` is then True as well: the whole DP step is one graph
        local_adam = self.grad_hook is None

        def compute(p, side_jobs=None, epilogue=None):
            batch_dev, labels_dev, pre = self._prefetched[(n, p)]
            self._parity = p
            # the next step's gather is split between this step's two big GEMM launches (layer-0 forward, grouped
            # weight gradient): both are latency-bound, so the HBM-bound gather waves back-fill their idle slots
            if side_jobs and self.cogather_tail > 0 and self._tail_ok():
                # the fused tail launch keeps only n/16 CUs busy: the rest of the chip streams a share of the gather
                f_fwd, f_tail = self.rider_shares(side_jobs, self.dims[1] if self.aggregator_type == "gcn" else 2 * self.dims[1])
                fwd_jobs, rest = ops.split_gather_jobs(side_jobs, f_fwd)
                tail_jobs, wgrad_jobs = ops.split_gather_jobs(rest, min(1.0, f_tail / max(1e-6, 1.0 - f_fwd)))
            else:
                fwd_jobs, wgrad_jobs = ops.split_gather_jobs(side_jobs, self.cogather_split)
                tail_jobs = []
            self._forward(batch_dev, labels_dev, n, train=True, prefetched=pre, side_jobs=fwd_jobs, epilogue=epilogue,
                          tail_jobs=tail_jobs)
            self._backward(n, fuse_adam=local_adam, wgrad_jobs=wgrad_jobs, epilogue=epilogue)
            if in_graph:
                self.grad_hook(self)          # ncclAllReduce on the engine stream, recorded in the graph
                self._optimize(advanced=True)

        def sample_into(parity):
            batch = self.ids_buffer(n, parity=parity)[0][:n]
            labels = e.ws_mat(("labels", parity), n, self.num_classes)
            samples, support = self._sample_phase(batch, n, parity, stage=(self._order, self._cursor, self.label_table, labels))
            return batch, labels, samples, support

        # sampler-in-optimizer-launch: the device counters are advanced BEFORE the optimizer launch -- by the fused tail
        # launch, or by the early epilogue of _backward (this pipeline only runs without dropout)
        per_root = 1
        for f in self.num_samples[:0:-1]:
            per_root *= f
        ride = (self.sampler_rides and mode == "fused" and (local_adam or in_graph) and fused and k > 1
                and (self._tail_ok() or self._dropout_rate() == 0) and self._fanout_fusable() and per_root <= 512)

        def body():
            p = p0
            staged = None
            for j in range(k):
                if True:
                    q = 1 - p
                    if staged is None:
                        staged = sample_into(q)                    # standalone sampler launch (first step of a graph)
                    batch_q, labels_q, samples, support = staged
                    self_all, neighs = self._layer0_inputs(samples, support, n)
                    means_q, jobs = self.aggregators[0].prefetch_jobs(self_all, neighs, tag=q)
                    staged = None
                    if ride and j + 1 < k:
                        # the sampler of the step AFTER the next one rides in this step's optimizer launch: by then
                        # this step's own id / label buffers (parity p) are free, and the tail launch has already
                        # advanced the sampler clock and the epoch cursor -- the draws are those of the standalone launch
                        e._defer_sampler = True
                        try:
                            staged = sample_into(p)
                        finally:
                            e._defer_sampler = False
                        if e._deferred_sampler is None:
                            raise ops._lib.GraphsageAmdError("sampler did not take the one-launch fan-out path")
                    self._prefetched[(n, q)] = (batch_q, labels_q, (samples, support, means_q))
                    # (fused-tail models: that sampler leaves with the weight-gradient launch -- Engine.launch_wgrads -- and
                    #  the tail launch copies this step's ids for the weight gradients, whose id buffer the sampler refills)
                    e._sampler_to_wgrad = bool(self.sampler_in_wgrad and e._deferred_sampler is not None and self._tail_ok()
                                               and sum(jb.n * jb.s * jb.d * 4 for jb in jobs) <= self.sampler_in_wgrad_max_bytes)
                    try:
                        compute(p, side_jobs=jobs, epilogue=dict(step=1 if fused else 0, clock=1, cursor=self._cursor,
                                                                 cursor_delta=n))
                    finally:
                        if e._sampler_to_wgrad:
                            self._wgrad_sampler_seen = bool(e.last_wgrad_sampler)   # (tests: did the sampler leave with that launch?)
                        e._sampler_to_wgrad = False
                    if e._deferred_sampler is not None:
                        raise ops._lib.GraphsageAmdError("deferred sampler was not consumed by the optimizer launch")
                p = 1 - p

        key = ("ptrain" if local_adam else ("ptrain_dp" if in_graph else "ptrain_fb"), mode, n, k, p0, self._adj_version())
        self._run(key, body)
        if not fused:
            assert k == 1
            self.grad_hook(self)              # RCCL all-reduce of engine.grads (ordered by stream events)
            self._run(("opt",), self._optimize)
        if k % 2 == 1:
            self._pipe_parity = 1 - p0

    def train_steps_device(self, n, steps, steps_per_launch=8):
        """`steps` training steps on the device-resident epoch; on a single GPU `steps_per_launch` consecutive steps
        are replayed per hipGraph launch (amortises the launch gap; the schedule and results are unchanged)."""
        fused = self.grad_hook is None or self._dp_in_graph()
        k = steps_per_launch - (steps_per_launch % 2)
        if not (getattr(self, "pipeline", True) and self._dropout_rate() == 0 and fused and self.use_graphs and k >= 2):
            for _ in range(steps):
                self.train_step_device(n)
            return
        done = 0
        data = self._data_fn(n)
        while done < steps:
            # multi-step graphs always start at buffer parity 0 (one captured graph per length); single steps realign the
            # parity.  A shorter tail (the drivers replay print_every - 1 steps between two printed iterations) takes the
            # largest even length that fits, so it still is one launch with the sampler riding in the optimizer launches.
            rem = steps - done
            kk = min(k, rem - (rem % 2))
            if self._primed == n and self._pipe_parity == 0 and kk >= 2:
                self._pipelined_steps(n, kk, data, fused)
                done += kk
            else:
                self.train_step_device(n)
                done += 1
        self._check_exchange()

    def predict(self):
        """sigmoid / softmax of the logits (supervised_models.py:122-126); filled by the last step."""
        return self.preds
