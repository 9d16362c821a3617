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
import os
from collections import namedtuple

import numpy as np
import torch

from . import ops
from .aggregators import GCNAggregator, MaxPoolingAggregator, MeanAggregator, MeanPoolingAggregator
from .engine import get_engine
from .inits import glorot
from .layers import Rows, identity
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
    """Base implementation of unsupervised GraphSAGE (graphsage/models.py:187-405): three sample+aggregate passes
    (batch1, batch2, 20 degree^0.75 negatives) sharing the aggregators, l2-normalise, skip-gram cross-entropy
    (prediction.py:102-110), loss/batch_size, clip +-5, Adam, MRR.  Here the three passes are ONE pass over the
    concatenated roots [batch1 | batch2 | negatives] (rows are independent and the aggregators are shared, so it is
    the same computation).  `FLAGS.learning_rate / weight_decay / neg_sample_size` are explicit keyword arguments.

        loss, ranks, aff_all, mrr, outputs1 = model.train_step(feed_dict)   # unsupervised_train.py:273-274
        loss, ranks, mrr = model.eval_step(feed_dict)                        # :69-71
    The supervised subclass is in supervised_models.py."""

    def __init__(self, placeholders, features, adj, degrees, layer_infos, concat=True, aggregator_type="mean",
                 model_size="small", identity_dim=0, learning_rate=0.00001, weight_decay=0.0, neg_sample_size=20,
                 world_size=1, rank=0, _defer_build=False, **kwargs):
        allowed_kwargs = {'name', 'logging', 'model_size'}
        for kwarg in kwargs.keys():
            assert kwarg in allowed_kwargs, 'Invalid keyword argument: ' + kwarg
        self.name = kwargs.get('name') or self.__class__.__name__.lower()
        self.engine = get_engine()
        self.aggregator_cls = _aggregator_cls(aggregator_type)
        self.aggregator_type = aggregator_type
        self.model_size = model_size
        self.adj_info = adj
        self.identity_dim = int(identity_dim)
        self._init_features(features, adj, self.identity_dim)
        self.degrees = degrees
        self.concat = concat
        self.dims = [self.features.d]        # == (0 if features is None else F) + identity_dim  (:244)
        self.dims.extend([layer_infos[i].output_dim for i in range(len(layer_infos))])
        self.placeholders = placeholders
        self.batch_size = placeholders.get("batch_size") if placeholders else None
        self.layer_infos = layer_infos
        self.aggregators = None
        self._tape = None
        self.learning_rate = float(learning_rate)
        self.weight_decay = float(weight_decay)
        self.neg_sample_size = int(neg_sample_size)
        self.world_size, self.rank = int(world_size), int(rank)
        self.engine.dropout_seed = 123 + 1000003 * self.rank      # every data-parallel rank draws its own masks
        self.row_offset = 0
        # shares of the prefetch gather carried by the step's launches (sweeps in DESIGN.md).  Two-launch form (layer-0
        # forward | weight gradients): 0.7 | 0.3 with the LDS-tiled kernels, 0.5 | 0.5 with the stream kernels.
        # Three-launch form of the supervised mean model (layer-0 forward | fused tail | weight gradients): 0.25 | 0.40 |
        # 0.35 since round 6 -- the four-wave tiled forward and weight-gradient workgroups leave room for two rider workgroups on
        # their own CUs (profiles/r06_tiled3_wgrad_ab.txt: headline 97.7 us/step; 0.15 | 0.5 | 0.35 with the stream kernels, where
        # the tail -- 32 busy CUs -- was the best host).
        # Round 6, two main workgroups per group in the fused tail (training launches over 256 input columns, gs_tail.hip): the
        # tail's own chain is 19 us instead of 26, and what it hosts for free is a fixed AMOUNT of gather (~64 MB: 64 + 128 idle
        # CUs for 19 / 11 us), not a fixed share -- rider_shares() sets (forward | tail | weight gradients) from the step's gather
        # bytes: 0.40 | 0.20 | 0.40 for the Reddit step (321 MB: 98.3 -> 94.0 us/step), 0.125 | 0.75 | 0.125 for RMAT
        # (79 MB: 67.4 -> 59.9); profiles/r06_tail_halves_ab.txt.  GS_COGATHER_TAIL / GS_COGATHER_SPLIT3 pin fixed shares.
        # OPT-IN (GS_SAMPLER_IN_WGRAD=1): the sampler of the step after the next rides in the weight-gradient launch instead of
        # the optimizer launch (supervised fused-tail models on the tiled kernels; one root per WAVE of a rider workgroup; the tail
        # launch makes the private id copy the weight gradients then read).  The optimizer launch waits 8 us for the sampler's
        # chain against 5.5 us of its own; same-call A/B: Reddit 95.0 -> 92.0 us/step, GCN 96.1 -> 95.8, RMAT 59.4 -> 57.8.  Every
        # test passes with it (bit-identical ids, steps and parameters against the optimizer-launch form) -- and in graphs of
        # >= 8 steps ONE training run in ~10^5 steps differs from the others (benchmarks/determinism.sh: 2 of 220 200-step runs;
        # benchmarks/race_hunt.py: a z-helper workgroup of the tail read a stale 128-byte line of h0; never in 1.9 M steps with the
        # sampler in the optimizer launch, never in 2-step or 4-step graphs): until that is understood the default stays off
        # (profiles/r06_determinism.txt).
        self.sampler_in_wgrad = os.environ.get("GS_SAMPLER_IN_WGRAD", "0") == "1"
        self.sampler_in_wgrad_max_bytes = 1e15
        self._wgrad_sampler_seen = None
        self.tail_halves = os.environ.get("GS_TAIL_HALVES", "1") != "0"
        self.cogather_auto = ("GS_COGATHER_TAIL" not in os.environ and "GS_COGATHER_SPLIT3" not in os.environ)
        self.tail_free_bytes = 64e6
        self.cogather_split = float(os.environ.get("GS_COGATHER_SPLIT", 0.5 if self.engine.stream_gemm else 0.7))
        tiled = self.engine.stream_gemm and self.engine.tiled3_fwd and self.engine.tiled3_wgrad
        self.cogather_split3 = float(os.environ.get("GS_COGATHER_SPLIT3", 0.25 if tiled else 0.15))
        self.cogather_tail = float(os.environ.get("GS_COGATHER_TAIL", (0.40 if tiled else 0.5) if self.engine.stream_gemm else 0.0))
        # unsupervised three-launch form (forward | fused link-prediction tail | weight gradients)
        self.cogather_lp_fwd = float(os.environ.get("GS_COGATHER_LP_FWD", 0.20))      # 0.30 with the stream forward (round 5); the tiled forward holds a CU per workgroup: 176.2 vs 179.5 us/step
        self.cogather_lp_tail = float(os.environ.get("GS_COGATHER_LP_TAIL", 0.25))
        self.cogather_lp_neg = float(os.environ.get("GS_COGATHER_LP_NEG", 0.10))
        self.cogather_z = 0.04            # measured: 0 | 0.04 | 0.08 | 0.12 -> 201.5 | 199.1 | 202.9 | 209.4 us
        # DIAGNOSTIC (one test pins it bit-identical): the fused tail as two launches (z helpers | row-group workgroups) --
        # no dependency between workgroups of a launch, the safe form under tools that serialise workgroups; +11 us per step
        self.tail_split = os.environ.get("GS_TAIL_SPLIT", "0") == "1"
        self.cogather_tail_z = 0.35           # part of the tail's gather share the z launch carries in that form
        # (Removed in round 4 after losing their measurements -- sources and numbers in benchmarks/variants/README.md: a gather
        #  share in the optimizer launch, one in the last layer's backward launch, a forked gather branch beside the in-graph
        #  all-reduce, the second-stream pipeline, the weight-stationary form of the layer-0 forward.)
        # inside a multi-step graph the sampler of step t+2 rides in step t's optimizer launch (see _pipelined_steps)
        self.sampler_rides = True         # (diagnostic: False = every sampler launch stands alone; one test)
        # the fused tail launches (supervised: gs_sage_tail_fwd_bwd; unsupervised: gs_linkpred_tail); 0 = per-operator schedule
        self.fuse_tail = os.environ.get("GS_FUSE_TAIL", "1") != "0"
        self._graphs, self._graph_outputs, self._warm = {}, {}, set()
        self.use_graphs = True
        self.grad_hook = None
        if self.identity_dim == 0:
            # next-step gather co-scheduled inside this step's launches (one stream); False = sequential schedule
            self.pipeline = os.environ.get("GS_PIPELINE", "1") != "0"
        self._primed = None
        self._prefetched = {}
        self._pending_stage = None
        if not _defer_build:
            self.inputs1 = placeholders["batch1"]
            self.inputs2 = placeholders["batch2"]
            self.build()

    # ------------------------------------------------------------------------------ identity features (:229-240)
    def _init_features(self, features, adj, identity_dim):
        """self.features = concat([node_embeddings, features], axis=1) (models.py:229-240).  The trainable
        `node_embeddings` [N+1, identity_dim] (tf.get_variable default initializer = glorot_uniform) lives in the flat
        parameter buffer; the concatenation is kept materialised as ONE table so that a sampled id still costs one
        row fetch, and its leading columns are refreshed after every optimizer step (Engine.post_update_hooks)."""
        e = self.engine
        self.embeds = None
        if identity_dim > 0:
            n_rows = adj.n_nodes + 1                                   # adj.get_shape()[0]
            self.embeds = e.add_variable("node_embeddings", glorot((n_rows, identity_dim)), decay=False, scatter=True)
        if features is None:
            if identity_dim == 0:
                raise Exception("Must have a positive value for identity feature dimension if no input features given.")
            self.features = Mat.zeros(n_rows, identity_dim, e.device, ld_multiple=32)
        elif self.embeds is None:
            self.features = device_features(features, e)
        else:
            fixed = np.asarray(features.numpy() if isinstance(features, Mat) else features, dtype=np.float32)
            if fixed.shape[0] != n_rows:
                raise ops._lib.GraphsageAmdError("features must have N+1 = %d rows (got %d)" % (n_rows, fixed.shape[0]))
            self.features = Mat.zeros(n_rows, identity_dim + fixed.shape[1], e.device, ld_multiple=32)
            self.features.buf[:, identity_dim: identity_dim + fixed.shape[1]].copy_(torch.from_numpy(fixed))
            torch.cuda.synchronize()
        if self.embeds is not None:
            e.post_update_hooks.append(self._refresh_embeds)
            # the table's leading columns are rewritten behind every optimizer launch: no cut-once copies of it (Engine.table16_of)
            e.mutable_tables.add(self.features.buf.data_ptr())
            # the prefetch pipeline gathers step t+1's rows before step t's update: not valid for a trainable table
            self.pipeline = False

    def _refresh_embeds(self):
        ops.copy_cols(self.embeds.value, self.features, self.embeds.rows, self.identity_dim, stream=self.engine.stream)

    def _embed_sink(self):
        return (self.embeds, self.identity_dim) if self.embeds is not None else None

    # ------------------------------------------------------------------------------ unsupervised build (:332-391)
    def build(self):
        e = self.engine
        self.num_samples = [layer_info.num_samples for layer_info in self.layer_infos]
        self.aggregators = self.make_aggregators(self.dims, self.num_samples, self.concat, self.model_size)
        from .prediction import BipartiteEdgePredLayer
        dim_mult = 2 if self.concat else 1
        self.link_pred_layer = BipartiteEdgePredLayer(dim_mult * self.dims[-1], dim_mult * self.dims[-1], self.placeholders,
                                                      act="sigmoid", bilinear_weights=False, name='edge_predict')
        e.finalize()
        if self.embeds is not None:
            self._refresh_embeds()
        self.loss_dev = torch.zeros(1, dtype=torch.float32, device=e.device)
        self.mrr_dev = torch.zeros(1, dtype=torch.float32, device=e.device)
        # fixed unigram distribution ~ degree^0.75 of tf.nn.fixed_unigram_candidate_sampler (:336-343) as a uint32 CDF
        w = np.power(np.asarray(self.degrees, dtype=np.float64), 0.75)
        if w.sum() <= 0:
            w = np.ones_like(w)
        c = np.cumsum(w) / w.sum()
        cdf = np.minimum(np.floor(c * 4294967296.0), 4294967295.0).astype(np.uint32)
        cdf[-1] = np.uint32(4294967295)
        self._neg_cdf = torch.from_numpy(cdf.view(np.int32).copy()).to(e.device)   # raw bits; the kernel reads uint32
        self._n_cdf = int(cdf.shape[0])
        # guide table of the inverse-cdf search: guide[b] = first index whose cdf exceeds b << (32 - bits); a draw r then
        # searches [guide[r >> s], guide[(r >> s) + 1]] only (same result, ~6 dependent loads instead of 18)
        self._guide_bits = 18          # 1 MB table; ~1 node per bucket: the search is 1-2 dependent loads
        thr = (np.arange((1 << self._guide_bits) + 1, dtype=np.uint64) << np.uint64(32 - self._guide_bits))
        guide = np.searchsorted(cdf.astype(np.uint64), thr, side="right")
        guide = np.minimum(guide, len(cdf) - 1).astype(np.int32)
        self._neg_guide = torch.from_numpy(guide).to(e.device)
        self.neg_seed = 123
        torch.cuda.synchronize()

    _OUT_ATTRS = ("samples1", "outputs_all", "outputs1", "agg_out", "_loss_rows", "_rr_rows", "aff_all", "_d_agg_out",
                  "_loss_accumulate", "_tape", "_lp_tail_used", "_tail_h0", "_tail_means", "_tail_dh0", "_lp_sync",
                  "_lp_sync_shape", "_epilogue_folded")

    def rider_shares(self, jobs, tail_d_in):
        """(forward share, tail share) of the next step's gather for the three-launch form (the weight gradients carry the
        rest).  Fixed shares unless the fused tail runs two main workgroups per group (tail_d_in == 256); then the tail takes
        what it hosts for free (tail_free_bytes) and the two contraction launches halve the rest."""
        tiled = self.engine.stream_gemm and self.engine.tiled3_fwd and self.engine.tiled3_wgrad
        if not (self.cogather_auto and self.tail_halves and tiled and tail_d_in == 256 and not self.tail_split):
            return self.cogather_split3, self.cogather_tail
        total = float(sum(j.n * j.s * j.d * 4 for j in jobs))
        f_tail = min(0.75, max(0.10, self.tail_free_bytes / max(total, 1.0)))
        return 0.5 * (1.0 - f_tail), f_tail

    def _roots(self, B, parity=None):
        """[batch1 (B) | batch2 (B) | negatives] = the head of the contiguous id buffer."""
        n_roots = 2 * B + self.neg_sample_size
        return self.ids_buffer(n_roots, parity=parity)[0][:n_roots], n_roots

    def _stage_negatives(self, roots, B, pairs=None, cursor=None):
        e = self.engine
        ops.call("gs_unsup_stage", ops.ptr(pairs), pairs.shape[0] if pairs is not None else 0, ops.ptr(cursor), B,
                 ops.ptr(self._neg_cdf), self._n_cdf, self.neg_sample_size, self.neg_seed, ops.ptr(e.sample_clock_dev),
                 int(getattr(self, "row_offset", 0)), ops.ptr(roots), e.stream)

    def inject_negatives(self, neg):
        """Parity tests: the negatives of the next host-fed step (the reference draws them from TF's candidate sampler,
        models.py:336-343; like the sampler's permutations they are injected so both sides see the same ids)."""
        self._injected_neg = None if neg is None else np.ascontiguousarray(neg, dtype=np.int32)
        self.use_graphs = False if neg is not None else self.use_graphs

    def _stage_negatives_or_injected(self, roots, B):
        neg = getattr(self, "_injected_neg", None)
        if neg is None:
            return self._stage_negatives(roots, B)
        assert neg.shape[0] == self.neg_sample_size
        self._injected_neg = None
        self.engine.sync()
        roots[2 * B: 2 * B + self.neg_sample_size].copy_(torch.from_numpy(neg))
        torch.cuda.current_stream().synchronize()

    def _forward_unsup(self, roots, B, n_roots, train, prefetched=None, side_jobs=None, epilogue=None, z_jobs=None, tail_jobs=None):
        """_build (:347-370) + _loss (:385-391) + _accuracy (:393-405) and, when training, the gradient of the
        link-prediction head w.r.t. the normalised embeddings."""
        e = self.engine
        self.reset_tapes()
        if prefetched is None:
            prefetched = self._data_phase(roots, n_roots, getattr(self, "_parity", 0))
        samples1, support_sizes1, means0 = prefetched
        contiguous = all(b.data_ptr() == a.data_ptr() + 4 * a.numel() for a, b in zip(samples1[:-1], samples1[1:]))
        self._lp_tail_used = bool(contiguous and self._lp_tail_ok() and n_roots == 2 * B + self.neg_sample_size)
        out, _ = self.aggregate(samples1, [self.features], self.dims, self.num_samples, support_sizes1, batch_size=n_roots,
                                aggregators=self.aggregators, concat=self.concat, model_size=self.model_size,
                                layer0_means=means0, layer0_side_jobs=side_jobs, last_layer_side_jobs=z_jobs,
                                _stop_after_layer=0 if self._lp_tail_used else None)
        self.samples1 = samples1
        self._loss_rows = e.ws_f32("loss_rows", B)
        self._rr_rows = e.ws_f32("rr_rows", B)
        self.aff_all = e.ws_mat("aff_all", B, self.neg_sample_size + 1)
        if self._lp_tail_used:
            if self._tape[0][0] != "batched":
                raise ops._lib.GraphsageAmdError("fused tail needs the contiguous id buffer (model.sample on ids_buffer)")
            tj, nj = tail_jobs, None
            if tail_jobs and self.cogather_lp_neg > 0 and self.cogather_lp_tail > 0:
                # the second launch (nine workgroups) carries its own share of the gather
                tj, nj = ops.split_gather_jobs(tail_jobs, self.cogather_lp_tail / (self.cogather_lp_tail + self.cogather_lp_neg))
            self._forward_lp_tail(B, n_roots, train, epilogue, tj, nj)
            return
        self.agg_out = out
        d = out.d
        self.outputs_all = e.ws_mat("outputs_all", n_roots, d)
        self.outputs1 = self.outputs_all.rows_slice(0, B)
        # l2_normalize (:368-370) + link-prediction loss / MRR ranks (:385-405) + their gradient carried back through the
        # normalisation: ONE launch (+ a 20-workgroup one for the negatives' rows); d_agg_out = dLoss/d(aggregator output)
        self._d_agg_out = e.ws_mat("d_agg_out", n_roots, d)
        # `epilogue` (the step's device-counter increments): without dropout the loss / mrr means and the counters ride in
        # the head's second launch (with dropout the clock must not move before the backward pass: separate launch later)
        fold = epilogue is not None and self._dropout_rate() == 0.0
        self._epilogue_folded = fold
        epi = None
        if fold:
            epi = (self.loss_dev, False, self.mrr_dev,
                   [(e.step_dev, epilogue.get("step", 0)), (e.sample_clock_dev, epilogue.get("clock", 0)),
                    (epilogue.get("cursor"), epilogue.get("cursor_delta", 0))])
        self.link_pred_layer.loss_and_grads_fused(out, self.outputs_all, B, self.neg_sample_size, 1.0 / B, self._loss_rows,
                                                  self._rr_rows, self.aff_all, self._d_agg_out, epilogue=epi)
        # loss = (sum_vars wd*l2_loss + xent) / batch_size  (:386-390, :378); the xent mean (and the mrr, :404) are formed
        # by the epilogue (folded: the weight-decay terms are added behind it; separate launch: it adds to them)
        self._loss_accumulate = False
        if self.weight_decay != 0.0:
            first = not fold
            for a in self.aggregators:
                for v in a.vars.values():
                    ops.call("gs_sumsq_scaled", v.value.ptr, v.size, 0.5 * self.weight_decay / B, self.loss_dev.data_ptr(),
                             0 if first else 1, e.stream)
                    first = False
            self._loss_accumulate = not fold

    def _lp_tail_ok(self):
        """The fused layer-1 + link-prediction launches (gs_linkpred_tail / gs_linkpred_tail_neg) apply to the two-layer mean
        model with concat, no aggregator bias, no dropout, no trainable identity features, at shapes the kernel supports."""
        if not getattr(self, "fuse_tail", True) or len(self.layer_infos) != 2 or self.aggregator_type != "mean":
            return False
        a1 = self.aggregators[1]
        return (self.concat and not a1.bias and self._dropout_rate() == 0 and self.embeds is None
                and self.num_samples[-1] <= 11
                and ops.linkpred_tail_supported(2 * self.dims[1], self.dims[2], self.neg_sample_size))

    def _forward_lp_tail(self, B, n_roots, train, epilogue, tail_jobs, neg_jobs=None):
        """Layer 1 + l2_normalize + link-prediction loss / MRR (+ every input gradient down to layer 0's pre-activations when
        training) as the two launches of gs_unsup_tail.hip; gather jobs of the next step ride in the first one."""
        e = self.engine
        h0 = self._tape[0][4]                       # [n + n*s, 2*dim_1]: layer-0 outputs of both hops
        a1 = self.aggregators[1]
        O = self.dims[2]
        Z = 2 * O
        s = self.num_samples[len(self.num_samples) - 1]
        nn = self.neg_sample_size
        self._tail_h0 = h0
        self._tail_means = e.ws_mat("tail_means", n_roots, h0.d)
        self.agg_out = e.ws_mat("tail_z", n_roots, Z)
        self.outputs_all = e.ws_mat("outputs_all", n_roots, Z)
        self.outputs1 = self.outputs_all.rows_slice(0, B)
        self._d_agg_out = e.ws_mat("d_agg_out", n_roots, Z)
        self._tail_dh0 = e.ws_mat((self.name, "d_hidden", 0), h0.rows, h0.d)
        slabs = e.ws_f32(("lp_neg_slabs", B, nn, Z), ((B + 7) // 8) * nn * Z)
        self._lp_sync = e.ws_i32(("lp_tail_sync", self.name, B, nn), ops.lp_tail_sync_words(B, nn))
        self._lp_sync_shape = (B, nn)
        desc = ops.linkpred_tail_desc(h0, B, nn, s, a1.vars['self_weights'].value, a1.vars['neigh_weights'].value, O,
                                      self._tail_means, self.agg_out, self.outputs_all, self._loss_rows, self._rr_rows,
                                      self.aff_all, self.link_pred_layer.neg_sample_weights, 1.0 / B, self._lp_sync,
                                      dz=self._d_agg_out if train else None, d_h0=self._tail_dh0 if train else None,
                                      neg_slabs=slabs if train else None)
        ops.linkpred_tail(desc, jobs=tail_jobs, stream=e.stream)
        # launch 2 always follows (it commits the hand-over state); the step epilogue rides in it unless dropout needs the
        # clock untouched until the backward pass has run (this path runs without dropout: always folded when given)
        fold = epilogue is not None
        self._epilogue_folded = fold
        counters = []
        if fold:
            counters = [(e.step_dev, epilogue.get("step", 0)), (e.sample_clock_dev, epilogue.get("clock", 0)),
                        (epilogue.get("cursor"), epilogue.get("cursor_delta", 0))]
        ops.linkpred_tail_neg(desc, loss_out=self.loss_dev if fold else None, accumulate=False,
                              mrr_out=self.mrr_dev if fold else None, counters=counters, jobs=neg_jobs, stream=e.stream)
        self._loss_accumulate = False
        if self.weight_decay != 0.0:
            first = not fold
            for a in self.aggregators:
                for v in a.vars.values():
                    ops.call("gs_sumsq_scaled", v.value.ptr, v.size, 0.5 * self.weight_decay / B, self.loss_dev.data_ptr(),
                             0 if first else 1, e.stream)
                    first = False
            self._loss_accumulate = not fold

    def _backward_unsup(self, B, n_roots, fuse_adam, wgrad_jobs=None, epilogue=None):
        """Reverse schedule.  The epilogue (loss / mrr means + device counters) runs FIRST: the fan-out sampler of a later
        step may ride in this pass's optimizer launch and must see the advanced sampler clock and pair cursor; the
        optimizer then uses step_offset = 0 if the step counter has been advanced already."""
        e = self.engine
        # dropout masks are a function of the device clock and the backward pass regenerates them: with dropout on, the
        # clock must not move before the backward pass has run (no sampler rides there: the prefetch pipeline is off)
        folded = epilogue is not None and getattr(self, "_epilogue_folded", False)     # already done by the forward pass
        early = epilogue is not None and not folded and self._dropout_rate() == 0.0
        if early:
            self._epilogue_unsup(B, **epilogue)
        advanced = (early or folded) and bool(epilogue.get("step"))
        e.begin_backward()
        if getattr(self, "_lp_tail_used", False):
            # the fused tail produced every input gradient; queue the weight gradients it feeds (as SupervisedGraphsage._backward)
            a1 = self.aggregators[1]
            o = self.dims[2]
            h0 = self._tail_h0
            e.wgrad(a1.vars['self_weights'], h0.rows_slice(0, n_roots), None, self._d_agg_out, 0, n_roots)
            e.wgrad(a1.vars['neigh_weights'], self._tail_means, None, self._d_agg_out, o, n_roots)
            mode, agg0, rows, offsets, outs = self._tape[0]
            agg0.backward_hops(self._tail_dh0, True, embed_sink=None)
        else:
            self.aggregate_backward(self._d_agg_out)
        # every term of the loss is divided by batch_size (:378) -> so is the weight-decay gradient
        e.finish_backward(self.weight_decay / B, fuse_adam=fuse_adam, lr=self.learning_rate, clip=5.0, side_jobs=wgrad_jobs,
                          step_offset=0 if advanced else 1)
        if epilogue is not None and not early and not folded:
            self._epilogue_unsup(B, **epilogue)

    def _epilogue_unsup(self, B, **counters):
        self.engine.advance(loss_rows=self._loss_rows, n=B, loss_out=self.loss_dev, accumulate=self._loss_accumulate,
                            aux_rows=self._rr_rows, aux_out=self.mrr_dev, **counters)

    def _optimize(self):
        e = self.engine
        e.adam(self.learning_rate, clip=5.0, grad_scale=1.0 / self.world_size)
        e.advance(step=1)

    def _stage_feed_unsup(self, feed_dict):
        e = self.engine
        ph = self.placeholders
        self._parity = "h"         # host-fed batches own their buffers (see SupervisedGraphsage._stage_feed)
        self._pending_stage = None
        e.sync()
        b1 = np.ascontiguousarray(np.asarray(feed_dict[ph['batch1']]), dtype=np.int32)
        b2 = np.ascontiguousarray(np.asarray(feed_dict[ph['batch2']]), dtype=np.int32)
        B = int(b1.shape[0])
        assert b2.shape[0] == B and int(feed_dict.get(ph['batch_size'], B)) == B
        self._feed_dropout(feed_dict)
        roots, n_roots = self._roots(B, parity="h")
        roots[:B].copy_(torch.from_numpy(b1))
        roots[B:2 * B].copy_(torch.from_numpy(b2))
        torch.cuda.current_stream().synchronize()
        return roots, B, n_roots

    def _fetch_unsup(self, B, with_outputs=True):
        self.engine.sync()
        if getattr(self, "_lp_sync", None) is not None:
            err = ops.lp_tail_sync_error(self._lp_sync, *self._lp_sync_shape)
            if err:
                raise ops._lib.GraphsageAmdError(
                    "fused link-prediction tail: hand-over between its workgroups failed (flags %d: 1 = a main workgroup gave up "
                    "waiting for its helpers, 2 = unexpected arrival count); results since the last fetch are invalid -- set "
                    "model.fuse_tail = False to use the per-operator schedule" % err)
        if hasattr(self.grad_hook, "check"):
            self.grad_hook.check()
        loss = float(self.loss_dev.item())
        mrr = float(self.mrr_dev.item())
        aff = self.aff_all.numpy()
        ranks = (aff[:, :-1] >= aff[:, -1:]).sum(axis=1)
        outs = self.outputs1.numpy() if with_outputs else None
        return loss, ranks, aff, mrr, outs

    def train_step(self, feed_dict, fetch=True):
        """sess.run([merged, opt_op, loss, ranks, aff_all, mrr, outputs1], feed_dict)  (unsupervised_train.py:273-274)."""
        e = self.engine
        roots, B, n_roots = self._stage_feed_unsup(feed_dict)
        fused = self.grad_hook is None
        in_graph = self._dp_in_graph()

        def fwd_bwd():
            self._stage_negatives_or_injected(roots, B)
            epilogue = dict(step=1 if fused else 0, clock=1)
            self._forward_unsup(roots, B, n_roots, True, epilogue=epilogue)
            self._backward_unsup(B, n_roots, fuse_adam=fused, epilogue=epilogue)
            if in_graph:
                # backward | all-reduce (recorded in the graph) | clip + Adam, as _pipelined_steps_unsup
                self.grad_hook(self)
                self._optimize()

        # With a capturable hook the whole data-parallel step is this one function (one hipGraph): the exchange and the
        # optimizer run exactly once per step.
        self._run(("utrain" if fused else ("utrain_dp" if in_graph else "utrain_fb"), B, self._adj_version()), fwd_bwd)
        if not fused and not in_graph:
            self.grad_hook(self)
            self._run(("opt",), self._optimize)
        return self._fetch_unsup(B) if fetch else None

    def eval_step(self, feed_dict):
        """sess.run([loss, ranks, mrr], feed_dict)  (unsupervised_train.py:69-71); also used for the (n, n) embedding
        pairs of save_val_embeddings (:94-117) -- `.outputs1` holds the embeddings of batch1."""
        roots, B, n_roots = self._stage_feed_unsup(feed_dict)

        def fwd():
            self._stage_negatives_or_injected(roots, B)
            self._forward_unsup(roots, B, n_roots, False, epilogue=dict(clock=1))
            if not self._epilogue_folded:
                self._epilogue_unsup(B, clock=1)

        self._run(("ueval", B, self._adj_version()), fwd)
        loss, ranks, aff, mrr, outs = self._fetch_unsup(B)
        return loss, ranks, mrr, outs

    # ---- device-resident epoch: edge pairs live in HBM; steady state is one hipGraph replay per step with the next
    #      step's sampling + gather co-scheduled with this step's layer-0 contraction (see supervised_models.py)
    def attach_device_pairs(self, pairs):
        e = self.engine
        if self.embeds is not None:
            raise NotImplementedError("the prefetching device-epoch path reads next-step rows before this step's update; "
                                      "with identity_dim > 0 use train_step(feed_dict)")
        self._pairs = torch.from_numpy(np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)).to(e.device)
        self._cursor = torch.zeros(1, dtype=torch.int64, device=e.device)
        self._primed = None
        torch.cuda.synchronize()

    def set_epoch_pairs(self, pairs):
        self.engine.sync()
        self._pairs.copy_(torch.from_numpy(np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)))
        self._cursor.zero_()
        if self._primed is not None:
            self.engine.sample_clock_dev -= 1
        self._primed = None
        torch.cuda.synchronize()

    def train_step_device(self, B, fetch=False):
        self._pipelined_steps_unsup(B, 1)
        return self._fetch_unsup(B) if fetch else None

    def train_steps_device(self, B, steps, steps_per_launch=8):
        """`steps` steps on the device-resident pairs; on one GPU `steps_per_launch` consecutive steps are replayed per
        hipGraph launch (keeps the GPU fed when the host is slow or shared)."""
        k = steps_per_launch - (steps_per_launch % 2)
        if not ((self.grad_hook is None or self._dp_in_graph()) and self.use_graphs and k >= 2):
            for _ in range(steps):
                self._pipelined_steps_unsup(B, 1)
            return
        done = 0
        while done < steps:
            # multi-step graphs always start at buffer parity 0 (one captured graph per length); single steps realign the
            # parity; a shorter tail takes the largest even length that fits
            rem = steps - done
            kk = min(k, rem - (rem % 2))
            if self._primed == B and self._pipe_parity == 0 and kk >= 2:
                self._pipelined_steps_unsup(B, kk)
                done += kk
            else:
                self._pipelined_steps_unsup(B, 1)
                done += 1
        self._check_exchange()

    def _check_exchange(self):
        """A gradient exchange with bounded device-side waits (PeerPushAllReduce) reports a tripped wait through an error
        word: read it once per train_steps_device call, BEFORE further optimizer steps are issued on un-reduced gradients
        (costs one stream synchronisation per call; hooks without check() skip it)."""
        if hasattr(self.grad_hook, "check"):
            self.engine.sync()
            self.grad_hook.check()

    def _dp_in_graph(self):
        """Data-parallel AND the all-reduce hook can be recorded inside the step's hipGraph (NativeAllReduce)."""
        return self.grad_hook is not None and getattr(self.grad_hook, "capturable", False)

    def measure_dp_allreduce(self, log=None):
        """Duration of the gradient all-reduce as a stand-alone call (HIP events around eager calls of the hook on the
        gradient buffer -- every rank calls this, it is a collective): what the in-graph data-parallel step exposes per
        step, echoed by bench.py (dp_schedule)."""
        e = self.engine
        if not self._dp_in_graph():
            return None
        evs = [(ops.Event(), ops.Event()) for _ in range(12)]
        for a, b in evs:
            a.record(e.stream)
            self.grad_hook(self)
            b.record(e.stream)
        e.sync()
        ar_us = float(np.median([a.elapsed_ms(b) for a, b in evs[2:]])) * 1e3
        e.grads.zero_()
        torch.cuda.synchronize()
        if log:
            log("data-parallel schedule: all-reduce %.1f us stand-alone, recorded in the step graph behind the backward pass" % ar_us)
        return {"allreduce_us_standalone": ar_us}

    def _pipelined_steps_unsup(self, B, k):
        e = self.engine
        local_adam = self.grad_hook is None
        in_graph = self._dp_in_graph()
        fused = local_adam or in_graph

        fusable = self._fanout_fusable()

        def sample_next(parity):
            roots, n_roots = self._roots(B, parity=parity)
            self._parity = parity
            if fusable:
                # the edge-pair batch and the negatives are staged by the fan-out sampler launch itself (one launch, and
                # one that can ride in the optimizer launch)
                stage = ("unsup", self._pairs, self._cursor, B, self._neg_cdf, self._neg_guide, self._guide_bits,
                         self.neg_sample_size, self.neg_seed)
                samples, support = self._sample_phase(roots, n_roots, parity, stage=stage)
            else:
                self._stage_negatives(roots, B, pairs=self._pairs, cursor=self._cursor)
                samples, support = self._sample_phase(roots, n_roots, parity)
            return roots, n_roots, samples, support

        if self._primed != B:
            self._pipe_parity = 0
            roots, n_roots, samples, support = sample_next(0)
            self_all, neighs = self._layer0_inputs(samples, support, n_roots)
            means0 = self.aggregators[0].prefetch(self_all, neighs, tag=0) if self_all is not None else None
            e.advance(clock=1, cursor=self._cursor, cursor_delta=B)
            self._prefetched[(B, 0)] = (roots, n_roots, (samples, support, means0))
            e.sync()
            self._primed = B
        p0 = self._pipe_parity
        per_root = 1
        for f in self.num_samples[:0:-1]:
            per_root *= f
        # inside a multi-step graph the sampler of the step after next rides in this step's optimizer launch
        ride = self.sampler_rides and fused and k > 1 and fusable and per_root <= 512

        def body():
            p = p0
            staged = None
            for j in range(k):
                q = 1 - p
                if staged is None:
                    staged = sample_next(q)                    # standalone sampler launch (first step of a graph)
                roots_q, n_roots_q, samples, support = staged
                self_all, neighs = self._layer0_inputs(samples, support, n_roots_q)
                means_q, jobs = self.aggregators[0].prefetch_jobs(self_all, neighs, tag=q)
                staged = None
                if ride and j + 1 < k:
                    # by the time the optimizer launch runs, this step's own id buffers (parity p) are free and the epilogue
                    # has advanced the sampler clock and the pair cursor: the draws are those of the standalone launch
                    e._defer_sampler = True
                    try:
                        staged = sample_next(p)
                    finally:
                        e._defer_sampler = False
                    if e._deferred_sampler is None:
                        raise ops._lib.GraphsageAmdError("sampler did not take the one-launch fan-out path")
                self._prefetched[(B, q)] = (roots_q, n_roots_q, (samples, support, means_q))
                roots, n_roots, pre = self._prefetched[(B, p)]
                self._parity = p
                tail_jobs = []
                if self._lp_tail_ok() and self.cogather_lp_tail > 0:
                    # three-launch form (layer-0 forward | fused link-prediction tail | weight gradients): the tail is long and
                    # thin (66 main workgroups), the rest of the chip streams its share of the gather at the full rate
                    f_fwd, f_tail = self.cogather_lp_fwd, self.cogather_lp_tail + self.cogather_lp_neg
                    fwd_jobs, rest = ops.split_gather_jobs(jobs, f_fwd)
                    tail_jobs, wgrad_jobs = ops.split_gather_jobs(rest, min(1.0, f_tail / max(1e-6, 1.0 - f_fwd)))
                else:
                    fwd_jobs, wgrad_jobs = ops.split_gather_jobs(jobs, self.cogather_split)
                z_jobs = []
                if not tail_jobs and self.cogather_z > 0 and len(self.num_samples) > 1 and not self._lp_tail_ok():
                    # the last layer's lean launch (gs_sage_tail_z) leaves most of the chip idle: a share rides there at
                    # the full HBM rate
                    wgrad_jobs, z_jobs = ops.split_gather_jobs(wgrad_jobs, max(0.0, 1.0 - self.cogather_z / max(1e-6, 1.0 - self.cogather_split)))
                epilogue = dict(step=1 if local_adam else 0, clock=1, cursor=self._cursor, cursor_delta=B)
                self._forward_unsup(roots, B, n_roots, True, prefetched=pre, side_jobs=fwd_jobs, epilogue=epilogue, z_jobs=z_jobs,
                                    tail_jobs=tail_jobs)
                self._backward_unsup(B, n_roots, fuse_adam=local_adam, wgrad_jobs=wgrad_jobs, epilogue=epilogue)
                if e._deferred_sampler is not None:
                    raise ops._lib.GraphsageAmdError("deferred sampler was not consumed by the optimizer launch")
                if in_graph:
                    # backward | ncclAllReduce (recorded in the graph) | clip + Adam
                    self.grad_hook(self)
                    self._optimize()
                p = q

        self._run(("updtrain" if local_adam else ("updtrain_dp" if in_graph else "updtrain_fb"), B, k, p0,
                   self._adj_version()), body)
        if not fused:
            assert k == 1
            self.grad_hook(self)
            self._run(("opt",), self._optimize)
        if k % 2 == 1:
            self._pipe_parity = 1 - p0

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

    def _fanout_fusable(self, layer_infos=None):
        """Will model.sample() on the contiguous id buffer take the one-launch fan-out path?"""
        from .neigh_samplers import CSRAdjacency
        layer_infos = layer_infos or self.layer_infos
        sampler0 = layer_infos[0].neigh_sampler
        return (len(layer_infos) <= 3 and all(li.neigh_sampler is sampler0 for li in layer_infos)
                and isinstance(sampler0.adj_info.current, CSRAdjacency) and getattr(self, "fuse_sampler", True))

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
            if sampler.root_segments is not None:
                rows_per_root = samples[k].numel() // batch_size
                sampler.call_segments = tuple(b * rows_per_root for b in sampler.root_segments)
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
                  layer0_side_jobs=None, _stop_after_layer=None, last_layer_side_jobs=None):
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
                jobs = layer0_side_jobs if layer == 0 else (last_layer_side_jobs if layer == K - 1 else None)
                h_all = aggregator.call_hops(self_all, neighs, means=means, side_jobs=jobs)   # all hops, one launch
                outs = [h_all.rows_slice(offsets[h], offsets[h + 1]) for h in range(n_hops)]
                tape.append(("batched", aggregator, rows, offsets, h_all))
            else:                                                           # non-adjacent inputs: hop by hop (:326)
                outs = [aggregator((hidden[hop], neighs[hop])) for hop in range(n_hops)]
                tape.append(("per_hop", aggregator, rows, offsets, outs))
            hidden = [Rows(o, None, requires_grad=True) for o in outs]
            if _stop_after_layer is not None and layer == _stop_after_layer:
                break          # the remaining layers run inside a fused launch (SupervisedGraphsage._forward)
        self._tape = tape
        return hidden[0].src, aggregators

    def aggregate_backward(self, d_out):
        """Reverse schedule of `aggregate`.  d_out: Mat = dLoss/d(hidden[0] of the last layer).
        No gradient flows into the fixed feature columns (models.py:238: trainable=False)."""
        e = self.engine
        tape = self._tape
        d_cur, pre_masked = d_out, False
        for layer in range(len(tape) - 1, -1, -1):
            mode, agg, rows, offsets, outs = tape[layer]
            if mode != "batched":
                raise NotImplementedError("backward through non-contiguous hop inputs (use the model's id buffer)")
            if layer == 0:
                # the fixed features need no gradient; trainable identity columns get theirs scattered per id
                agg.backward_hops(d_cur, pre_masked, embed_sink=self._embed_sink())
                break
            prev_mode, prev_agg, prev_rows, prev_offsets, prev_out = tape[layer - 1]
            d_prev = e.ws_mat((self.name, "d_hidden", layer - 1), prev_out.rows, prev_out.d)
            # the previous layer is never the last one, so its activation is relu (models.py:307-314): its
            # relu gradient is fused into the scatter of this layer's input gradients
            mask = prev_out if prev_agg.act_code == ops.ACT_RELU else None
            agg.backward_hops(d_cur, pre_masked, d_prev=d_prev, prev_mask=mask, prev_offsets=prev_offsets)
            d_cur, pre_masked = d_prev, mask is not None

    # ------------------------------------------------------------------------------ step machinery (shared)
    def _samplers(self):
        seen = []
        for li in self.layer_infos:
            if li.neigh_sampler not in seen:
                seen.append(li.neigh_sampler)
        return seen

    def _sample_phase(self, batch, n, parity, stage=None):
        """Batch/label staging + neighbor sampling into the parity-keyed id buffer (weight-free)."""
        self._parity = parity
        self.reset_tapes()      # a forward-only step (eval) leaves saved activations behind: buffer keys count them
        for s in self._samplers():
            s.new_step()
            # the unsupervised model's one pass over [batch1 | batch2 | negatives] stands for three sample() calls of the
            # reference (:347-357), each with its own column permutations
            s.root_segments = self._root_segments(n)
            s.calls_per_sample = len(self.layer_infos)
        self._pending_stage = stage
        try:
            return self.sample(batch, self.layer_infos, n)
        finally:
            for s in self._samplers():
                s.root_segments = None

    def _root_segments(self, n_roots):
        """(start of batch2, start of the negatives) inside the roots [batch1 | batch2 | negatives] of the unsupervised
        pass (this class); the supervised subclass has one sample() call per step and returns None."""
        B = (n_roots - self.neg_sample_size) // 2
        return (B, 2 * B)

    def _layer0_inputs(self, samples, support_sizes, n):
        hidden = [Rows(self.features, sm, requires_grad=False) for sm in samples]
        self_all, neighs, _, _ = self.layer_inputs(hidden, 0, n, self.num_samples, support_sizes, self.dims, self.concat)
        return self_all, neighs

    def _data_phase(self, batch, n, parity, stage=None):
        """The weight-free half of a step: batch/label staging, neighbor sampling and the layer-0 gather+mean.
        Writes only parity-keyed buffers, so it can run ahead of (or concurrently with) the previous step's compute."""
        samples, support_sizes = self._sample_phase(batch, n, parity, stage)
        self_all, neighs = self._layer0_inputs(samples, support_sizes, n)
        means0 = self.aggregators[0].prefetch(self_all, neighs, tag=parity) if self_all is not None else None
        return samples, support_sizes, means0

    def _schedule_signature(self):
        """Everything that shapes the launches a captured step graph contains: changing one of these on a live model
        (tests and A/B runs do) selects / captures another graph instead of silently replaying the old one."""
        e = self.engine
        law = tuple((s.law, s.max_degree, s.seed) for s in self._samplers())
        return (getattr(self, "fuse_tail", True), getattr(self, "fuse_head", True), getattr(self, "fuse_sampler", True),
                self.sampler_rides, self.cogather_split, self.cogather_split3, self.cogather_tail, self.tail_split,
                self.cogather_auto, self.tail_halves, self.tail_free_bytes, self.sampler_in_wgrad, self.sampler_in_wgrad_max_bytes,
                self.cogather_z, self.cogather_lp_fwd, self.cogather_lp_tail, self.cogather_lp_neg, e.stream_gemm, e.tiled3_fwd, e.tiled3_wgrad, e.split_pool, e.pool_f16, str(getattr(self, "pipeline", None)),
                type(self.grad_hook).__name__,
                id(self.grad_hook), law)

    def _run(self, key, fn):
        """Eager on first use, captured into a hipGraph on the second, replayed afterwards.  The Python attributes
        that name a step's output buffers are snapshotted per key and restored on replay (the Python of `fn` does
        not run again, and other step shapes -- e.g. a validation batch -- may have re-pointed them meanwhile)."""
        e = self.engine
        # the dropout rate and the schedule choices are baked into the captured launches
        key = tuple(key) + (self._dropout_rate(), self._schedule_signature())
        g = self._graphs.get(key)
        if g is not None:
            for name, val in self._graph_outputs[key].items():
                setattr(self, name, val)
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
        except Exception:
            try:                      # leave the stream out of capture mode, or every later launch on it fails too
                g.end()
            except Exception:
                pass
            raise
        g.end()
        self._graphs[key] = g
        self._graph_outputs[key] = {name: getattr(self, name) for name in self._OUT_ATTRS if hasattr(self, name)}
        g.launch()

    def _feed_dropout(self, feed_dict):
        """feed_dict[placeholders['dropout']] (supervised_train.py:117; validation feeds omit it = 0): every layer
        holds the placeholder and reads the rate when it runs."""
        ph = self.placeholders['dropout']
        ph.value = float(feed_dict.get(ph, 0.0))
        return self._dropout_rate()

    def _dropout_rate(self):
        from .layers import _rate
        return _rate(self.placeholders['dropout']) if self.placeholders and 'dropout' in self.placeholders else 0.0

    def _needs_host_rng(self):
        from .neigh_samplers import PaddedAdjacency
        return any(isinstance(s.adj_info.current, PaddedAdjacency) for s in self._samplers())

    def _adj_version(self):
        return tuple(id(s.adj_info.current) for s in self._samplers())

    def reset_tapes(self):
        if self.aggregators:
            for a in self.aggregators:
                a.reset()
        self._tape = None
