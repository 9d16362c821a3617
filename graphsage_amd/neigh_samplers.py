"""Classes that are used to sample node neighborhoods -- call surface of graphsage/neigh_samplers.py.

`UniformNeighborSampler(adj_info)((ids, num_samples))` returns int32 [len(ids), num_samples]
(neigh_samplers.py:24-29).  `adj_info` is a mutable adjacency handle (the analogue of the
`adj_info` tf.Variable that the driver re-assigns between train and test adjacency,
supervised_train.py:147-148,260-261,280,285): the sampler holds a reference, not a copy.

Two adjacency layouts:
  * PaddedAdjacency  -- the reference's [N+1, max_degree] table (minibatch.py:227-259); sampling is
    `adj[ids][:, perm[:num_samples]]` with ONE column permutation per call (exact reference
    semantics; the permutation is drawn on the host or injected for parity tests).
  * CSRAdjacency     -- MI355X-native: rowptr/col on the device, counter-based per-slot draws
    (gs_sample_uniform_csr).  Default for training; hipGraph-replayable (no host RNG).
"""
import os

import numpy as np
import torch

from . import ops
from .layers import Layer


class PaddedAdjacency(object):
    def __init__(self, adj, device):
        adj = np.ascontiguousarray(adj, dtype=np.int32)
        self.table = torch.from_numpy(adj).to(device)
        self.n_nodes = adj.shape[0] - 1
        self.max_degree = adj.shape[1]


class CSRAdjacency(object):
    def __init__(self, rowptr, col, n_nodes, device):
        self.rowptr = torch.from_numpy(np.ascontiguousarray(rowptr, dtype=np.int64)).to(device)
        col = np.ascontiguousarray(col, dtype=np.int32)
        if col.size == 0:
            col = np.zeros((1,), dtype=np.int32)
        self.col = torch.from_numpy(col).to(device)
        self.n_nodes = int(n_nodes)

    @classmethod
    def from_device(cls, rowptr, col, n_nodes):
        """Wrap CSR arrays that already live in HBM (int64 rowptr [N+1], int32 col [E])."""
        assert rowptr.dtype == torch.int64 and col.dtype == torch.int32 and rowptr.numel() == n_nodes + 1
        self = cls.__new__(cls)
        self.rowptr, self.col, self.n_nodes = rowptr.contiguous(), col.contiguous(), int(n_nodes)
        return self


class AdjInfo(object):
    """Mutable handle: `assign()` swaps the adjacency every sampler sees (tf.assign(adj_info, ...))."""

    def __init__(self, adjacency):
        self.current = adjacency
        self.version = 0

    def assign(self, adjacency):
        if adjacency is not self.current:
            self.current = adjacency
            self.version += 1

    @property
    def n_nodes(self):
        return self.current.n_nodes


class UniformNeighborSampler(Layer):
    """Uniformly samples neighbors (graphsage/neigh_samplers.py:15-29)."""

    def __init__(self, adj_info, seed=123, law="iid", max_degree=0, **kwargs):
        """`law` (CSR adjacency only; gs_sample_uniform_csr in include/graphsage_amd.h):
          "iid"        independent uniform draws with replacement from the true neighbor list (default)
          "reference"  the reference's joint law: virtual padded [N+1, max_degree] table frozen for the run
                       (minibatch.py:227-245) + num_samples distinct columns per call shared by all rows
                       (neigh_samplers.py:26-28); needs max_degree (FLAGS.max_degree, 128 by default)
          "distinct"   per-row draws without replacement when the (optionally max_degree-capped) list is long enough"""
        super(UniformNeighborSampler, self).__init__(**kwargs)
        if law not in ops._lib.SAMPLER_LAWS:
            raise ops._lib.GraphsageAmdError("unknown sampler law %r (iid | reference | distinct)" % (law,))
        self.law, self.max_degree = ops._lib.SAMPLER_LAWS[law], int(max_degree)
        if law == "reference" and self.max_degree <= 0:
            raise ops._lib.GraphsageAmdError("sampler law 'reference' needs max_degree > 0 (the padded table width)")
        if not isinstance(adj_info, AdjInfo):
            adj_info = AdjInfo(adj_info)
        self.adj_info = adj_info
        self.seed = int(seed)
        self._rng = np.random.RandomState(seed)
        self._injected_perms = None
        self._call_index = 0          # sampler calls so far in this step = the `hop` stream id
        self.next_out = None          # optional destination of the next call (the model's contiguous id buffer)
        self.global_row_offset = 0    # first global row of this rank's slice (data-parallel invariance)
        # unsupervised model: the roots [batch1 | batch2 | negatives] of ONE pass stand for the reference's THREE sample()
        # calls (models.py:347-357), each with its own column permutation per hop (neigh_samplers.py:27).  root_segments =
        # (start of batch2, start of the negatives) in roots; calls_per_sample = K.  Under law "reference" (and on the
        # padded table) segment g of call k then uses the reference's call id g * K + k.
        self.root_segments = None
        self.call_segments = None     # the same boundaries in rows of the NEXT hop-by-hop call (set by model.sample)
        self.calls_per_sample = 1
        # law "reference": materialise the law's padded table once per adjacency (what minibatch.py:227-245 builds; 119 MB
        # for Reddit at max_degree 128) so that a draw is one lookup -- same ids bit for bit, see padded_table()
        self.use_table = True          # False: the law's padded table stays virtual (evaluated per draw)

    def padded_table(self, adj):
        """The reference law's padded table of `adj` ([N+1, max_degree] int32 on the device, gs_build_padded_table), built on
        first use and cached on the adjacency object per (seed, max_degree); None for the other laws, when switched off, or
        when it would exceed GS_SAMPLER_TABLE_MAX_GB (default 16): the sampler then evaluates the same entries per draw."""
        if self.law != ops._lib.SAMPLER_LAWS["reference"] or not self.use_table or not isinstance(adj, CSRAdjacency):
            return None
        cache = adj.__dict__.setdefault("_padded_tables", {})
        key = (self.seed, self.max_degree)
        if key not in cache:
            limit = float(os.environ.get("GS_SAMPLER_TABLE_MAX_GB", "16")) * (1 << 30)
            if (adj.n_nodes + 1) * self.max_degree * 4 > limit:
                cache[key] = None
            else:
                e = self.engine
                e.sync()
                cache[key] = ops.build_padded_table(adj.rowptr, adj.col, adj.n_nodes, adj.n_nodes, self.max_degree, self.seed,
                                                    stream=e.stream)
                e.sync()
        return cache[key]

    # -- step bookkeeping driven by the model -------------------------------------------------
    def new_step(self):
        self._call_index = 0

    def inject_perms(self, perms):
        """Parity tests: column permutations to use for the next calls (padded layout only).  With root segments (the
        unsupervised model) the list is indexed by the reference's call id g * K + k instead of being consumed in order."""
        self._injected_perms = list(perms) if perms is not None else None

    def _segment_slices(self, n):
        """[(segment, row_begin, row_end)] of the next hop-by-hop call."""
        segs = self.call_segments
        self.call_segments = None
        if segs is None:
            return [(0, 0, n)]
        b1, b2 = int(segs[0]), int(segs[1])
        assert 0 <= b1 <= b2 <= n
        return [(g, lo, hi) for g, lo, hi in ((0, 0, b1), (1, b1, b2), (2, b2, n)) if hi > lo]

    def _call(self, inputs):
        ids, num_samples = inputs
        e = self.engine
        adj = self.adj_info.current
        n = ids.numel()
        out = self.next_out if self.next_out is not None else e.ws_i32((self.name, "out", self._call_index),
                                                                        n * num_samples)
        self.next_out = None
        assert out.numel() >= n * num_samples
        segmented = self.call_segments is not None
        slices = self._segment_slices(n)
        if isinstance(adj, PaddedAdjacency):
            if num_samples > adj.max_degree:
                raise ops._lib.GraphsageAmdError("num_samples %d > max_degree %d" % (num_samples, adj.max_degree))
            for g, lo, hi in slices:
                if self._injected_perms and segmented:
                    perm = np.asarray(self._injected_perms[g * self.calls_per_sample + self._call_index])[:num_samples]
                elif self._injected_perms:
                    perm = np.asarray(self._injected_perms.pop(0))[:num_samples]
                else:
                    perm = self._rng.permutation(adj.max_degree)[:num_samples]
                perm_dev = e.ws_i32((self.name, "perm", self._call_index, g), num_samples)
                perm_dev.copy_(torch.from_numpy(np.ascontiguousarray(perm, dtype=np.int32)))
                torch.cuda.current_stream().synchronize()
                ops.sample_padded(adj.table, ids[lo:hi], perm_dev, num_samples, out=out[lo * num_samples: hi * num_samples],
                                  stream=e.stream)
        elif segmented and self.law == ops._lib.SAMPLER_LAWS["reference"]:
            for g, lo, hi in slices:
                ops.sample_uniform_csr(adj.rowptr, adj.col, adj.n_nodes, adj.n_nodes, ids[lo:hi], num_samples, self.seed,
                                       step=0, step_dev=e.sample_clock_dev, hop=g * self.calls_per_sample + self._call_index,
                                       global_row_offset=self.global_row_offset + lo,
                                       out=out[lo * num_samples: hi * num_samples], stream=e.stream, law=self.law,
                                       max_degree=self.max_degree)
        else:
            ops.sample_uniform_csr(adj.rowptr, adj.col, adj.n_nodes, adj.n_nodes, ids, num_samples, self.seed,
                                   step=0, step_dev=e.sample_clock_dev, hop=self._call_index,
                                   global_row_offset=self.global_row_offset, out=out, stream=e.stream,
                                   law=self.law, max_degree=self.max_degree)
        self._call_index += 1
        return out[: n * num_samples].view(n, num_samples)

    def fanout(self, ids_all, offsets, fans, batch_size, root_offset=0, stage=None):
        """All hops of a mini-batch in ONE launch (gs_sample_fanout_csr): `fans[h]` is the fan-out of the h-th
        sampler call.  `stage` = (order, cursor_dev, label_table, labels_out) additionally selects the batch from
        the device-resident epoch order and gathers its label rows in the same launch.  CSR adjacency only."""
        e = self.engine
        adj = self.adj_info.current
        assert isinstance(adj, CSRAdjacency)
        order = cursor = table = labels_out = None
        if stage is not None and len(stage) and isinstance(stage[0], str) and stage[0] == "unsup":
            # the unsupervised model's roots [batch1 | batch2 | negatives] are staged by the launch itself
            _, pairs, cursor, n_pair_roots, cdf, guide, guide_bits, n_neg, neg_seed = stage
            desc = ops.fanout_desc(adj.rowptr, adj.col, adj.n_nodes, adj.n_nodes, fans, offsets, ids_all, batch_size,
                                   self.seed, step_dev=e.sample_clock_dev, hop0=self._call_index, root_offset=root_offset,
                                   cursor_dev=cursor, law=self.law, max_degree=self.max_degree,
                                   unsup=(pairs, n_pair_roots, cdf, guide, guide_bits, n_neg, neg_seed),
                                   padded_table=self.padded_table(adj), segments=self.root_segments)
            if getattr(e, "_defer_sampler", False):
                e._deferred_sampler = desc
            else:
                ops.sample_fanout_desc(desc, stream=e.stream)
            self._call_index += len(fans)
            return
        if stage is not None:
            order, cursor, table, labels_out = stage
        ptable = self.padded_table(adj)
        if getattr(e, "_defer_sampler", False) or ptable is not None or self.root_segments is not None:
            desc = ops.fanout_desc(adj.rowptr, adj.col, adj.n_nodes, adj.n_nodes, fans, offsets, ids_all,
                                   batch_size, self.seed, step_dev=e.sample_clock_dev, hop0=self._call_index,
                                   root_offset=root_offset, order=order, cursor_dev=cursor,
                                   label_table=table, labels_out=labels_out, law=self.law,
                                   max_degree=self.max_degree, padded_table=ptable, segments=self.root_segments)
            if getattr(e, "_defer_sampler", False):
                # not launched here: the descriptor rides in the step's optimizer launch (engine.finish_backward)
                e._deferred_sampler = desc
            else:
                ops.sample_fanout_desc(desc, stream=e.stream)
        else:
            ops.sample_fanout_csr(adj.rowptr, adj.col, adj.n_nodes, adj.n_nodes, fans, offsets, ids_all, batch_size,
                                  self.seed, step_dev=e.sample_clock_dev, hop0=self._call_index, root_offset=root_offset,
                                  order=order, cursor_dev=cursor, label_table=table, labels_out=labels_out,
                                  stream=e.stream, law=self.law, max_degree=self.max_degree)
        self._call_index += len(fans)
