"""Execution context of the MI355X engine: the launch stream, persistent workspaces (so a whole
step can be captured into a hipGraph and replayed), trainable variables in ONE flat fp32 buffer
(one Adam launch, one RCCL all-reduce), and split-K gradient slab arenas.

This replaces what the TF1 runtime provided to the reference: tf.Variable storage
(supervised_models.py:60, aggregators.py:30-33), gradient accumulation of
optimizer.compute_gradients (supervised_models.py:95) and apply_gradients (:99).
"""
import ctypes
import os

import numpy as np
import torch

from . import ops
from .ops import Mat, round_up

MAX_SLABS = 64  # slab capacity per variable (split-K partial sums of one backward pass)


class Variable(object):
    """A trainable fp32 matrix [rows, cols] living in the engine's flat parameter buffer."""

    def __init__(self, name, init, decay=False, scatter=False):
        init = np.asarray(init, dtype=np.float32)
        if init.ndim == 1:
            init = init[None, :]
        self.name = name
        self.rows, self.cols = init.shape
        self.ld = round_up(self.cols, 4)
        self.init = init
        self.decay = decay  # member of aggregator.vars / node_pred.vars -> weight-decayed (supervised_models.py:104-108)
        # scatter: the gradient arrives through atomic row scatters (a gathered embedding table), not as split-K
        # slabs: ONE accumulator slab that the reduce kernel zeroes after reading it
        self.scatter = scatter
        self.offset = None  # float offset in the flat buffers
        self.value = None   # Mat view into engine.params
        self.grad = None    # Mat view into engine.grads
        self.slabs = None   # flat tensor [MAX_SLABS * rows * ld]
        self.n_slabs = 0    # slabs written so far in the current backward pass
        # three-piece bf16 copy of value^T for the split-MFMA contractions (gs_split_rows); made by Engine.split_of on first
        # use, re-made after every update of the value
        self.split3 = None
        self.split_dirty = True
        self.split_form = "bf16x3"   # or "f16x2": two fp16 pieces under a column scale (gs_split_rows_f16, the pooling MLP)
        # one buffer per form, never freed or replaced: captured step graphs hold these pointers and re-cut into them
        # (Engine._params_updated), so switching Engine.pool_f16 on a live model must not move them
        self.splits = {}
        self.engine = None  # set by Engine.add_variable

    @property
    def size(self):
        return self.rows * self.ld

    def numpy(self):
        return self.value.numpy()

    def assign(self, a):
        a = np.asarray(a, dtype=np.float32).reshape(self.rows, self.cols)
        self.value.buf[:, : self.cols].copy_(torch.from_numpy(a))
        self.split_dirty = True
        if self.split3 is not None and self.engine is not None:
            # a captured step graph does not re-run split_of's Python: bring the bf16 pieces up to date right here
            torch.cuda.current_stream().synchronize()
            self.recut(self.engine.stream)

    def recut(self, stream):
        """Bring every cut copy of the value up to date (one per form a consumer has asked for, Engine.split_of)."""
        for form, buf in self.splits.items():
            if form == "f16x2":
                ops.split_rows_f16(self.value, out=buf, stream=stream)
            else:
                ops.split_rows(self.value, out=buf, stream=stream)
        self.split_dirty = False

    def slab_ptr(self, k):
        return self.slabs.data_ptr() + 4 * k * self.size


class Engine(object):
    def __init__(self, device=None, stream=None):
        if not torch.cuda.is_available():
            raise ops._lib.GraphsageAmdError("graphsage_amd needs a HIP device (no CPU fallback)")
        ops._lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self._stream_obj = ops.Stream() if stream is None else None
        self.stream = self._stream_obj.handle if stream is None else stream
        self.variables = []
        self.params = self.grads = self.adam_m = self.adam_v = None
        self._ws = {}
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)  # optimizer step counter t-1
        self.sample_clock_dev = torch.zeros(1, dtype=torch.int64, device=self.device)  # sampler RNG step
        self.finalized = False
        self._pending = []
        self.post_update_hooks = []
        self.dropout_seed = 123
        self._n_sites = 0
        # Schedule choices are plain attributes (tests and A/B scripts set them on a live engine; Model._schedule_signature
        # keys the captured graphs on them); only the few a deployment may want to flip are read from the environment.
        # "stream" contraction kernels (gs_stream.hip) for the layer-0 forward and the grouped weight gradients; False = the
        # LDS-tiled family of gs_gemm.hip
        self.stream_gemm = True
        # the layer-0 forward LDS-tiled on the bf16 matrix pipe (three-piece arithmetic, gs_sage_dense_fwd_tiled3) instead of the
        # register-streaming fp32-MFMA kernel
        self.tiled3_fwd = os.environ.get("GS_TILED3_FWD", "1") == "1"
        # the pooling MLP on the step's distinct ids as a split-MFMA contraction (gs_split16.hip / gs_split.hip); False = fp32 MFMA
        self.split_pool = True
        # split-K policy of the weight gradients (measured sweeps: DESIGN.md section 4 / profiles/r02..r05)
        self.tiled3_max_slabs = TILED3_MAX_SLABS   # (sweep: RMAT 57.6 us/step at 24, 59.2 at 20, 60.6 at 16, 64.0 at 12; Reddit 94.6 at 11, 99.1 at 8)
        self._wgrad_blocks = 768          # tiled kernel: ~768 (tile x slice) workgroups per problem
        self._wgrad_max_slabs = 32
        self._wgrad_big_n = 16384         # reductions this long take their own 128 x 128-tile launch
        self._stream_max_slabs = 32
        # contraction waves of ONE round: one per SIMD (4 per CU); the stream weight-gradient launch is cut to fit it (launch_wgrads)
        self._stream_wave_slots = (4 * torch.cuda.get_device_properties(self.device).multi_processor_count
                                   if self.device.type == "cuda" else 1024)
        self._stream_slice_rows = 256.0   # stream kernel: ~one wave per SIMD with >= 256 reduction rows per slice
        # the grouped weight gradients LDS-tiled on the bf16 matrix pipe (three-piece arithmetic, gs_dense_wgrad_grouped_tiled3): one
        # 8-wave workgroup per (64 x 128 tile, slice), cut so that the launch is ONE round of workgroups (one per CU)
        self.tiled3_wgrad = os.environ.get("GS_TILED3_WGRAD", "1") == "1"
        self.last_wgrad_kernel = None     # "tiled3" | "stream/tiled": what the last grouped weight-gradient launch took (tests)
        self._tiled3_wg_slots = (torch.cuda.get_device_properties(self.device).multi_processor_count
                                 if self.device.type == "cuda" else 256)
        self._injected_keep = {}          # dropout site -> injected keep bits (parity tests)
        self._table16 = {}                # constant feature tables cut into two fp16 pieces (table16_of)
        # feature tables some launch of the step REWRITES (identity features: the trainable leading columns are refreshed behind
        # every optimizer launch, SampleAndAggregate._init_features): data_ptr of their buffers.  A copy of such a table cut once
        # (table16_of) would go stale after the first update, so consumers ask is_constant_table() first.
        self.mutable_tables = set()
        # the pooling MLP on the fp16 matrix pipe with two-piece operands (gs_split16.hip): half the matrix-pipe work of the
        # three-piece bf16 form, same accuracy class; needs a constant feature table (no trainable identity features)
        self.pool_f16 = os.environ.get("GS_POOL_F16", "1") == "1"
        self._split_vars = []             # variables with a three-piece bf16 copy, re-cut behind every optimizer launch
        self._defer_sampler = False       # neigh_samplers.fanout: hand the launch to the next optimizer launch instead
        self._deferred_sampler = None
        self._sampler_to_wgrad = False    # ... to the weight-gradient launch instead (set by the model around a step's launches)
        self.last_wgrad_sampler = False

    # -------------------------------------------------------------------------------- variables
    def add_variable(self, name, init, decay=False, scatter=False):
        assert not self.finalized, "variables must be created before Engine.finalize()"
        v = Variable(name, init, decay, scatter)
        v.engine = self
        self.variables.append(v)
        return v

    def new_site(self):
        """A block of 16 dropout call-site ids (role + 4 * call index) for one layer object."""
        self._n_sites += 1
        return 16 * self._n_sites

    def dropout(self, rate, site, row0=0):
        """gs_dropout descriptor keyed by the device step clock (None when rate == 0)."""
        return ops.dropout_desc(self.dropout_seed, self.sample_clock_dev, site, rate, row0, keep=self._injected_keep.get(site))

    def inject_dropout_masks(self, masks):
        """Parity tests: {site id: uint8 array [rows, d] of keep bits} used INSTEAD of the counter hash by every dropout call
        of that site (rows = the site's global row index: all hops of a layer in call order) until cleared with None.  The
        reference's masks come from TF's RNG, so they are injected like the sampler's permutations."""
        self._injected_keep = {}
        for site, m in (masks or {}).items():
            m = np.ascontiguousarray(m, dtype=np.uint8)
            buf = torch.zeros((m.shape[0], round_up(m.shape[1], 4)), dtype=torch.uint8, device=self.device)
            buf[:, : m.shape[1]].copy_(torch.from_numpy(m))
            self._injected_keep[int(site)] = buf
        torch.cuda.synchronize()

    def finalize(self):
        """Lay all variables out in one flat buffer (16-byte aligned segments)."""
        if self.finalized:
            return
        off = 0
        for v in self.variables:
            v.offset = off
            off += v.size
        total = max(off, 4)
        self.n_param_floats = total
        self.params = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.adam_m = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.adam_v = torch.zeros(total, dtype=torch.float32, device=self.device)
        for v in self.variables:
            v.value = Mat(self.params[v.offset: v.offset + v.size].view(v.rows, v.ld), v.cols)
            v.grad = Mat(self.grads[v.offset: v.offset + v.size].view(v.rows, v.ld), v.cols)
            v.value.buf[:, : v.cols].copy_(torch.from_numpy(v.init))
            v.slabs = torch.zeros((1 if v.scatter else MAX_SLABS) * v.size, dtype=torch.float32, device=self.device)
            v.n_slabs = 1 if v.scatter else 0
        self._init_params = self.params.clone()
        torch.cuda.synchronize()
        self.finalized = True

    def snapshot_initial_parameters(self):
        """Make the CURRENT weights the ones reset_parameters() returns to (after assigning weights from outside)."""
        self.sync()
        torch.cuda.synchronize()
        self._init_params.copy_(self.params)
        torch.cuda.synchronize()
        self._params_updated()
        self.sync()

    def reset_parameters(self):
        """Back to the initial weights with fresh optimizer state (Adam moments, step counter, sampler clock): a second
        training run on the same model / captured graphs (bench.py's micro-F1 leg)."""
        self.sync()
        self.params.copy_(self._init_params)
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.grads.zero_()
        self.step_dev.zero_()
        self.sample_clock_dev.zero_()
        torch.cuda.synchronize()
        self._params_updated()
        self.sync()

    def n_trainable(self):
        return sum(v.rows * v.cols for v in self.variables)

    # -------------------------------------------------------------------------------- workspaces
    def ws_mat(self, key, rows, d, ld_multiple=4):
        """Persistent [rows, d] matrix keyed by (key, rows, d): same buffer every step."""
        k = ("m", key, rows, d)
        m = self._ws.get(k)
        if m is None:
            m = Mat.zeros(rows, d, self.device, ld_multiple)
            torch.cuda.synchronize()  # the zero-fill ran on torch's stream; order it before ours
            self._ws[k] = m
        return m

    def ws_i32(self, key, n):
        k = ("i", key, n)
        t = self._ws.get(k)
        if t is None:
            t = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
            torch.cuda.synchronize()
            self._ws[k] = t
        return t

    def ws_f32(self, key, n):
        k = ("f", key, n)
        t = self._ws.get(k)
        if t is None:
            t = torch.zeros(max(n, 1), dtype=torch.float32, device=self.device)
            torch.cuda.synchronize()
            self._ws[k] = t
        return t

    # -------------------------------------------------------------------------------- gradients
    def begin_backward(self):
        for v in self.variables:
            v.n_slabs = 1 if v.scatter else 0
        self._pending = []

    def pick_slabs(self, n_rows, tiles):
        """Split-K slices for a weight gradient.  The kernel is latency-bound per workgroup (one global round trip
        per 32-row stage), so the goal is many short workgroups: ~768 (tile x slice) workgroups per problem, at
        least 64 reduction rows per slice, at most 32 slices."""
        target, cap = self._wgrad_blocks, self._wgrad_max_slabs
        want = max(1, (target + tiles - 1) // tiles)
        return int(max(1, min(cap, want, (n_rows + 63) // 64)))

    def ones(self, n):
        """[n, 1] matrix of ones: bias gradients are the grouped-GEMM problem ones^T · dZ."""
        m = self._ws.get(("ones",))
        if m is None or m.rows < n:
            m = Mat(torch.ones((max(n, 1024), 4), dtype=torch.float32, device=self.device), 1)
            torch.cuda.synchronize()
            # captured hipGraphs hold the pointer of every earlier (smaller) buffer: keep them all alive
            self._ws.setdefault(("ones_keepalive",), []).append(m)
            self._ws[("ones",)] = m
        return m.rows_slice(0, n)

    def wgrad(self, var, A, a_idx, dZ, col0, n):
        """Queue var.slabs += A[a_idx]^T · dZ[:, col0:col0+var.cols] (split-K slabs); all queued problems of a
        backward pass are issued as ONE grouped launch by launch_wgrads()."""
        assert A.d == var.rows, (var.name, A.d, var.rows)
        big = n >= self._wgrad_big_n and var.rows >= 128 and var.cols >= 128   # throughput-bound: own launch with 128x128 tiles
        t = 128 if big else 64
        tiles = ((var.rows + t - 1) // t) * ((var.cols + t - 1) // t)
        if big:
            k = self.pick_slabs(n, tiles)
            if var.n_slabs + k > MAX_SLABS:
                raise ops._lib.GraphsageAmdError("slab arena of %s exhausted" % var.name)
            ops.call("gs_dense_wgrad", A.ptr, A.ld, ops.ptr(a_idx), A.d, dZ.ptr, dZ.ld, col0, var.cols, n, k,
                     var.slab_ptr(var.n_slabs), var.ld, self.stream)
            var.n_slabs += k
            return
        # queued: slab counts are assigned in launch_wgrads(), once it is known whether EVERY problem of the pass can take the
        # stream kernel (its slab count is tuned differently from the tiled kernel's)
        self._pending.append((var, A, a_idx, dZ, col0, n, tiles))

    def _stream_slabs(self, var, A, a_idx, dZ, n, reserved=0):
        """(eligible for the stream kernel, its slab count): one WAVE per (64x64 tile, slice), ~1 contraction wave per SIMD over
        the whole launch (~900 items for the Reddit step) with >= 256 reduction rows per slice."""
        ks = int(max(1, min(self._stream_max_slabs, MAX_SLABS - var.n_slabs - reserved, round(n / self._stream_slice_rows))))
        lda, ldz = A.ld, dZ.ld
        if a_idx is not None:
            # (32-bit BYTE offsets of the gathered rows: tables beyond 4 GB -- RMAT's 10 GB -- take the tiled kernel; a wide-offset
            #  form of the stream kernel was measured slower there, 95.2 vs 77.3 us/step: benchmarks/variants/README.md)
            ok = (n + ks - 1) // ks <= 510 and (A.rows + 1) * lda * 4 < 1 << 32 and (n + 1) * ldz * 4 < 1 << 32
        else:
            ok = (n + 1) * max(lda, ldz) * 4 < 1 << 32
        return ok, ks

    def _fit_one_round(self, ks_list):
        """The stream weight-gradient launch runs one contraction WAVE per (64 x 64 tile, slice) and a wave holds its SIMD for its
        whole slice: with between one and two waves per SIMD the launch takes TWO wave times although the second round is mostly
        empty (unsupervised Reddit step: 40 tiles x 32 slices = 1280 waves of 359 rows on 1024 SIMDs = 2 x 359 row times;
        25 slices = 1000 waves of 460 rows = 1 x 460 -- and 7 fewer slabs for the optimizer launch to sum: 186.9 vs 190.0
        us/step, gpurun_out/r5z).  Slab counts of the problems that are cut at all are scaled down to fit one round when that
        keeps a row-gathered problem's slices within the 510 rows its offsets have registers for; launches beyond ~2 rounds
        are left alone (the quantisation matters less and less)."""
        slots = self._stream_wave_slots
        waves = sum(p[6] * k for p, k in zip(self._pending, ks_list))
        if not (slots < waves < 1.8 * slots):
            return ks_list
        fixed = sum(p[6] * k for p, k in zip(self._pending, ks_list) if k <= 1)
        f = (slots - fixed) / float(max(1, waves - fixed))
        out = []
        for (var, A, a_idx, dZ, col0, n, tiles), k in zip(self._pending, ks_list):
            k2 = max(1, int(k * f)) if k > 1 else k
            if a_idx is not None and (n + k2 - 1) // k2 > 510:
                return ks_list                    # a gathered problem would outgrow its slice: keep the launch as it was
            out.append(k2)
        return out

    def _tiled3_slabs(self):
        """Slab counts for gs_dense_wgrad_grouped_tiled3 (tiled3_slab_policy over the pending problems), or None when a problem
        cannot take it."""
        if len(self._pending) > 12:
            return None
        probs, reserved = [], {}
        for v, A, ai, dZ, _, n, _ in self._pending:
            tiles = ((v.rows + 63) // 64) * ((v.cols + 127) // 128)
            cap = MAX_SLABS - v.n_slabs - reserved.get(id(v), 0)
            if cap < (n + 1023) // 1024:
                return None
            probs.append((tiles, n, cap))
            reserved[id(v)] = reserved.get(id(v), 0) + (n + 1023) // 1024    # (what the later problems of the variable can count on)
        ks = tiled3_slab_policy(probs, self._tiled3_wg_slots, self.tiled3_max_slabs)
        used = {}
        for (v, *_), k in zip(self._pending, ks):               # several problems of one variable share its arena
            used[id(v)] = used.get(id(v), 0) + k
        for v in {id(p[0]): p[0] for p in self._pending}.values():
            if v.n_slabs + used.get(id(v), 0) > MAX_SLABS:
                return None
        return ks

    def _assign_slabs(self, ks_list):
        """Pending problems -> gs_wgrad_desc list with slabs assigned behind what each variable already holds.  ks_list: the
        slab count of every pending problem, decided by launch_wgrads (stream policy, or None = tiled policy)."""
        descs = []
        for (var, A, a_idx, dZ, col0, n, tiles), k in zip(self._pending, ks_list):
            if k is None:
                k = self.pick_slabs(n, tiles)
            if var.n_slabs + k > MAX_SLABS:
                raise ops._lib.GraphsageAmdError("slab arena of %s exhausted" % var.name)
            d = ops._lib.WgradDesc()
            d.A, d.a_idx, d.dZ = A.ptr, ops.ptr(a_idx), dZ.ptr
            d.slabs = var.slab_ptr(var.n_slabs)
            d.lda, d.ldz, d.ld_slab, d.n = A.ld, dZ.ld, var.ld, n
            d.d, d.col0, d.out_dim, d.n_slabs = var.rows, col0, var.cols, k
            d.a_rows = A.rows if a_idx is not None else 0
            descs.append(d)
            var.n_slabs += k
        self._pending = []
        return descs

    def bgrad(self, var, dZ, n, n_cols, col0=0):
        """Bias gradient = column sums of dZ[:, col0:col0+n_cols] = ones^T · dZ (one more grouped problem)."""
        assert var.rows == 1 and var.cols == n_cols
        self.wgrad(var, self.ones(n), None, dZ, col0, n)

    def sparse_pool_wgrad(self, var, X, ids, n_groups, s, argmax, dpm):
        """MaxPool MLP weight gradient from the arg-max rows only (gs_maxpool_sparse_wgrad): new slabs of `var`."""
        k = int(max(1, min(48, (n_groups + 31) // 32, MAX_SLABS - var.n_slabs)))
        if k < 1 or var.n_slabs + k > MAX_SLABS:
            raise ops._lib.GraphsageAmdError("slab arena of %s exhausted" % var.name)
        ops.maxpool_sparse_wgrad(X, ids, n_groups, s, argmax, dpm, var.cols, k, var.slab_ptr(var.n_slabs), var.ld,
                                 stream=self.stream)
        var.n_slabs += k

    def scatter_grad(self, var, d, ids, n, s, scale):
        """var.grad[ids[i*s + j], :] += scale * d[i, :var.cols]: gradient of a row gather from a trainable table."""
        assert var.scatter
        ops.scatter_add_rows(d, n, s, var.cols, scale, ids, Mat(var.slabs.view(var.rows, var.ld), var.cols),
                             stream=self.stream)

    def launch_wgrads(self, side_jobs=None):
        """ONE grouped launch for every queued weight gradient; `side_jobs` (gather+mean descriptors of the next step)
        ride along in the same launch (horizontal fusion)."""
        if not self._pending:
            self.launch_gather_jobs(side_jobs)
            return
        # the whole pass takes the stream kernel, or the tiled one: decided here, over every queued problem, BEFORE any slab
        # is assigned (each kernel has its own slab policy).  The stream kernel's (eligible, slab count) is computed ONCE per
        # problem, in order, against the slabs the earlier problems of the same variable will have taken (`reserved`), and
        # those counts are the ones assigned -- so a variable whose arena is nearly full cannot shrink a gathered problem's
        # slab count below what the eligibility check saw.
        if self.stream_gemm and self.tiled3_wgrad:
            ks3 = self._tiled3_slabs()
            if ks3 is not None:
                pending = self._assign_slabs(ks3)
                jobs = list(side_jobs or ())
                arr = (ops._lib.WgradDesc * len(pending))(*pending)
                jarr = (ops._lib.GatherDesc * max(len(jobs), 1))(*jobs)
                rider = self._deferred_sampler if getattr(self, "_sampler_to_wgrad", False) else None
                if rider is not None:
                    # a later mini-batch's fan-out sampler rides in THIS launch (the caller has made sure that no problem gathers
                    # through the id buffer it fills; the library checks)
                    self._deferred_sampler = None
                    ops.call("gs_dense_wgrad_grouped_tiled3_sample", ctypes.addressof(arr), len(pending), ctypes.addressof(jarr),
                             len(jobs), ctypes.addressof(rider), self.stream)
                else:
                    ops.call("gs_dense_wgrad_grouped_tiled3", ctypes.addressof(arr), len(pending), ctypes.addressof(jarr), len(jobs),
                             self.stream)
                self.last_wgrad_kernel = "tiled3"
                self.last_wgrad_sampler = rider is not None
                return
        self.last_wgrad_kernel = "stream/tiled"
        stream, ks_list, reserved = self.stream_gemm, [], {}
        for v, A, ai, dZ, _, n, _ in self._pending:
            ok, ks = self._stream_slabs(v, A, ai, dZ, n, reserved=reserved.get(id(v), 0))
            stream = stream and ok
            ks_list.append(ks)
            reserved[id(v)] = reserved.get(id(v), 0) + ks
        if stream:
            ks_list = self._fit_one_round(ks_list)
        pending = self._assign_slabs(ks_list if stream else [None] * len(self._pending))
        if stream:
            jobs = list(side_jobs or ())
            for a in range(0, len(pending), 12):       # the kernel takes up to 12 problems per launch
                chunk = pending[a:a + 12]
                arr = (ops._lib.WgradDesc * len(chunk))(*chunk)
                jarr = (ops._lib.GatherDesc * max(len(jobs), 1))(*jobs)
                ops.call("gs_dense_wgrad_grouped_stream", ctypes.addressof(arr), len(chunk), ctypes.addressof(jarr),
                         len(jobs), self.stream)
                jobs = []
            return
        arr = (ops._lib.WgradDesc * len(pending))(*pending)
        if side_jobs:
            jarr = (ops._lib.GatherDesc * len(side_jobs))(*side_jobs)
            ops.call("gs_dense_wgrad_grouped_cogather", ctypes.addressof(arr), len(pending), ctypes.addressof(jarr),
                     len(side_jobs), self.stream)
        else:
            ops.call("gs_dense_wgrad_grouped", ctypes.addressof(arr), len(pending), self.stream)

    def _var_descs(self):
        arr = (ops._lib.VarDesc * len(self.variables))()
        for i, v in enumerate(self.variables):
            arr[i].offset, arr[i].size = v.offset, v.size
            arr[i].slabs = v.slabs.data_ptr()
            arr[i].n_slabs, arr[i].decay = v.n_slabs, 1 if v.decay else 0
            arr[i].clear = 1 if v.scatter else 0
        return arr

    def finish_backward(self, weight_decay, fuse_adam=False, lr=0.0, clip=5.0, grad_scale=1.0, side_jobs=None,
                        loss=None, step_offset=1):
        """One grouped launch for every queued weight gradient, then ONE launch that sums the slabs into the
        flat gradient buffer (+ weight decay) and, if fuse_adam, applies clip + Adam in the same pass.
        loss = (loss_rows, n, scale, loss_out, accumulate): the step's scalar loss is formed by that launch too;
        step_offset = 0 when an earlier launch of the step has already advanced the optimizer step counter.
        side_jobs: gather+mean descriptors of the next step riding in the weight-gradient launch."""
        self.launch_wgrads(side_jobs)
        arr = self._var_descs()
        lr_, ln, lscale, lout, lacc = loss if loss is not None else (None, 0, 0.0, None, False)
        args = (ctypes.addressof(arr), len(self.variables), ops.ptr(self.params),
                ops.ptr(self.grads), ops.ptr(self.adam_m), ops.ptr(self.adam_v), self.n_param_floats,
                float(weight_decay), 1 if fuse_adam else 0, lr, 0.9, 0.999, 1e-8, clip, grad_scale,
                ops.ptr(self.step_dev), int(step_offset), ops.ptr(lr_), ln, float(lscale), ops.ptr(lout),
                1 if lacc else 0)
        rider = getattr(self, "_deferred_sampler", None)
        if rider is not None:
            # a later mini-batch's fan-out sampler rides in this launch (neigh_samplers.fanout under _defer_sampler)
            self._deferred_sampler = None
            ops.call("gs_flat_reduce_adam_sample", *args, ctypes.addressof(rider), None, 0, self.stream)
        else:
            ops.call("gs_flat_reduce_adam", *args, self.stream)
        if fuse_adam:
            self._params_updated()

    def adam(self, lr, clip=5.0, grad_scale=1.0, step_offset=1):
        """Separate optimizer launch (data-parallel path: runs after the RCCL all-reduce of engine.grads)."""
        ops.adam_step(self.params, self.grads, self.adam_m, self.adam_v, self.n_param_floats, lr, self.step_dev,
                      clip=clip, grad_scale=grad_scale, step_offset=step_offset, stream=self.stream)
        self._params_updated()

    def _params_updated(self):
        """Launches that must follow every optimizer step (e.g. refreshing a materialised copy of a variable).
        Every variable that has a three-piece bf16 copy (split_of) is re-cut HERE, right behind the launch that changed it:
        inside a captured step the re-cut is then part of the graph whatever the host-side state was at capture time (a
        host flag read at capture decided it before round 5 -- a train graph captured with the flag clean replayed the
        pooling MLP on stale pieces)."""
        for v in self._split_vars:
            v.recut(self.stream)
        for hook in self.post_update_hooks:
            hook()

    def split_of(self, var, form="bf16x3"):
        """The current three-piece copy of var.value^T (gs_split_rows).  Made on first use; from then on re-made behind every
        optimizer launch (_params_updated) and -- for values written from the host (Variable.assign sets split_dirty) --
        here.  A dirty flag met while capturing records one redundant re-cut in the graph, never a missing one."""
        if form not in var.splits:
            K, N = var.rows, var.cols
            words = ops.split_rows_f16_words(K, N) if form == "f16x2" else ops.split_rows_words(K, N)
            # a NEW buffer per form; the other form's buffer (and the graphs that captured its pointer) stay valid
            var.splits[form] = torch.empty(words, dtype=torch.int32, device=self.device)
            var.split_dirty = True
            if var not in self._split_vars:
                self._split_vars.append(var)
        var.split3, var.split_form = var.splits[form], form
        if var.split_dirty:
            var.recut(self.stream)
        return var.split3

    def is_constant_table(self, X):
        """No launch of the step rewrites this feature table (see mutable_tables)."""
        return X.buf.data_ptr() not in self.mutable_tables

    def table16_bytes(self, X):
        """HBM the two-piece fp16 copy of X takes: [rows][2 pieces][KP] fp16 + one int32 exponent per row."""
        return X.rows * (2 * 2 * round_up(X.d, 64) + 4)

    def table16_fits(self, X):
        """The copy is a second table-sized allocation (Reddit: 596 MB, but several GB at GS_POOL_DEDUP_MAX_RATIO = 16 on a
        10^7-row table): taken only within GS_TABLE16_MAX_GB (default 16; beyond it the three-piece bf16 kernel, which cuts its
        rows in registers, runs instead)."""
        return self.table16_bytes(X) <= float(os.environ.get("GS_TABLE16_MAX_GB", "16")) * (1 << 30)

    def table16_of(self, X):
        """The two-piece fp16 copy of a CONSTANT feature table (gs_split_table_f16): made once per table (keyed by its buffer), on
        the first -- eager -- execution of a step; (X2, row exponents)."""
        if not self.is_constant_table(X):
            raise ops._lib.GraphsageAmdError("table16_of: the table has trainable columns (identity features); a copy cut once "
                                             "would go stale after the first optimizer step")
        key = (X.ptr, X.rows, X.d, X.ld)
        hit = self._table16.get(key)
        if hit is None:
            hit = ops.split_table_f16(X, stream=self.stream) + (X.buf,)      # (the table itself is kept alive with its copy)
            self._table16[key] = hit
        return hit[0], hit[1]

    def advance(self, step=0, clock=0, cursor=None, cursor_delta=0, loss_rows=None, n=0, loss_out=None, accumulate=False,
                aux_rows=None, aux_out=None):
        """Step epilogue, ONE launch: (optionally) loss_out = mean(loss_rows) [and aux_out = mean(aux_rows)] and advance
        the optimizer step / sampler clock / epoch cursor."""
        if aux_rows is not None:
            ops.call("gs_finalize_step2", ops.ptr(loss_rows), n, (1.0 / n) if n else 0.0, ops.ptr(loss_out),
                     1 if accumulate else 0, ops.ptr(aux_rows), (1.0 / n) if n else 0.0, ops.ptr(aux_out),
                     ops.ptr(self.step_dev) if step else None, step,
                     ops.ptr(self.sample_clock_dev) if clock else None, clock,
                     ops.ptr(cursor) if (cursor is not None and cursor_delta) else None, cursor_delta, self.stream)
            return
        ops.call("gs_finalize_step", ops.ptr(loss_rows), n, (1.0 / n) if n else 0.0, ops.ptr(loss_out),
                 1 if accumulate else 0,
                 ops.ptr(self.step_dev) if step else None, step,
                 ops.ptr(self.sample_clock_dev) if clock else None, clock,
                 ops.ptr(cursor) if (cursor is not None and cursor_delta) else None, cursor_delta, self.stream)

    def launch_gather_jobs(self, jobs):
        """Stand-alone gather+mean launches (K2) of a list of gs_gather_desc jobs on the current stream."""
        for j in jobs or ():
            ops.call("gs_gather_mean_fwd", j.X, j.ldx, j.idx, j.n, j.s, j.d, j.self_src, j.ld_self, j.self_idx, j.out,
                     j.ldo, self.stream)

    def sync(self):
        ops.call("gs_stream_sync", self.stream)


_default_engine = None


TILED3_MAX_SLABS = 24


def tiled3_slab_policy(probs, slots, max_slabs=TILED3_MAX_SLABS):
    """Split-K slab counts for ONE launch of gs_dense_wgrad_grouped_tiled3.  probs: (tiles, reduction rows, slab capacity) per
    problem; slots: workgroups of one round (one per CU).  A workgroup holds its CU for its slice's 32-row stages, so the launch is
    cut into ONE round: the smallest stage count L per workgroup with sum(tiles x ceil(stages / L)) <= slots -- the Reddit step: 11
    slices of 512 rows for the two 602 x 128 layer-0 problems, 1 for the 512-row layer-1 / head problems = 233 workgroups (12 + 2
    would be 265 = two rounds: 32 vs 24 us alone, benchmarks/micro_wgrad.py).  A slice holds at most 1024 rows (its row list lives
    in LDS), so L <= 32 and launches beyond slots x 32 stages take more than one round."""
    st = [(n + 31) // 32 for _, n, _ in probs]
    kmin = [(n + 1023) // 1024 for _, n, _ in probs]
    total = sum(t * s_ for (t, _, _), s_ in zip(probs, st))
    L = max(4, (total + slots - 1) // slots)
    # (at most 24 slabs where the slice limit allows: the optimizer launch sums a variable's slabs 24 loads at a time, a 25th is
    #  a second memory round trip of that launch -- RMAT's two 256 x 128 problems took 26)
    while True:
        ks = [int(max(km, min(cap, max_slabs, (s_ + L - 1) // L))) for (t, _, cap), s_, km in zip(probs, st, kmin)]
        if sum(t * k for (t, _, _), k in zip(probs, ks)) <= slots or L >= 32:
            return ks
        L += 1


def get_engine():
    """Process-wide default engine (the analogue of TF's default graph + session)."""
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine()
    return _default_engine


def set_engine(e):
    global _default_engine
    _default_engine = e


def reset_engine():
    set_engine(None)
