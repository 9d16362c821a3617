"""Data parallelism: one process per GPU, mini-batch ROOT nodes sharded across ranks, feature table and
CSR replicated in each GPU's HBM, ONE RCCL all-reduce of the flat gradient buffer per step over xGMI
(SURVEY.md §8e).  torch.distributed (backend "nccl" == RCCL on ROCm) is the transport; the reference
has no distributed layer at all (single device, supervised_train.py:55-59).
"""
import os
import weakref

import numpy as np
import torch


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns (rank, local_rank, world)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("GS_DIST_BACKEND", "nccl")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_order(order, rank, world_size, batch_size):
    """Rank r's slice of the (already shuffled) epoch order.  The global batch of step i is the
    concatenation over ranks of their i-th local batches; every rank gets the same number of full
    local batches (the ragged tail is dropped so the collective never hangs)."""
    order = np.asarray(order)
    n_steps = len(order) // (batch_size * world_size)
    order = order[: n_steps * batch_size * world_size].reshape(n_steps, world_size, batch_size)
    return np.ascontiguousarray(order[:, rank, :].reshape(-1))


def allreduce_sum_(flat, group=None):
    """In-place sum of the flat gradient buffer over all ranks (one collective, 0.9 MB for Reddit-mean)."""
    import torch.distributed as dist
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class GradAllReduce(object):
    """grad_hook for SupervisedGraphsage: sums engine.grads over ranks.  The all-reduce is issued with
    the engine's HIP stream as torch's current stream, so RCCL orders itself after the backward graph
    and the optimizer graph orders itself after RCCL with stream events -- no host synchronisation.
    Adam then applies grad_scale = 1/world_size (mean of per-rank batch-mean gradients = gradient of
    the global-batch mean loss, i.e. single-process semantics of supervised_models.py:95-99)."""

    def __init__(self, engine):
        self.engine = engine
        self.ext_stream = torch.cuda.ExternalStream(engine.stream, device=engine.device)

    def __call__(self, model):
        with torch.cuda.stream(self.ext_stream):
            allreduce_sum_(self.engine.grads)


def _agree(ok, engine):
    """MIN of a 0/1 flag over all ranks (identity without a process group): every rank learns whether ALL succeeded."""
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return int(bool(ok))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                        device=engine.device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item())


class NativeAllReduce(object):
    """grad_hook backed by the C ABI's RCCL binding (gs_comm_*): ONE ncclAllReduce(sum, fp32) of the flat gradient
    buffer, enqueued on the ENGINE stream.  `capturable = True`: the model records it inside the step's hipGraph, so a
    data-parallel step is one graph launch (backward | all-reduce | clip+Adam) and several steps replay per launch.
    The RCCL unique id travels from rank 0 to the other ranks over torch.distributed (plumbing only).

    Construction is failure-safe ACROSS ranks: before any rank enters the collective ncclCommInitRank the ranks agree
    (MIN all-reduce over the bootstrap process group) that RCCL can be bound everywhere and that rank 0 produced an id;
    ncclCommInitRank itself runs under a watchdog (GS_DP_INIT_TIMEOUT_S, default 180 s): a rank whose peers never show
    up raises instead of blocking forever, and make_grad_hook() then lets all ranks fall back together."""
    capturable = True

    def __init__(self, engine, world_size=None, rank=None):
        import ctypes
        import threading
        import torch.distributed as dist
        from . import _lib, ops
        self.engine = engine
        self._comm = None
        self._lib = _lib
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
            rank = dist.get_rank() if dist.is_initialized() else 0
        self.world_size, self.rank = int(world_size), int(rank)
        multi = self.world_size > 1
        # ---- 1. can every rank bind RCCL?  (a collective decision BEFORE the collective init)
        err = None
        try:
            ops.call("gs_comm_available")
        except Exception as ex:
            err = ex
        if multi and not _agree(err is None, engine):
            raise RuntimeError("RCCL cannot be bound on every rank (this rank: %r)" % (err,))
        if err is not None:
            raise RuntimeError("RCCL cannot be bound: %r" % (err,))
        # ---- 2. rank 0's unique id reaches everybody, and everybody agrees that it is valid
        nbytes = 128
        buf = (ctypes.c_uint8 * nbytes)()
        id_error = None
        if self.rank == 0:
            try:
                ops.call("gs_comm_unique_id", ctypes.addressof(buf), nbytes)
            except Exception as ex:      # still take part in the broadcast below (an all-zero id)
                id_error = ex
                buf = (ctypes.c_uint8 * nbytes)()
        if multi:
            dev = engine.device if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=0)
            raw = bytes(t.cpu().tolist())
            buf = (ctypes.c_uint8 * nbytes)(*raw)
        if id_error is not None or not any(bytes(buf)):       # identical on every rank: all raise together
            raise RuntimeError("RCCL unique id unavailable on rank 0: %r" % (id_error,))
        # ---- 3. the collective init under a watchdog
        h = ctypes.c_void_p()
        torch.cuda.set_device(engine.device)
        box = {}

        def init():
            try:
                torch.cuda.set_device(engine.device)          # ncclCommInitRank binds the calling thread's device
                ops.call("gs_comm_init_rank", ctypes.byref(h), self.world_size, self.rank, ctypes.addressof(buf), nbytes)
                box["ok"] = True
            except Exception as ex:
                box["err"] = ex

        timeout = float(os.environ.get("GS_DP_INIT_TIMEOUT_S", "180"))
        th = threading.Thread(target=init, name="gs_comm_init_rank", daemon=True)
        th.start()
        th.join(timeout)
        if th.is_alive():
            # FATAL for this process: the abandoned thread still sits inside ncclCommInitRank with a half-built communicator on
            # this device and may write `h` later; any further collective on the device (the NCCL-backend fallback included)
            # could hang behind it.  make_grad_hook does not fall back from this one.
            raise RcclInitTimeout("ncclCommInitRank did not return within %.0f s on rank %d (a peer failed before entering it?)"
                                  % (timeout, self.rank))
        if "err" in box:
            raise box["err"]
        self._comm = h.value

    def ranks(self):
        """ncclCommCount of the communicator (what RCCL itself says it spans)."""
        import ctypes
        from . import ops
        n = ctypes.c_int32()
        ops.call("gs_comm_count", self._comm, ctypes.byref(n))
        return int(n.value)

    def all_reduce(self, flat, stream=None):
        from . import ops
        ops.call("gs_comm_allreduce_sum_f32", self._comm, ops.ptr(flat), flat.numel(), self.engine.stream if stream is None else stream)

    def __call__(self, model):
        self.all_reduce(self.engine.grads)

    def self_test(self):
        """The all-reduce eagerly and replayed from a captured hipGraph on a scratch buffer: sum of (rank + 1)."""
        from . import ops
        e = self.engine
        x = torch.full((1024,), float(self.rank + 1), device=e.device)
        torch.cuda.synchronize()
        want = float(sum(range(1, self.world_size + 1)))
        self.all_reduce(x)
        e.sync()
        if abs(float(x[0].item()) - want) > 1e-6 or abs(float(x[-1].item()) - want) > 1e-6:
            raise RuntimeError("eager RCCL all-reduce gave %r, expected %r" % (float(x[0].item()), want))
        g = ops.Graph(e.stream)
        g.begin()
        try:
            self.all_reduce(x)
        finally:
            g.end()
        x.fill_(float(self.rank + 1))
        torch.cuda.synchronize()
        g.launch()
        e.sync()
        if abs(float(x[5].item()) - want) > 1e-6:
            raise RuntimeError("graph-replayed RCCL all-reduce gave %r, expected %r" % (float(x[5].item()), want))
        return True

    def close(self):
        if self._comm:
            self._lib.load().gs_comm_destroy(self._comm)
            self._comm = None


class PeerPushAllReduce(object):
    """grad_hook backed by the C ABI's peer-store exchange (gs_peer_*, csrc/gs_peer.hip; GS_DP_PEER_PUSH=1): every rank
    stores slice p of its flat gradient straight into rank p's window over xGMI, rank p sums the copies in rank order and
    stores the sum into every window -- one hop out, one hop back, ONE kernel launch on the engine stream instead of a
    2(N-1)-hop ring for 0.9 MB.  `capturable = True`: recorded inside the step's hipGraph exactly like NativeAllReduce.
    The 64-byte IPC handles travel over torch.distributed (plumbing only).  Every device-side wait is bounded; `check()`
    (called by the models when results are fetched) turns a tripped wait into an exception."""
    capturable = True

    def __init__(self, engine, chunks=None, spin_limit=None):
        import torch.distributed as dist
        self.engine = engine
        self._peer = None
        multi = dist.is_initialized() and dist.get_world_size() > 1
        self.world_size = dist.get_world_size() if multi else 1
        self.rank = dist.get_rank() if multi else 0
        self.chunks = 0 if chunks is None else int(chunks)                    # 0: the library's default chunking
        self.spin_limit = 0 if spin_limit is None else int(spin_limit)        # 0: the library's default bound (2^24 polls)
        self.attempts = 0
        self.failures = []          # this rank's failed self-test attempts (diagnostics)
        self._open()

    # One window per (shape, rank) for the PROCESS'S LIFETIME: a window is never freed while a peer may still map it, and a
    # re-created hook (a second model in the same process, bench.py's legs) re-uses the window AND the peers' mappings of it.
    # Round 4 freed and re-allocated windows per hook: a peer's hipIpcOpenMemHandle could then return its stale mapping of the
    # old allocation at the same address (5 of 30 re-created hooks timed out on their first exchange) and the self test retried
    # on fresh windows -- a workaround; with the cache no mapping is ever re-made.  The epoch / flag state in the window simply
    # continues (all ranks continue in lockstep); a window whose sticky error word is set is dropped from the cache.
    _windows = {}
    _claimed = {}               # cache key -> weakref of the LIVE hook using that window (one user per window: its epoch / flag state)

    def _cache_key(self):
        return (int(self.engine.grads.numel()), self.world_size, self.rank, self.chunks, self.spin_limit, str(self.engine.device))

    def _open(self):
        """Allocate this rank's window, exchange IPC handles, map every peer (collective; every stage's outcome is agreed)."""
        import ctypes
        import torch.distributed as dist
        from . import _lib, ops
        engine = self.engine
        multi = self.world_size > 1
        torch.cuda.set_device(engine.device)
        key = self._cache_key()
        owner = PeerPushAllReduce._claimed.get(key)
        owner = owner() if owner is not None else None
        if owner is not None and owner is not self and owner._peer and not getattr(self, "_fresh", False):
            # two hooks alive at once with equal parameter counts would share one window's epoch / flag words across two
            # engine streams -- a race if both exchange concurrently.  Refuse: close() the other hook first.
            raise RuntimeError("a live PeerPushAllReduce hook of this process already uses the exchange window for %r; "
                               "close() it before creating another" % (key,))
        cached = PeerPushAllReduce._windows.get(key) if not getattr(self, "_fresh", False) else None
        if cached is not None and self._window_errored(cached):
            # a sticky error word from an earlier run: every later exchange on it would return early -- never re-use it
            # (left allocated: a peer may still map it)
            PeerPushAllReduce._windows.pop(key, None)
            cached = None
        if (not multi and cached) or (multi and _agree(cached is not None, engine)):
            self._peer = cached                    # every rank still holds its window and its mappings of the peers' windows
            self.reused = True
            PeerPushAllReduce._claimed[key] = weakref.ref(self)
            return
        self.reused = False
        h = ctypes.c_void_p()
        err = None
        try:
            ops.call("gs_peer_create", engine.grads.numel(), self.world_size, self.rank, self.chunks, self.spin_limit, ctypes.byref(h))
            self._peer = h.value
        except Exception as ex:
            err = ex
        if multi and not _agree(err is None, engine):
            self.close()
            raise RuntimeError("the exchange window could not be allocated on every rank (this rank: %r)" % (err,))
        if err is not None:
            raise err
        if multi:
            nb = _lib.GS_PEER_HANDLE_BYTES
            buf = (ctypes.c_uint8 * nb)()
            try:
                ops.call("gs_peer_export", self._peer, ctypes.addressof(buf), nb)
            except Exception as ex:       # still take part in the gather below
                err = ex
            dev = engine.device if dist.get_backend() == "nccl" else torch.device("cpu")
            mine = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=dev)
            got = [torch.empty_like(mine) for _ in range(self.world_size)]
            dist.all_gather(got, mine)
            if err is None:
                try:
                    for r, t in enumerate(got):
                        if r != self.rank:
                            raw = (ctypes.c_uint8 * nb)(*bytes(t.cpu().tolist()))
                            ops.call("gs_peer_attach", self._peer, r, ctypes.addressof(raw), nb)
                except Exception as ex:
                    err = ex
            if not _agree(err is None, engine):       # nobody launches the exchange unless everybody mapped everybody
                self.close()
                raise RuntimeError("peer windows could not be mapped on every rank (this rank: %r)" % (err,))
        PeerPushAllReduce._windows[key] = self._peer
        PeerPushAllReduce._claimed[key] = weakref.ref(self)

    @staticmethod
    def _window_errored(peer):
        import ctypes
        from . import ops
        ep, er = ctypes.c_int64(), ctypes.c_int32()
        try:
            ops.call("gs_peer_status", peer, ctypes.byref(ep), ctypes.byref(er))
        except Exception:
            return True
        return int(er.value) != 0

    def all_reduce(self, flat, stream=None):
        from . import ops
        ops.call("gs_peer_allreduce_sum_f32", self._peer, ops.ptr(flat), flat.numel(), self.engine.stream if stream is None else stream)

    def __call__(self, model):
        self.all_reduce(self.engine.grads)

    def status(self):
        """(exchanges completed on this rank, error word) -- call after the engine stream has been synchronised."""
        import ctypes
        from . import ops
        ep, er = ctypes.c_int64(), ctypes.c_int32()
        ops.call("gs_peer_status", self._peer, ctypes.byref(ep), ctypes.byref(er))
        return int(ep.value), int(er.value)

    def check(self):
        ep, er = self.status()
        if er:
            # the error word is sticky: this window is finished -- a later hook of this process must not be handed it
            if PeerPushAllReduce._windows.get(self._cache_key()) == self._peer:
                PeerPushAllReduce._windows.pop(self._cache_key(), None)
            raise RuntimeError("peer exchange on rank %d gave up waiting for rank(s) %r (stage bits %d, %d exchanges): the "
                               "gradients of this step are not the sum over ranks"
                               % (self.rank, [r for r in range(self.world_size) if (er >> (8 + r)) & 1], er & 3, ep))
        return ep

    def ranks(self):
        return self.world_size

    def self_test(self, attempts=3):
        """One exchange on the gradient buffer itself (zeroed afterwards): sum of (rank + 1), and the error word.  Collective.
        Why attempts: when a process frees a window and allocates the next one at the SAME address (a second hook in one
        process), a peer's hipIpcOpenMemHandle can hand back its mapping of the OLD allocation -- the peer's stores then land
        in memory nobody polls and the first exchange times out (measured with two processes on one device: 5 of 30
        re-created hooks, 0 of 30 when no window is ever freed, benchmarks/peer_exchange_2proc.py PEER_KEEP_WINDOWS=1).  A
        failed attempt is agreed on by all ranks and the windows are re-created WHILE THE OLD ONES STILL EXIST (new
        addresses, new handles).  One hook per process -- the deployment -- never takes the second attempt."""
        from . import ops
        e = self.engine
        want = float(sum(range(1, self.world_size + 1)))
        last = None
        for attempt in range(attempts):
            self.attempts = attempt + 1
            e.grads.fill_(float(self.rank + 1))
            torch.cuda.synchronize()
            _agree(True, e)      # a barrier: first-use costs (kernel loading) must not eat into the exchange's bounded wait
            self.all_reduce(e.grads)
            e.sync()
            got = (float(e.grads[0].item()), float(e.grads[-1].item()))
            e.grads.zero_()
            torch.cuda.synchronize()
            ok = True
            try:
                self.check()
                if abs(got[0] - want) > 1e-6 or abs(got[1] - want) > 1e-6:
                    raise RuntimeError("peer exchange gave %r, expected %r" % (got, want))
            except RuntimeError as ex:
                ok, last = False, ex
                self.failures.append(repr(ex))
            if _agree(ok, e):
                return True
            old, self._peer = self._peer, None      # the new windows are allocated while the old ones still exist: new addresses,
            PeerPushAllReduce._windows.pop(self._cache_key(), None)
            self._fresh = True
            try:
                self._open()                        # new IPC handles
            finally:
                self._fresh = False
                ops.call("gs_peer_destroy", old)    # also when _open() raises (the caller then falls back to RCCL): no leaked window
        raise RuntimeError("peer exchange failed its self test %d times (this rank's last error: %r)" % (attempts, last))

    def close(self):
        """Detach from the window.  A cached window (the normal case) stays allocated and mapped for the process's lifetime
        (see _windows); only a window that is not in the cache -- a failed open -- is freed here."""
        if self._peer:
            from . import ops
            peer, self._peer = self._peer, None
            PeerPushAllReduce._claimed.pop(self._cache_key(), None)
            if PeerPushAllReduce._windows.get(self._cache_key()) != peer:
                # not (or no longer) in the cache: a failed open frees its window; an ERRORED window stays allocated for
                # the process's lifetime (a peer may still map it) and is simply never handed out again
                if not self._window_errored(peer):
                    ops.call("gs_peer_destroy", peer)


class SpinHook(object):
    """Diagnostics: a grad_hook that holds the engine stream for `us` microseconds with ONE sleeping wave (gs_spin_us) --
    the single-GPU stand-in for a latency-bound all-reduce when the data-parallel step schedule is probed without peers
    (bench.py GS_PROBE_DP_SCHEDULE=<us>; us = 0: a no-op hook, the schedule's own cost).  Capturable like NativeAllReduce."""
    capturable = True

    def __init__(self, engine, us=0.0):
        self.engine, self.us = engine, float(us)

    def __call__(self, model):
        if self.us > 0:
            from . import ops
            ops.call("gs_spin_us", self.us, self.engine.stream)


class RcclInitTimeout(RuntimeError):
    """ncclCommInitRank hung past the watchdog: the process must not issue further GPU collectives."""


def make_grad_hook(engine, log=None):
    """The gradient all-reduce hook of a data-parallel run: the in-graph RCCL binding of the C ABI when it initialises
    and passes its self test on EVERY rank (each stage's outcome is agreed collectively, so no rank is left inside a
    collective its peers never enter), else the eager torch.distributed all-reduce between two graphs (GradAllReduce).
    GS_DP_NATIVE=0 forces the fallback; GS_DP_PEER_PUSH=1 tries the peer-store exchange (PeerPushAllReduce) first."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    hook, ok = None, 0
    if os.environ.get("GS_DP_PEER_PUSH", "0") == "1":
        # opt-in: direct peer stores over xGMI (gs_peer.hip).  Construction agrees collectively at every stage; the self
        # test's waits are bounded on the device, so a failing rank cannot leave its peers blocked.
        try:
            hook = PeerPushAllReduce(engine)
            ok = 1
        except Exception as ex:
            if log:
                log("peer-push hook unavailable on rank %d: %r" % (rank, ex))
        ok = _agree(ok, engine)
        if ok:
            try:
                hook.self_test()
            except Exception as ex:
                ok = 0
                if log:
                    log("peer-push hook failed its self test on rank %d: %r" % (rank, ex))
            ok = _agree(ok, engine)
        if ok:
            return hook
        if hook is not None:
            hook.close()
        hook, ok = None, 0
    if os.environ.get("GS_DP_NATIVE", "1") == "1":
        try:
            hook = NativeAllReduce(engine)
            ok = 1
        except RcclInitTimeout as ex:
            # a bootstrap abandoned in flight: falling back to another collective on the same device is only safe when that
            # collective runs on the host (gloo); otherwise stop this rank -- its peers time out in _agree
            if log:
                log("FATAL on rank %d: %r" % (rank, ex))
            if not (dist.is_initialized() and dist.get_backend() == "gloo"):
                raise
        except Exception as ex:      # RCCL missing / id / init failure
            if log:
                log("native RCCL hook unavailable on rank %d: %r" % (rank, ex))
        ok = _agree(ok, engine)      # every rank constructed it -- only then is the (collective) self test entered
        if ok:
            try:
                hook.self_test()
            except Exception as ex:
                ok = 0
                if log:
                    log("native RCCL hook failed its self test on rank %d: %r" % (rank, ex))
            ok = _agree(ok, engine)
    if ok:
        return hook
    if hook is not None:
        hook.close()
    return GradAllReduce(engine)
