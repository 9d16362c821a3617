"""World-size-2 gloo test of the data-parallel math (runs on CPU): summing the per-rank gradients with ONE
all-reduce of the flat buffer and scaling by 1/world_size reproduces the single-process gradient of the
concatenated global batch (oracle), which is what SupervisedGraphsage's grad_hook + Adam(grad_scale) rely on."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "GS_DIST_BACKEND": "gloo"})
    import torch.distributed as dist
    from graphsage_amd import distributed as gsd
    from oracle import graphsage_oracle as orc
    r, _, w = gsd.init_from_env()
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(0)                       # same data on every rank (replicated graph)
    N, F, C, B = 200, 12, 4, 8
    feat = np.vstack([rng.normal(size=(N, F)), np.zeros((1, F))]).astype(np.float32)
    neigh = [list(rng.choice(N, size=rng.integers(1, 9), replace=False)) for _ in range(N)]
    adj, _ = orc.construct_adj(neigh, 8, rng)
    dims, ns = [F, 8, 8], [3, 2]
    params = orc.make_supervised_params("mean", dims, C, True, rng)
    order = rng.permutation(N)
    perms = [rng.permutation(8), rng.permutation(8)]
    labels_all = np.eye(C, dtype=np.float32)[rng.integers(0, C, N)]

    def grads_of(batch):
        samples, ss = orc.sample(adj, batch, ns, perms)
        res = orc.supervised_fwd_bwd(params, feat, samples, ss, labels_all[batch], dims, ns, len(batch), "mean", True, False)
        return np.concatenate([g.reshape(-1) for _, g in orc.flat_param_items(res["grads"], "mean")])

    mine = gsd.shard_order(order, rank, world, B)[:B]
    flat = torch.from_numpy(grads_of(mine).copy())
    gsd.allreduce_sum_(flat)                              # ONE collective on the flat buffer
    flat /= world                                         # Adam's grad_scale
    want = grads_of(order[: B * world])                   # single-process gradient of the global batch
    q.put((rank, float(np.abs(flat.numpy() - want).max()), float(np.abs(want).max())))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_gradient_equals_global_batch_gradient():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, scale in res:
        assert err < 1e-5 * max(1.0, scale), (rank, err, scale)


def _agree_worker(rank, world, port, q):
    """NativeAllReduce's construction with a rank that cannot bind RCCL: the ranks must AGREE (before anyone enters the
    collective ncclCommInitRank) and all raise promptly -- nobody hangs."""
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "GS_DIST_BACKEND": "gloo"})
    import time
    import torch.distributed as dist
    from graphsage_amd import distributed as gsd
    from graphsage_amd import ops
    gsd.init_from_env()
    real_call = ops.call
    calls = []

    def fake_call(name, *args):
        calls.append(name)
        if name == "gs_comm_available" and rank == 1:
            raise ops._lib.GraphsageAmdError("RCCL not found (test)")
        if name == "gs_comm_init_rank":
            raise AssertionError("a rank entered ncclCommInitRank although a peer cannot bind RCCL")
        return real_call(name, *args)

    ops.call = fake_call

    class FakeEngine(object):
        device = torch.device("cpu")

    t0 = time.time()
    try:
        gsd.NativeAllReduce(FakeEngine())
        outcome = "constructed"
    except RuntimeError as ex:
        outcome = "raised: %s" % ex
    # the collective helper itself: MIN over ranks
    assert gsd._agree(rank == 0, FakeEngine()) == 0 and gsd._agree(True, FakeEngine()) == 1
    q.put((rank, outcome, time.time() - t0, "gs_comm_init_rank" in calls))
    dist.barrier()
    dist.destroy_process_group()


def test_native_allreduce_init_is_failure_safe_across_ranks():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outcome, dt, entered_init in res:
        assert outcome.startswith("raised: RCCL cannot be bound on every rank"), (rank, outcome)
        assert not entered_init and dt < 60


def _peer_worker(rank, world, port, q):
    """PeerPushAllReduce's construction when (a) one rank cannot allocate its window, (b) one rank cannot map a peer's window:
    every stage's outcome is agreed over the bootstrap group, all ranks raise together, windows that were created are
    destroyed, and nobody launches an exchange."""
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "GS_DIST_BACKEND": "gloo"})
    import torch.distributed as dist
    from graphsage_amd import distributed as gsd
    from graphsage_amd import ops
    gsd.init_from_env()
    torch.cuda.set_device = lambda d: None          # no GPU here: the device binding is not what is under test
    calls = []
    scenario = {"name": None}

    def fake_call(name, *args):
        if name == "gs_peer_status":                    # the cached window's sticky error word is read before it is handed on
            args[1]._obj.value, args[2]._obj.value = 0, 0
            return 0
        calls.append(name)
        if name == "gs_peer_create":
            if scenario["name"] == "create" and rank == 1:
                raise ops._lib.GraphsageAmdError("out of device memory (test)")
            args[-1]._obj.value = 0x1000 + len(calls)          # an opaque handle
            return 0
        if name == "gs_peer_attach" and scenario["name"] == "attach" and rank == 0:
            raise ops._lib.GraphsageAmdError("hipIpcOpenMemHandle failed (test)")
        if name in ("gs_peer_export", "gs_peer_attach", "gs_peer_destroy"):
            return 0
        raise AssertionError("unexpected C call %s" % name)

    ops.call = fake_call

    class FakeEngine(object):
        device = torch.device("cpu")
        grads = torch.zeros(1000)

    out = {}
    for name in ("create", "attach", "ok"):
        scenario["name"] = name
        del calls[:]
        try:
            hook = gsd.PeerPushAllReduce(FakeEngine())
            out[name] = ("constructed", list(calls))
            hook.close()
            # process-lifetime windows (round 5): a window that opened on every rank stays allocated and mapped -- close()
            # detaches only -- and a re-created hook of the same shape re-uses it without a single C call (but the status read)
            assert "gs_peer_destroy" not in calls
            n_calls = len(calls)
            again = gsd.PeerPushAllReduce(FakeEngine())
            assert again.reused and len(calls) == n_calls
            again.close()
            assert "gs_peer_destroy" not in calls
        except RuntimeError as ex:
            out[name] = ("raised: %s" % ex, list(calls))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_peer_push_hook_construction_is_failure_safe_across_ranks():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_peer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        out = res[rank]
        assert out["create"][0].startswith("raised: the exchange window could not be allocated on every rank"), out["create"]
        assert "gs_peer_export" not in out["create"][1] and "gs_peer_allreduce_sum_f32" not in out["create"][1]
        # the rank whose window was created gives it back
        assert ("gs_peer_destroy" in out["create"][1]) == (rank == 0), out["create"]
        assert out["attach"][0].startswith("raised: peer windows could not be mapped on every rank"), out["attach"]
        assert out["attach"][1][-1] == "gs_peer_destroy", out["attach"]
        assert out["ok"][0] == "constructed" and out["ok"][1].count("gs_peer_attach") == world - 1, out["ok"]

