"""-m gpu: data-parallel training on 2 ranks (gloo transport, both ranks on cuda:0 -- RCCL refuses two ranks per
device, the 8-GPU RCCL run is the driver's) must reproduce single-process training on the concatenated global
batch: same sampled neighbor sets (the sampler is keyed by the GLOBAL row), same parameters after several steps."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, B = 4, 32


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _build(world, rank):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graphsage_amd import distributed as gsd
    from graphsage_amd import engine as eng
    from test_model_gpu import build
    G, it, ph, sampler, model, ns = build(torch.device("cuda:0"), "mean", True, False, csr=True)
    model.world_size, model.rank = world, rank
    model.row_offset = rank * B
    order = it.train_nodes[: STEPS * B * 2]
    return gsd, eng, G, it, model, order, ns


def _collect(procs, q, n, timeout):
    """n results from the workers' queue; stops as soon as a worker has died (a crashed rank must not cost the full timeout)."""
    import queue
    import time
    got, t0 = [], time.time()
    while len(got) < n:
        try:
            got.append(q.get(timeout=1.0))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > timeout:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise RuntimeError("worker exit codes %r after %.0f s" % ([p.exitcode for p in procs], time.time() - t0))
    return got


def _worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": "0", "GS_DIST_BACKEND": "gloo"})
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)
    sys.path.insert(0, ROOT)
    from graphsage_amd import distributed as gsd0
    gsd0.init_from_env()
    gsd, eng, G, it, model, order, ns = _build(world, rank)
    e = eng.get_engine()
    model.attach_device_epoch(gsd.shard_order(order, rank, world, B), it.label_matrix)
    model.grad_hook = gsd.GradAllReduce(e)
    for _ in range(STEPS):
        model.train_step_device(B)
    e.sync()
    torch.cuda.synchronize()
    q.put((rank, e.params.cpu().numpy().copy(), model.samples1[2].cpu().numpy().copy()))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dp_matches_single_process_global_batch(dev):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (w, s)) for r, w, s in _collect(procs, q, world, 240))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][0], res[1][0])                       # replicas stay identical
    # single process, global batch 2B, same epoch order
    gsd, eng, G, it, model, order, ns = _build(1, 0)
    model.attach_device_epoch(order, it.label_matrix)
    for _ in range(STEPS):
        model.train_step_device(2 * B)
    eng.get_engine().sync()
    single = eng.get_engine().params.cpu().numpy()
    s2 = model.samples1[2].cpu().numpy()
    per = s2.size // (2 * B)
    assert np.array_equal(s2[: B * per], res[0][1]) and np.array_equal(s2[B * per:], res[1][1])   # same neighbor sets
    np.testing.assert_allclose(res[0][0], single, rtol=2e-4, atol=2e-6)


def test_native_rccl_hook_in_graph_single_rank(dev):
    """The C ABI's RCCL binding (gs_comm_*) with one rank: unique id -> ncclCommInitRank -> ncclAllReduce enqueued on the
    engine stream, eagerly and replayed from a hipGraph; then the data-parallel step schedule with the all-reduce
    recorded INSIDE the step graph (backward | all-reduce | clip+Adam, several steps per launch) must reproduce the
    single-GPU fused schedule (world = 1: the sum is the identity, grad_scale = 1)."""
    import numpy as np
    from graphsage_amd import engine as eng
    from graphsage_amd.distributed import NativeAllReduce
    from test_model_gpu import build
    outs = []
    for dp in (False, True):
        G, it, ph, sampler, model, ns = build(dev, "mean", True, False, csr=True)
        if dp:
            hook = NativeAllReduce(eng.get_engine(), world_size=1, rank=0)
            assert hook.self_test()
            model.grad_hook = hook
            assert model._dp_in_graph()
        model.attach_device_epoch(it.train_nodes[:320], it.label_matrix)
        model.train_steps_device(32, 9, steps_per_launch=2)
        loss, preds = model._fetch(32)
        outs.append((loss, preds.copy(), eng.get_engine().params.cpu().numpy().copy()))
        if dp:
            assert any(k[0] == "ptrain_dp" and k[3] == 2 for k in model._graphs), list(model._graphs)   # 2-step DP graphs
            hook.close()
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5)
    np.testing.assert_allclose(outs[0][2], outs[1][2], rtol=1e-5, atol=1e-7)


def test_dp_in_graph_schedule_single_rank_matches_and_unsup_in_graph(dev):
    """(a) supervised: the data-parallel in-graph schedule (backward | all-reduce | clip + Adam in ONE graph; here the 1-rank
    RCCL all-reduce, and a sleeping-wave stand-in for a slow collective) gives the same parameters as the single-GPU fused
    schedule; (b) unsupervised: the same for SampleAndAggregate, several steps per launch."""
    import numpy as np
    from graphsage_amd import engine as eng
    from graphsage_amd.distributed import NativeAllReduce, SpinHook
    from test_model_gpu import build
    outs = []
    for mode in ("single", "rccl", "spin"):
        G, it, ph, sampler, model, ns = build(dev, "mean", True, False, csr=True)
        hook = None
        if mode != "single":
            hook = NativeAllReduce(eng.get_engine(), world_size=1, rank=0) if mode == "rccl" else SpinHook(eng.get_engine(), 20.0)
            model.grad_hook = hook
            assert model._dp_in_graph()
            if mode == "rccl":
                info = model.measure_dp_allreduce()
                assert info["allreduce_us_standalone"] > 0 and hook.ranks() == 1
        model.attach_device_epoch(it.train_nodes[:320], it.label_matrix)
        model.train_steps_device(32, 9, steps_per_launch=2)
        loss, preds = model._fetch(32)
        outs.append((loss, eng.get_engine().params.cpu().numpy().copy()))
        if hook is not None and hasattr(hook, "close"):
            hook.close()
    for o in outs[1:]:
        np.testing.assert_allclose(outs[0][0], o[0], rtol=1e-5)
        np.testing.assert_allclose(outs[0][1], o[1], rtol=1e-5, atol=1e-7)
    assert np.array_equal(outs[1][1], outs[2][1])               # the collective's duration does not change the bits
    # ---- a model WITHOUT the fused tail (gcn): the step epilogue runs ahead of its backward pass, the sampler rides in the
    #      reduce launch in front of the collective
    gouts = []
    for dp in (False, True):
        G, it, ph, sampler, model, ns = build(dev, "gcn", False, False, csr=True)
        if dp:
            hook = NativeAllReduce(eng.get_engine(), world_size=1, rank=0)
            model.grad_hook = hook
            assert model._dp_in_graph()
        model.attach_device_epoch(it.train_nodes[:320], it.label_matrix)
        model.train_steps_device(32, 9, steps_per_launch=2)
        loss, preds = model._fetch(32)
        gouts.append((loss, eng.get_engine().params.cpu().numpy().copy()))
        if dp:
            assert any(k[0] == "ptrain_dp" and k[3] == 2 for k in model._graphs), list(model._graphs)
            hook.close()
    np.testing.assert_allclose(gouts[0][0], gouts[1][0], rtol=1e-5)
    np.testing.assert_allclose(gouts[0][1], gouts[1][1], rtol=1e-5, atol=1e-7)
    # ---- unsupervised
    from test_unsup_gpu import build as build_unsup
    res = []
    for dp in (False, True):
        G, it, ph, sampler, model, ns = build_unsup("mean", True, csr=True)
        if dp:
            hook = NativeAllReduce(eng.get_engine(), world_size=1, rank=0)
            model.grad_hook = hook
        model.attach_device_pairs(it.train_edges[:320])
        model.train_steps_device(32, 9, steps_per_launch=2)
        eng.get_engine().sync()
        res.append(eng.get_engine().params.cpu().numpy().copy())
        if dp:
            assert any(k[0] == "updtrain_dp" and k[2] == 2 for k in model._graphs), list(model._graphs)
            hook.close()
    np.testing.assert_allclose(res[0], res[1], rtol=1e-5, atol=1e-7)


def _rccl_worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank), "GS_DIST_BACKEND": "nccl", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.cuda.set_device(rank)
    from graphsage_amd import distributed as gsd
    from graphsage_amd import engine as eng
    from test_model_gpu import build
    gsd.init_from_env()
    out = {}
    for mode in ("native", "eager"):
        os.environ["GS_DP_NATIVE"] = "1" if mode == "native" else "0"
        G, it, ph, sampler, model, ns = build(torch.device("cuda:%d" % rank), "mean", True, False, csr=True)
        model.world_size, model.rank = world, rank
        model.row_offset = rank * B
        order = it.train_nodes[: 8 * B * 2]
        e = eng.get_engine()
        model.grad_hook = gsd.make_grad_hook(e)
        if mode == "native":
            assert type(model.grad_hook).__name__ == "NativeAllReduce" and model.grad_hook.ranks() == world
            model.measure_dp_allreduce()
        model.attach_device_epoch(gsd.shard_order(order, rank, world, B), it.label_matrix)
        model.train_steps_device(B, 7, steps_per_launch=2)
        e.sync()
        torch.cuda.synchronize()
        out[mode] = e.params.cpu().numpy().copy()
        if mode == "native":
            assert any(k[0] == "ptrain_dp" for k in model._graphs)
    q.put((rank, out["native"], out["eager"]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_rccl_in_graph_matches_eager_hook():
    """TWO GPUs, real RCCL: the in-graph schedule (ncclAllReduce recorded in the step hipGraph, gather share forked
    beside it, 2 steps per launch) == the eager torch.distributed hook between two graphs; replicas stay identical.
    Auto-skips on a 1-GPU box (the driver's multi-GPU run is the first to execute it)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (a, b)) for r, a, b in _collect(procs, q, world, 300))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])      # replicas identical
    np.testing.assert_allclose(res[0][0], res[0][1], rtol=1e-5, atol=1e-7)                    # in-graph == eager hook


def test_bench_gpus_2_launches_its_own_ranks(dev):
    """`python bench.py --gpus 2` WITHOUT a launcher starts its two ranks itself (torch.distributed.run contract) and prints
    ONE line with n_gpus == 2; ranks share cuda:0 over the gloo bootstrap (a functional check of the N > 1 bench path: sharded
    epoch order, gradient hook, max-over-ranks timing).  A WORLD_SIZE that disagrees with --gpus is refused, not reported."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"GS_DIST_BACKEND": "gloo", "GS_DP_NATIVE": "0", "GS_FAULT_DUMP_S": "280"})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--nodes", "20000",
           "--avg_degree", "40", "--no-cpu-baseline", "--no-aux"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 1024
    assert d["config"]["allreduce"] == "GradAllReduce" and d["value"] > 0 and d["ms_per_step_events"]["launches"] >= 10
    # the mismatch is an error, not a line with the wrong n_gpus
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stderr + r2.stdout) and not [l for l in r2.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("agg", ["mean", "maxpool"])
def test_peer_exchange_in_graph_single_rank_equals_local_adam(dev, agg):
    """The in-graph data-parallel schedule (slab sum | exchange | clip + Adam: three launches) with the peer-store exchange and a
    world of one rank == the single-GPU fused optimizer launch, bit for bit, through multi-step hipGraphs with the sampler riding
    in the slab-sum launch.  (Round 5's one-launch form of the three, gs_peer_step, lost its measurement -- 129.7 vs 115.4
    us/step before any peer latency -- and was removed in round 6: benchmarks/variants/README.md.)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graphsage_amd import distributed as gsd
    from graphsage_amd import engine as eng
    from test_model_gpu import build
    outs = []
    for mode in ("local", "peer"):
        G, it, ph, sampler, model, ns = build(torch.device("cuda:0"), agg, True, False, csr=True, wd=0.01)
        e = eng.get_engine()
        if mode != "local":
            hook = gsd.PeerPushAllReduce(e)
            assert hook.self_test()
            model.grad_hook = hook
            assert model._dp_in_graph()
        model.attach_device_epoch(it.train_nodes[: 8 * B], it.label_matrix)
        model.train_steps_device(B, 7, steps_per_launch=2)
        loss, preds = model._fetch(B)
        outs.append((loss, e.params.cpu().numpy().copy(), e.adam_v.cpu().numpy().copy()))
        if mode != "local":
            assert hook.check() >= 7
            hook.close()
    for o in outs[1:]:
        assert o[0] == outs[0][0]
        assert np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])


@pytest.mark.parametrize("hook_kind", ["peer", "rccl"])
def test_unsup_host_fed_train_step_applies_one_update_per_step_under_capturable_hooks(dev, hook_kind):
    """SampleAndAggregate.train_step(feed_dict) (unsupervised_train.py:273-274) under a hook that can be recorded in the step
    graph: backward | exchange | clip + Adam are ONE hipGraph and run ONCE per step, as on the device-epoch path.  (Round 5's
    review found the host-fed path applying the exchange and the optimizer twice under the one-launch peer step, since
    removed.)  World of one rank: parameters, Adam state and the step counter equal the hook-free model's."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graphsage_amd import distributed as gsd
    from graphsage_amd import engine as eng
    from test_unsup_gpu import build as build_unsup
    outs = []
    for mode in ("local", hook_kind):
        G, it, ph, sampler, model, ns = build_unsup("mean", True, csr=True, wd=0.01)
        e = eng.get_engine()
        hook = None
        if mode == "rccl":
            hook = gsd.NativeAllReduce(e, world_size=1, rank=0)
        elif mode != "local":
            hook = gsd.PeerPushAllReduce(e)
            assert hook.self_test()
        if hook is not None:
            model.grad_hook = hook
            assert model._dp_in_graph()
        edges = it.train_edges[:32 * 4]
        for i in range(4):                      # eager, captured, replayed, replayed
            b = edges[32 * i: 32 * (i + 1)]
            model.train_step({ph['batch1']: b[:, 0], ph['batch2']: b[:, 1], ph['batch_size']: 32}, fetch=(i == 3))
        e.sync()
        outs.append((e.params.cpu().numpy().copy(), e.adam_v.cpu().numpy().copy(), int(e.step_dev.item())))
        if hook is not None:
            assert any(k[0] == "utrain_dp" for k in model._graphs), list(model._graphs)
            if hasattr(hook, "check"):
                hook.check()
            hook.close()
    assert outs[0][2] == outs[1][2] == 4
    if hook_kind == "rccl":
        np.testing.assert_allclose(outs[1][0], outs[0][0], rtol=1e-6, atol=1e-8)
    else:
        assert np.array_equal(outs[1][0], outs[0][0]) and np.array_equal(outs[1][1], outs[0][1])


def test_peer_window_has_one_live_user_and_an_errored_window_is_never_reused(dev):
    """Process-lifetime exchange windows (PeerPushAllReduce._windows) are keyed by shape: a second LIVE hook with the same key
    is refused (it would share the first one's epoch / flag words across two streams); after close() the window is handed on."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graphsage_amd import distributed as gsd
    from graphsage_amd import engine as eng
    from test_model_gpu import build
    build(torch.device("cuda:0"), "mean", True, False, csr=True)
    e = eng.get_engine()
    a = gsd.PeerPushAllReduce(e)
    assert a.self_test()
    with pytest.raises(RuntimeError, match="already uses the exchange window"):
        gsd.PeerPushAllReduce(e)
    win = a._peer
    a.close()
    b = gsd.PeerPushAllReduce(e)
    assert b.reused and b._peer == win and b.self_test()
    b.close()
