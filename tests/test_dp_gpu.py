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
    res = dict((r, (w, s)) for r, w, s in [q.get(timeout=240) for _ in range(world)])
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
