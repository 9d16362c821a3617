"""Data parallelism: one process per GPU, mini-batch ROOT nodes sharded across ranks, feature table and
CSR replicated in each GPU's HBM, ONE RCCL all-reduce of the flat gradient buffer per step over xGMI
(SURVEY.md §8e).  torch.distributed (backend "nccl" == RCCL on ROCm) is the transport; the reference
has no distributed layer at all (single device, supervised_train.py:55-59).
"""
import os

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


def make_grad_hook(engine):
    """The gradient all-reduce hook of a data-parallel run (see GradAllReduce)."""
    return GradAllReduce(engine)
