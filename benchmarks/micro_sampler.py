"""Diagnostic: the fan-out sampler launch of the unsupervised configuration (1044 roots = 512 pairs + 20 negatives, 10 x 25)
stand-alone, with and without the in-launch root staging, next to the supervised configuration's (512 roots + labels).
    python benchmarks/micro_sampler.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graphsage_amd import ops  # noqa: E402
from graphsage_amd.utils import random_walk_pairs_device, reddit_shaped_device  # noqa: E402
from benchmarks.micro_stream import timeit  # noqa: E402


def main():
    args = bench.parse_args(["--unsupervised"])
    dev = torch.device("cuda:0")
    B = args.batch_size
    DG = reddit_shaped_device(dev, n_nodes=args.nodes, feat_dim=args.feat_dim, num_classes=args.classes,
                              avg_degree=args.avg_degree, seed=123, feat_signal=args.feat_signal)
    res = {}
    e, model, ph, adj_info = bench.build_model(DG, args, 1, 0, args.model, True)
    pairs = random_walk_pairs_device(DG.train_csr[0], DG.train_csr[1], DG.train_nodes, max_pairs=2000000, seed=123).cpu().numpy()
    model.attach_device_pairs(pairs)
    s = e.stream
    roots, n_roots = model._roots(B, parity=0)
    stage = ("unsup", model._pairs, model._cursor, B, model._neg_cdf, model._neg_guide, model._guide_bits, model.neg_sample_size,
             model.neg_seed)
    res["unsup_staged_in_launch_us"] = timeit(lambda: model._sample_phase(roots, n_roots, 0, stage=stage), s)
    model._stage_negatives(roots, B, pairs=model._pairs, cursor=model._cursor)
    res["unsup_roots_prestaged_us"] = timeit(lambda: model._sample_phase(roots, n_roots, 0), s)
    res["unsup_stage_kernel_alone_us"] = timeit(lambda: model._stage_negatives(roots, B, pairs=model._pairs, cursor=model._cursor), s)
    # half the roots (512) through the same unstaged path
    r2 = model.ids_buffer(512, parity=1)[0][:512]
    r2.copy_(roots[:512])
    torch.cuda.synchronize()
    res["unsup_512_roots_prestaged_us"] = timeit(lambda: model._sample_phase(r2, 512, 1), s)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
