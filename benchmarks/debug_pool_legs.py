"""Two max-pool models in ONE process (two-piece fp16 MLP, then three-piece bf16), same seed / data / step count: initial-weight
checksums, loss per checkpoint.  (bench.py's aux legs showed different last-batch losses.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graphsage_amd import inits  # noqa: E402
from graphsage_amd.utils import reddit_shaped_device  # noqa: E402


def main():
    args = bench.parse_args(["--no-cpu-baseline", "--no-aux"])
    dev = torch.device("cuda:0")
    DG = reddit_shaped_device(dev, n_nodes=args.nodes, feat_dim=args.feat_dim, num_classes=args.classes, avg_degree=args.avg_degree,
                              seed=123, feat_signal=args.feat_signal)
    B = 512
    for form in ("1", "0", "1"):
        os.environ["GS_POOL_F16"] = form
        inits.set_seed(123)
        e, model, ph, _ = bench.build_model(DG, args, 1, 0, "graphsage_maxpool")
        model.attach_device_epoch(np.random.RandomState(123).permutation(DG.train_nodes), DG.label_table)
        cs0 = float(e.params.double().abs().sum().item())
        out = []
        for chunk in (22, 40, 40, 40, 40, 40):
            model.train_steps_device(B, chunk, steps_per_launch=32)
            e.sync()
            out.append(round(float(model._fetch(B)[0]), 4))
        print("GS_POOL_F16=%s pool_f16=%s: init checksum %.6f, losses %s, step counter %d" % (form, e.pool_f16, cs0, out, int(e.step_dev.item())), flush=True)
        del model


if __name__ == "__main__":
    main()
