"""micro-F1 half of the metric as a measurement (bench.f1_legs on the benched graph, more seeds / steps than the
default bench line): per seed the torch-CPU port of the reference graph, the MI355X engine on the same padded tables
with the same permutations injected (numerics only), and the MI355X engine with the native CSR sampler under the
reference's law and under the default iid law -- all from the same initial weights.  Mean +- std per leg and paired
differences vs the CPU port -> gpurun_out/f1_parity.json (copied to profiles/ when committed).

    python benchmarks/f1_parity.py [--seeds 10] [--steps 150] [--val-nodes 8192] [--avg_degree 492]
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=10)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--val-nodes", dest="val_nodes", type=int, default=8192)
    ap.add_argument("--avg_degree", type=int, default=492)
    ap.add_argument("--nodes", type=int, default=232965)
    ap.add_argument("--feat_signal", type=float, default=0.02)
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from graphsage_amd.utils import reddit_shaped_device
    args = bench.parse_args(["--nodes", str(a.nodes), "--avg_degree", str(a.avg_degree), "--feat_signal", str(a.feat_signal)])
    dev = torch.device("cuda:0")
    t0 = time.time()
    DG = reddit_shaped_device(dev, n_nodes=args.nodes, feat_dim=args.feat_dim, num_classes=args.classes,
                              avg_degree=args.avg_degree, seed=123, feat_signal=args.feat_signal)
    f1, cb = bench.f1_legs(DG, args, args.batch_size, args.samples_1, args.samples_2, args.feat_dim, args.steps_per_launch,
                           seeds=a.seeds, steps=a.steps, n_val=a.val_nodes)
    res = {"graph": {"nodes": args.nodes, "avg_degree": args.avg_degree, "feat_signal": args.feat_signal}, "micro_f1": f1,
           "cpu_baseline": cb, "wall_s": time.time() - t0}
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "f1_parity.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
