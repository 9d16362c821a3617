"""Micro-benchmark (round 5): the layer-0 contraction as "stream" kernel (gs_stream.hip: 32 x 64 tiles, K quartered) vs
"panel" kernel (gs_panel.hip: one workgroup per 48 x 128 panel for the whole K), alone (hot and cold operands) and with a
share of the next step's gather riding, at the shapes of every timed configuration.
    python benchmarks/micro_panel.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import ops  # noqa: E402
from graphsage_amd.ops import Mat  # noqa: E402


def timeit(fn, stream, iters=30, warmup=5):
    for _ in range(warmup):
        fn()
    e0, e1 = ops.Event(), ops.Event()
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    return e0.elapsed_ms(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    st = ops.Stream()
    s = st.handle
    g = torch.Generator(device="cpu").manual_seed(0)
    N, s1, s2 = 232965, 25, 10
    big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)

    def cold(fn):
        tot = 0.0
        for _ in range(8):
            big.add_(1.0)
            torch.cuda.synchronize()
            e0, e1 = ops.Event(), ops.Event()
            e0.record(s); fn(); e1.record(s)
            tot += e0.elapsed_ms(e1) * 1e3
        return tot / 8

    res = {}
    X602 = Mat(torch.randn((N + 1, 608), generator=g).to(dev), 602)
    X602.buf[:, 602:] = 0
    X256 = Mat(torch.randn((N + 1, 256), generator=g).to(dev), 256)
    # (name, table, n rows, F, out_dim, two terms, roots of the gather jobs, fan-out of hop 2)
    cfgs = [("reddit_mean", X602, 5632, 602, 128, True, 512, 25), ("rmat_mean", X256, 5632, 256, 128, True, 512, 15),
            ("unsup_mean", X602, 11484, 602, 128, True, 1044, 25), ("reddit_gcn", X602, 5632, 602, 256, False, 512, 25)]
    for name, X, n, F, D, two, B, fan in cfgs:
        ids_self = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
        means = Mat.zeros(n, F, dev, 32)
        means.buf[:, :F].normal_()
        Ws = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
        Wn = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
        out = Mat.zeros(n, (2 if two else 1) * D, dev)
        idx2 = torch.randint(0, N, (B * s2 * fan,), generator=g, dtype=torch.int32).to(dev)
        idx1 = torch.randint(0, N, (B * s2,), generator=g, dtype=torch.int32).to(dev)
        m2, m1 = Mat.zeros(B * s2, F, dev, 32), Mat.zeros(B, F, dev, 32)
        jobs_all = [ops.gather_job(X, idx2, B * s2, fan, m2), ops.gather_job(X, idx1, B, s2, m1)]
        torch.cuda.synchronize()
        a_self, a_idx, w_self = (X, ids_self, Ws) if two else (None, None, None)
        forms = (("stream", ops.sage_dense_fwd_stream), ("panel", ops.sage_dense_fwd_panel))
        res["%s/gather_alone_us" % name] = timeit(lambda: [ops.gather_mean_fwd(X, idx2, B * s2, fan, out=m2, stream=s),
                                                          ops.gather_mean_fwd(X, idx1, B, s2, out=m1, stream=s)], s)
        for fname, fn in forms:
            call = lambda jobs: fn(a_self, a_idx, means, n, w_self, Wn, D, ops.ACT_RELU, None, out, jobs, stream=s)
            res["%s/%s_alone_hot_us" % (name, fname)] = timeit(lambda: call([]), s)
            res["%s/%s_alone_cold_us" % (name, fname)] = cold(lambda: call([]))
            for frac in (0.15, 0.3, 0.5, 1.0):
                head, _ = ops.split_gather_jobs(jobs_all, frac)
                res["%s/%s_gather_%.2f_us" % (name, fname, frac)] = timeit(lambda: call(head), s)
        flops = 2.0 * n * F * D * (2 if two else 1)
        res["%s/gflop" % name] = flops / 1e9
    for k in sorted(res):
        print("%-44s %8.2f" % (k, res[k]))
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/micro_panel.json"
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    json.dump(res, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
