"""Micro-benchmarks of the individual HIP kernels at the Reddit shapes (B=512, F=602, 25x10, dim 128).

    python benchmarks/micro.py [--iters 50]
Prints one JSON line per kernel with average launch time (hipEvents on the launch stream),
algorithmic bytes / flops and the implied GB/s or TFLOP/s.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import ops  # noqa: E402
from graphsage_amd.ops import Mat  # noqa: E402


def timeit(fn, stream, iters, warmup=5):
    for _ in range(warmup):
        fn()
    e0, e1 = ops.Event(), ops.Event()
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    return e0.elapsed_ms(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--nodes", type=int, default=232965)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    st = ops.Stream()
    s = st.handle
    N, F, B, s1, s2, D = args.nodes, 602, 512, 25, 10, 128
    g = torch.Generator(device="cpu").manual_seed(0)
    X = Mat(torch.randn((N + 1, 608), generator=g).to(dev), F)
    X.buf[:, F:] = 0
    idx2 = torch.randint(0, N, (B * s2 * s1,), generator=g, dtype=torch.int32).to(dev)
    idx1 = torch.randint(0, N, (B * s2,), generator=g, dtype=torch.int32).to(dev)
    res = []

    def rec(name, us, bytes_=None, flops=None):
        r = {"kernel": name, "us": round(us, 2)}
        if bytes_:
            r["GBps"] = round(bytes_ / us / 1e3, 1)
            r["frac_hbm_8TBs"] = round(bytes_ / us / 1e3 / 8000, 3)
        if flops:
            r["TFLOPs"] = round(flops / us / 1e6, 2)
            r["frac_mfma_f32"] = round(flops / us / 1e6 / 157.3, 3)
        print(json.dumps(r), flush=True)
        res.append(r)

    # K2 hop-2 gather+mean: [5120, 25] rows of 602 floats
    n2 = B * s2
    mean2 = Mat.zeros(n2, F, dev)
    us = timeit(lambda: ops.gather_mean_fwd(X, idx2, n2, s1, out=mean2, stream=s), s, args.iters)
    rec("K2 gather_mean hop2 [5120x25x602]", us, bytes_=n2 * s1 * F * 4 + n2 * s1 * 4 + n2 * F * 4)
    mean1 = Mat.zeros(B, F, dev)
    us = timeit(lambda: ops.gather_mean_fwd(X, idx1, B, s2, out=mean1, stream=s), s, args.iters)
    rec("K2 gather_mean hop1 [512x10x602]", us, bytes_=B * s2 * F * 4 + B * s2 * 4 + B * F * 4)

    # K1 sampler over a synthetic CSR (avg degree 50)
    deg = torch.randint(1, 100, (N,), generator=g)
    rowptr = torch.zeros(N + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(deg, 0)
    col = torch.randint(0, N, (int(rowptr[-1]),), generator=g, dtype=torch.int32).to(dev)
    rowptr = rowptr.to(dev)
    out2 = torch.empty(n2 * s1, dtype=torch.int32, device=dev)
    us = timeit(lambda: ops.sample_uniform_csr(rowptr, col, N, N, idx1, s1, 1, out=out2, stream=s), s, args.iters)
    rec("K1 sample_csr hop2 [5120x25]", us)

    # K3 layer-0 hop-1 dual GEMM with gathered self rows
    Ws = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    Wn = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    h1 = Mat.zeros(n2, 2 * D, dev)
    us = timeit(lambda: ops.sage_dense_fwd(X, idx1, mean2, None, n2, Ws, Wn, D, True, ops.ACT_RELU, None, h1, stream=s),
                s, args.iters)
    rec("K3 sage_dense_fwd L0 hop1 [5120x602x128 x2]", us, flops=2 * 2 * n2 * F * D)
    # K6 weight gradient slabs for the same shape
    nsl = 16
    slabs = torch.zeros(nsl * F * D, device=dev)
    us = timeit(lambda: ops.dense_wgrad(X, idx1, h1, 0, D, n2, nsl, slabs, D, stream=s), s, args.iters)
    rec("K6 dense_wgrad L0 hop1 [602x128 <- 5120] x16 slabs", us, flops=2 * n2 * F * D)
    nb = n2 + B
    idsb = torch.randint(0, N, (nb,), generator=g, dtype=torch.int32).to(dev)
    hb = Mat(torch.randn((nb, 2 * D), generator=g).to(dev), 2 * D)
    for nsl in (13, 26, 32, 44):
        slabs = torch.zeros(nsl * F * D, device=dev)
        us = timeit(lambda: ops.dense_wgrad(X, idsb, hb, 0, D, nb, nsl, slabs, D, stream=s), s, args.iters)
        rec("K6 dense_wgrad L0 all hops [602x128 <- 5632] x%d slabs tile=%s" % (nsl, os.environ.get("GS_WGRAD_TILE", "6464")), us,
            flops=2 * nb * F * D)
    # layer-1 sized contractions (tiny M)
    h0 = Mat(torch.randn((B, 2 * D), generator=g).to(dev), 2 * D)
    m1 = Mat(torch.randn((B, 2 * D), generator=g).to(dev), 2 * D)
    W1s = Mat(torch.randn((2 * D, D), generator=g).to(dev) * 0.05, D)
    W1n = Mat(torch.randn((2 * D, D), generator=g).to(dev) * 0.05, D)
    o1 = Mat.zeros(B, 2 * D, dev)
    us = timeit(lambda: ops.sage_dense_fwd(h0, None, m1, None, B, W1s, W1n, D, True, ops.ACT_IDENTITY, None, o1, stream=s), s, args.iters)
    rec("K3 sage_dense_fwd L1 [512x256x128 x2]", us, flops=2 * 2 * B * 2 * D * D)
    t2 = Mat.zeros(B, 4 * D, dev)
    us = timeit(lambda: ops.sage_dense_dgrad(o1, B, D, True, W1s, W1n, 2 * D, t2, stream=s), s, args.iters)
    rec("K6 sage_dense_dgrad L1 [512x128 -> 2x 512x256]", us, flops=2 * 2 * B * 2 * D * D)
    # fused head
    C = 41
    Wh = Mat.zeros(2 * D, C, dev)
    Wh.buf[:, :C] = torch.randn((2 * D, C), generator=g).to(dev) * 0.05
    bh = torch.zeros(C, device=dev)
    lab = Mat.zeros(B, C, dev)
    lab.buf[:, :C] = torch.nn.functional.one_hot(torch.randint(0, C, (B,), generator=g), C).float().to(dev)
    y, lo, pr, dl = Mat.zeros(B, 2 * D, dev), Mat.zeros(B, C, dev), Mat.zeros(B, C, dev), Mat.zeros(B, C, dev)
    lr_, dx = torch.zeros(B, device=dev), Mat.zeros(B, 2 * D, dev)
    us = timeit(lambda: ops.head_fwd_bwd(o1, B, Wh, bh, lab, C, False, y, lo, pr, dl, lr_, dx, stream=s), s, args.iters)
    rec("K5 head_fwd_bwd [512x256x41]", us)
    # K4 MaxPool MLP GEMM [128000, 602] x [602, 512] with gathered rows
    Wm = Mat(torch.randn((F, 512), generator=g).to(dev) * 0.05, 512)
    H = Mat.zeros(n2 * s1, 512, dev)
    us = timeit(lambda: ops.sage_dense_fwd(None, None, X, idx2, n2 * s1, None, Wm, 512, False, ops.ACT_RELU, None, H,
                                           stream=s), s, max(5, args.iters // 10), warmup=2)
    rec("K4 maxpool MLP GEMM [128000x602x512]", us, flops=2 * n2 * s1 * F * 512)
    st.sync()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/micro.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
