"""Micro-benchmark: the layer-0 contraction at the Reddit / unsupervised / GCN step shapes -- fp32-MFMA stream kernel
(gs_sage_dense_fwd_stream) vs the LDS-tiled split-MFMA kernel (gs_sage_dense_fwd_tiled3: fp32 operands as three bf16 pieces), alone
(hot operands, back-to-back launches) and with a share of the next step's gather co-scheduled.
    python benchmarks/micro_split.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import ops  # noqa: E402
from graphsage_amd.ops import Mat  # noqa: E402


def timeit(fn, stream, iters=40, warmup=8):
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
    N, F, B, s1, s2, D = 232965, 602, 512, 25, 10, 128
    g = torch.Generator(device="cpu").manual_seed(0)
    X = Mat(torch.randn((N + 1, 608), generator=g).to(dev), F)
    X.buf[:, F:] = 0
    res = {}
    for tag, roots in (("reddit", B), ("unsup", 2 * B + 20)):
        n = roots * (1 + s2)
        ids_self = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
        idx2 = torch.randint(0, N, (roots * s2 * s1,), generator=g, dtype=torch.int32).to(dev)
        means = Mat.zeros(n, F, dev, 32)
        means.buf[:, :F].normal_()
        Ws = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
        Wn = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
        W3s, W3n = ops.split_rows(Ws, stream=s), ops.split_rows(Wn, stream=s)
        out = Mat.zeros(n, 2 * D, dev)
        m2 = Mat.zeros(roots * s2, F, dev, 32)
        job = [ops.gather_job(X, idx2, roots * s2, s1, m2)]
        r = {}
        r["split_rows_us"] = timeit(lambda: ops.split_rows(Ws, out=W3s, stream=s), s)
        r["fp32_stream_alone_us"] = timeit(lambda: ops.sage_dense_fwd_stream(X, ids_self, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, [], stream=s), s)
        r["split_alone_us"] = timeit(lambda: ops.sage_dense_fwd_tiled3(X, ids_self, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, [], stream=s), s)
        r["gather_alone_us"] = timeit(lambda: ops.gather_mean_fwd(X, idx2, roots * s2, s1, out=m2, stream=s), s)
        for frac in (0.15, 0.5, 1.0):
            head, _ = ops.split_gather_jobs(job, frac)
            r["fp32_stream_cogather_%.2f_us" % frac] = timeit(lambda: ops.sage_dense_fwd_stream(X, ids_self, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, head, stream=s), s)
            r["split_cogather_%.2f_us" % frac] = timeit(lambda: ops.sage_dense_fwd_tiled3(X, ids_self, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, head, stream=s), s)
        # GCN form: one term, N = 256
        Wg = Mat(torch.randn((F, 2 * D), generator=g).to(dev) * 0.05, 2 * D)
        W3g = ops.split_rows(Wg, stream=s)
        r["gcn_fp32_stream_alone_us"] = timeit(lambda: ops.sage_dense_fwd_stream(None, None, means, n, None, Wg, 2 * D, ops.ACT_RELU, None, out, [], stream=s), s)
        r["gcn_split_alone_us"] = timeit(lambda: ops.sage_dense_fwd_tiled3(None, None, means, n, None, Wg, 2 * D, ops.ACT_RELU, None, out, [], stream=s), s)
        res[tag] = r
    # RMAT's layer 0: F = 256 (rows from a small table here: the contraction's own time)
    F2 = 256
    X2 = Mat(torch.randn((200000, F2), generator=g).to(dev), F2)
    n = B * (1 + s2)
    ids_self = torch.randint(0, 200000, (n,), generator=g, dtype=torch.int32).to(dev)
    means = Mat.zeros(n, F2, dev, 32)
    means.buf[:, :F2].normal_()
    Ws = Mat(torch.randn((F2, D), generator=g).to(dev) * 0.05, D)
    Wn = Mat(torch.randn((F2, D), generator=g).to(dev) * 0.05, D)
    W3s, W3n = ops.split_rows(Ws, stream=s), ops.split_rows(Wn, stream=s)
    out = Mat.zeros(n, 2 * D, dev)
    res["rmat_f256"] = {
        "fp32_stream_alone_us": timeit(lambda: ops.sage_dense_fwd_stream(X2, ids_self, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, [], stream=s), s),
        "split_alone_us": timeit(lambda: ops.sage_dense_fwd_tiled3(X2, ids_self, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, [], stream=s), s)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def pool():
    """The pooling MLP GEMM of the max-pool step at its real shape: 83 k distinct rows x 602 -> 512."""
    dev = torch.device("cuda:0")
    st = ops.Stream()
    s = st.handle
    N, F, H = 232965, 602, 512
    g = torch.Generator(device="cpu").manual_seed(0)
    X = Mat(torch.randn((N + 1, 608), generator=g).to(dev), F)
    X.buf[:, F:] = 0
    rows = 133120
    ids = torch.sort(torch.randperm(N, generator=g)[:100000]).values.to(torch.int32).to(dev)
    cnt = torch.tensor([83000], dtype=torch.int32, device=dev)
    W = Mat(torch.randn((F, H), generator=g).to(dev) * 0.05, H)
    b = torch.zeros(H, device=dev)
    W3 = ops.split_rows(W, stream=s)
    out = Mat.zeros(rows, H, dev)
    r = {}
    r["pool_fp32_tiled_us"] = timeit(lambda: ops.call("gs_dense_fwd_rows_dev", X.ptr, X.ld, ops.ptr(ids), F, rows, ops.ptr(cnt), W.ptr, W.ld,
                                                       H, ops.ACT_RELU, ops.ptr(b), out.ptr, out.ld, s), s, iters=10, warmup=3)
    r["pool_split_tiled_us"] = timeit(lambda: ops.call("gs_dense_fwd_rows_split", X.ptr, X.ld, ops.ptr(ids), F, rows, ops.ptr(cnt), ops.ptr(W3),
                                                        H, ops.ACT_RELU, ops.ptr(b), out.ptr, out.ld, s), s, iters=10, warmup=3)
    ws = torch.empty(ops.split_tiled_ws_words(), dtype=torch.float32, device=dev)
    # row counts around the Reddit step's distinct ids: 1300 tiles (83 k), an exact multiple of 256 (81,920), and in between
    for c in (83000, 81920, 84500, 90000, 98000):
        cnt.fill_(c)
        r["pool_split_tiled_%d_us" % c] = timeit(lambda: ops.call("gs_dense_fwd_rows_split", X.ptr, X.ld, ops.ptr(ids), F, rows, ops.ptr(cnt), ops.ptr(W3),
                                                                   H, ops.ACT_RELU, ops.ptr(b), out.ptr, out.ld, s), s, iters=10, warmup=3)
        r["pool_split_tiled_ws_%d_us" % c] = timeit(lambda: ops.call("gs_dense_fwd_rows_split_ws", X.ptr, X.ld, ops.ptr(ids), F, rows, ops.ptr(cnt),
                                                                      ops.ptr(W3), H, ops.ACT_RELU, ops.ptr(b), out.ptr, out.ld, ops.ptr(ws),
                                                                      4 * ws.numel(), s), s, iters=10, warmup=3)
    X2, rexp = ops.split_table_f16(X, stream=s)
    W2 = ops.split_rows_f16(W, stream=s)
    for c in (83000, 81920, 90000):
        cnt.fill_(c)
        r["pool_split16_%d_us" % c] = timeit(lambda: ops.call("gs_dense_fwd_rows_split16", ops.ptr(X2), ops.ptr(rexp), ops.ptr(ids), F, rows, ops.ptr(cnt),
                                                               ops.ptr(W2), H, ops.ACT_RELU, ops.ptr(b), out.ptr, out.ld, None, 0, s), s, iters=10, warmup=3)
        r["pool_split16_ws_%d_us" % c] = timeit(lambda: ops.call("gs_dense_fwd_rows_split16", ops.ptr(X2), ops.ptr(rexp), ops.ptr(ids), F, rows, ops.ptr(cnt),
                                                                  ops.ptr(W2), H, ops.ACT_RELU, ops.ptr(b), out.ptr, out.ld, ops.ptr(ws),
                                                                  4 * ws.numel(), s), s, iters=10, warmup=3)
    r["GF"] = 2.0 * 83000 * F * H / 1e9
    print(json.dumps(r, indent=1))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "pool":
    pool()


def poolloop(seconds=24.0):
    """The pooling MLP in a loop for `seconds` (benchmarks/r5_clock_probe.sh polls rocm-smi meanwhile)."""
    import time
    dev = torch.device("cuda:0")
    st = ops.Stream()
    s = st.handle
    N, F, H = 232965, 602, 512
    g = torch.Generator(device="cpu").manual_seed(0)
    X = Mat(torch.randn((N + 1, 608), generator=g).to(dev), F)
    X.buf[:, F:] = 0
    rows = 133120
    ids = torch.sort(torch.randperm(N, generator=g)[:100000]).values.to(torch.int32).to(dev)
    cnt = torch.tensor([81920], dtype=torch.int32, device=dev)
    W = Mat(torch.randn((F, H), generator=g).to(dev) * 0.05, H)
    b = torch.zeros(H, device=dev)
    W3 = ops.split_rows(W, stream=s)
    out = Mat.zeros(rows, H, dev)
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        us = timeit(lambda: ops.call("gs_dense_fwd_rows_split", X.ptr, X.ld, ops.ptr(ids), F, rows, ops.ptr(cnt), ops.ptr(W3),
                                      H, ops.ACT_RELU, ops.ptr(b), out.ptr, out.ld, s), s, iters=200, warmup=0)
        n += 1
        print("t=%.2fs: %.1f us per launch (81,920 rows = five full rounds)" % (time.time() - t0, us), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "poolloop":
    poolloop()


def pool16():
    """gs_dense_fwd_rows_split16 alone at 81,920 rows (five full rounds of one-per-CU workgroups) and 83,000 (diagnostics variants)."""
    dev = torch.device("cuda:0")
    st = ops.Stream()
    s = st.handle
    N, F, H = 232965, 602, 512
    g = torch.Generator(device="cpu").manual_seed(0)
    X = Mat(torch.randn((N + 1, 608), generator=g).to(dev), F)
    X.buf[:, F:] = 0
    rows = 133120
    ids = torch.sort(torch.randperm(N, generator=g)[:100000]).values.to(torch.int32).to(dev)
    cnt = torch.tensor([81920], dtype=torch.int32, device=dev)
    W = Mat(torch.randn((F, H), generator=g).to(dev) * 0.05, H)
    b = torch.zeros(H, device=dev)
    out = Mat.zeros(rows, H, dev)
    X2, rexp = ops.split_table_f16(X, stream=s)
    W2 = ops.split_rows_f16(W, stream=s)
    ws = torch.empty(ops.split_tiled_ws_words(), dtype=torch.float32, device=dev)
    r = {}
    for c in (81920, 83000):
        cnt.fill_(c)
        r["pool_split16_ws_%d_us" % c] = timeit(lambda: ops.call("gs_dense_fwd_rows_split16", ops.ptr(X2), ops.ptr(rexp), ops.ptr(ids), F, rows, ops.ptr(cnt),
                                                                  ops.ptr(W2), H, ops.ACT_RELU, ops.ptr(b), out.ptr, out.ld, ops.ptr(ws),
                                                                  4 * ws.numel(), s), s, iters=20, warmup=5)
    r["split_rows_f16_us"] = timeit(lambda: ops.split_rows_f16(W, out=W2, stream=s), s, iters=20, warmup=3)
    print(json.dumps(r))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "pool16":
    pool16()


def poolloop16(seconds=14.0):
    """gs_dense_fwd_rows_split16 in a loop for `seconds` (rocm-smi polled meanwhile: clock / power of the two-piece form)."""
    import time
    dev = torch.device("cuda:0")
    st = ops.Stream()
    s = st.handle
    N, F, H = 232965, 602, 512
    g = torch.Generator(device="cpu").manual_seed(0)
    X = Mat(torch.randn((N + 1, 608), generator=g).to(dev), F)
    X.buf[:, F:] = 0
    rows = 133120
    ids = torch.sort(torch.randperm(N, generator=g)[:100000]).values.to(torch.int32).to(dev)
    cnt = torch.tensor([81920], dtype=torch.int32, device=dev)
    W = Mat(torch.randn((F, H), generator=g).to(dev) * 0.05, H)
    b = torch.zeros(H, device=dev)
    out = Mat.zeros(rows, H, dev)
    X2, rexp = ops.split_table_f16(X, stream=s)
    W2 = ops.split_rows_f16(W, stream=s)
    t0 = time.time()
    while time.time() - t0 < seconds:
        us = timeit(lambda: ops.call("gs_dense_fwd_rows_split16", ops.ptr(X2), ops.ptr(rexp), ops.ptr(ids), F, rows, ops.ptr(cnt),
                                      ops.ptr(W2), H, ops.ACT_RELU, ops.ptr(b), out.ptr, out.ld, None, 0, s), s, iters=200, warmup=0)
        print("t=%.2fs: %.1f us per launch (81,920 rows = five full rounds)" % (time.time() - t0, us), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "poolloop16":
    poolloop16()
