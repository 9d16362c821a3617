"""Micro-benchmark: layer-0 contraction and grouped weight gradients at the Reddit step shapes, LDS-tiled kernels vs
the stream (LDS-free wave) kernels, alone and with the next step's gather co-scheduled.
    python benchmarks/micro_stream.py"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import _lib, ops  # noqa: E402
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
    N, F, B, s1, s2, D = 232965, 602, 512, 25, 10, 128
    g = torch.Generator(device="cpu").manual_seed(0)
    X = Mat(torch.randn((N + 1, 608), generator=g).to(dev), F)
    X.buf[:, F:] = 0
    n = B + B * s2
    idx2 = torch.randint(0, N, (B * s2 * s1,), generator=g, dtype=torch.int32).to(dev)
    idx1 = torch.randint(0, N, (B * s2,), generator=g, dtype=torch.int32).to(dev)
    ids_self = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
    selfd, means = Mat.zeros(n, F, dev), Mat.zeros(n, F, dev)
    selfd.buf[:, :F].normal_(); means.buf[:, :F].normal_()
    Ws = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    Wn = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    out = Mat.zeros(n, 2 * D, dev)
    m2, m1, sd2 = Mat.zeros(B * s2, F, dev), Mat.zeros(B, F, dev), Mat.zeros(n, F, dev)
    jobs_all = [ops.gather_job(X, idx2, B * s2, s1, m2), ops.gather_job(X, idx1, B, s2, m1), ops.gather_job(X, ids_self, n, 1, sd2)]
    res = {}
    res["gather_alone_us"] = timeit(lambda: [ops.gather_mean_fwd(X, idx2, B * s2, s1, out=m2, stream=s),
                                             ops.gather_mean_fwd(X, idx1, B, s2, out=m1, stream=s),
                                             ops.gather_rows(X, ids_self, out=sd2, stream=s)], s)
    res["fwd_tiled_dense_alone_us"] = timeit(lambda: ops.sage_dense_fwd(selfd, None, means, None, n, Ws, Wn, D, True, ops.ACT_RELU, None, out, stream=s), s)
    res["fwd_tiled_gathered_alone_us"] = timeit(lambda: ops.sage_dense_fwd(X, ids_self, means, None, n, Ws, Wn, D, True, ops.ACT_RELU, None, out, stream=s), s)
    res["fwd_stream_alone_us"] = timeit(lambda: ops.sage_dense_fwd_stream(selfd, None, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, [], stream=s), s)
    for frac in (1.0, 0.7, 0.5, 0.3):
        head, tail = ops.split_gather_jobs(jobs_all, frac)
        res["fwd_stream_cogather_%.1f_us" % frac] = timeit(lambda: ops.sage_dense_fwd_stream(selfd, None, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, head, stream=s), s)
        res["fwd_tiled_cogather_%.1f_us" % frac] = timeit(lambda: ops.sage_dense_fwd_cogather(X, ids_self, means, None, n, Ws, Wn, D, True, ops.ACT_RELU, None, out, head[:4], stream=s), s)
    # weight gradients of the step
    dz0 = Mat.zeros(n, 2 * D, dev); dz0.buf.normal_()
    h0 = Mat.zeros(B, 2 * D, dev); h0.buf.normal_()
    m1b = Mat.zeros(B, 2 * D, dev); m1b.buf.normal_()
    dz1 = Mat.zeros(B, 2 * D, dev); dz1.buf.normal_()
    y = Mat.zeros(B, 2 * D, dev); y.buf.normal_()
    dl = Mat.zeros(B, 41, dev); dl.buf.normal_()
    ones = Mat(torch.ones((B, 4), device=dev), 1)

    def descs(slices0, slices1, a_self, a_idx):
        probs = [(a_self, a_idx, dz0, 0, F, D, n, slices0), (means, None, dz0, D, F, D, n, slices0), (h0, None, dz1, 0, 2 * D, D, B, slices1),
                 (m1b, None, dz1, D, 2 * D, D, B, slices1), (y, None, dl, 0, 2 * D, 41, B, slices1), (ones, None, dl, 0, 1, 41, B, slices1)]
        arr = (_lib.WgradDesc * len(probs))()
        keep = []
        for i, (A, ai, Z, col0, d, o, nn, ns) in enumerate(probs):
            ld_slab = (o + 3) & ~3
            sl = torch.zeros(ns * d * ld_slab, device=dev)
            keep.append(sl)
            arr[i].A, arr[i].a_idx, arr[i].dZ, arr[i].slabs = A.ptr, ops.ptr(ai), Z.ptr, sl.data_ptr()
            arr[i].lda, arr[i].ldz, arr[i].ld_slab, arr[i].n = A.ld, Z.ld, ld_slab, nn
            arr[i].d, arr[i].col0, arr[i].out_dim, arr[i].n_slabs = d, col0, o, ns
            arr[i].a_rows = A.rows if ai is not None else 0
        return arr, keep
    torch.cuda.synchronize()
    arr_t, k1 = descs(32, 8, X, ids_self)
    res["wgrad_tiled_gathered_alone_us"] = timeit(lambda: ops.call("gs_dense_wgrad_grouped", ctypes.addressof(arr_t), 6, s), s)
    arr_g, kg = descs(22, 2, X, ids_self)
    jn0 = (_lib.GatherDesc * 1)()
    res["wgrad_stream_GATHERED_A_alone_slices22_us"] = timeit(lambda: ops.call("gs_dense_wgrad_grouped_stream", ctypes.addressof(arr_g), 6, ctypes.addressof(jn0), 0, s), s)
    # cold operands: flush L2/MALL between launches by streaming a 1 GB buffer
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
    arr_d, kd = descs(22, 2, selfd, None)
    res["wgrad_stream_dense_COLD_slices22_us"] = cold(lambda: ops.call("gs_dense_wgrad_grouped_stream", ctypes.addressof(arr_d), 6, ctypes.addressof(jn0), 0, s))
    res["wgrad_stream_GATHERED_COLD_slices22_us"] = cold(lambda: ops.call("gs_dense_wgrad_grouped_stream", ctypes.addressof(arr_g), 6, ctypes.addressof(jn0), 0, s))
    res["wgrad_tiled_gathered_COLD_us"] = cold(lambda: ops.call("gs_dense_wgrad_grouped", ctypes.addressof(arr_t), 6, s))
    res["fwd_tiled_gathered_COLD_us"] = cold(lambda: ops.sage_dense_fwd(X, ids_self, means, None, n, Ws, Wn, D, True, ops.ACT_RELU, None, out, stream=s))
    res["fwd_tiled_dense_COLD_us"] = cold(lambda: ops.sage_dense_fwd(selfd, None, means, None, n, Ws, Wn, D, True, ops.ACT_RELU, None, out, stream=s))
    for sl0 in (22,):
        arr_s, k2 = descs(sl0, 2, selfd, None)
        jn = (_lib.GatherDesc * 1)()
        res["wgrad_stream_alone_slices%d_us" % sl0] = timeit(lambda: ops.call("gs_dense_wgrad_grouped_stream", ctypes.addressof(arr_s), 6, ctypes.addressof(jn), 0, s), s)
        for frac in (0.5, 0.3):
            head, tail = ops.split_gather_jobs(jobs_all, 1.0 - frac)
            jt = (_lib.GatherDesc * len(tail))(*tail)
            res["wgrad_stream_slices%d_cogather_%.1f_us" % (sl0, frac)] = timeit(
                lambda: ops.call("gs_dense_wgrad_grouped_stream", ctypes.addressof(arr_s), 6, ctypes.addressof(jt), len(tail), s), s)
    for k in sorted(res):
        print("%-48s %8.2f" % (k, res[k]))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/micro_stream.json", "w"), indent=1)


if __name__ == "__main__":
    main()
