"""Micro-benchmark: the grouped weight-gradient launch of a supervised mean step (layer 0: 602 x 128 x 2 over 5632 rows, the self term
row-gathered; layer 1: 256 x 128 x 2 over 512 rows; head 256 x 41 + bias) -- gs_dense_wgrad_grouped_stream (fp32 MFMA, one wave per
64 x 64 tile and slice) vs gs_dense_wgrad_grouped_tiled3 (three bf16 pieces, LDS-tiled, one workgroup per 64 x 128 tile and slice),
alone and with a share of the next step's gather riding; also the unsupervised step's shape (11484 rows) and RMAT's (F = 256).
    python benchmarks/micro_wgrad.py"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import ops, _lib  # noqa: E402
from graphsage_amd.ops import Mat  # noqa: E402
from benchmarks.micro_split import timeit  # noqa: E402


def problems(dev, g, n0, F, slabs0, slabs1=2, N=232965):
    X = Mat(torch.randn((N + 1, ((F + 31) // 32) * 32), generator=g).to(dev), F)
    ids = torch.randint(0, N, (n0,), generator=g, dtype=torch.int32).to(dev)
    means = Mat.zeros(n0, F, dev, 32)
    means.buf[:, :F].normal_()
    dh0 = Mat(torch.randn((n0, 256), generator=g).to(dev) * 0.1, 256)
    n1 = 512
    h0 = Mat(torch.randn((n1, 256), generator=g).to(dev), 256)
    m1 = Mat(torch.randn((n1, 256), generator=g).to(dev), 256)
    dz = Mat(torch.randn((n1, 256), generator=g).to(dev) * 0.1, 256)
    o1 = Mat(torch.randn((n1, 256), generator=g).to(dev), 256)
    dl = Mat(torch.randn((n1, 44), generator=g).to(dev) * 0.1, 41)
    ones = Mat(torch.ones((n1, 4), device=dev), 1)
    probs = [(X, ids, dh0, 0, 128, n0, F, slabs0), (means, None, dh0, 128, 128, n0, F, slabs0), (h0, None, dz, 0, 128, n1, 256, slabs1),
             (m1, None, dz, 128, 128, n1, 256, slabs1), (o1, None, dl, 0, 41, n1, 256, slabs1), (ones, None, dl, 0, 41, n1, 1, 1)]
    descs = (_lib.WgradDesc * len(probs))()
    keep = [X, ids, means, dh0, h0, m1, dz, o1, dl, ones]
    for i, (A, ai, Z, col0, o, n, d, ns) in enumerate(probs):
        ld_slab = (o + 3) & ~3
        sl = torch.zeros(ns * d * ld_slab, device=dev)
        keep.append(sl)
        descs[i].A, descs[i].a_idx, descs[i].dZ, descs[i].slabs = A.ptr, ops.ptr(ai), Z.ptr, sl.data_ptr()
        descs[i].lda, descs[i].ldz, descs[i].ld_slab, descs[i].n = A.ld, Z.ld, ld_slab, n
        descs[i].d, descs[i].col0, descs[i].out_dim, descs[i].n_slabs = d, col0, o, ns
        descs[i].a_rows = A.rows if ai is not None else 0
    return descs, keep, X


def main():
    dev = torch.device("cuda:0")
    st = ops.Stream()
    s = st.handle
    g = torch.Generator(device="cpu").manual_seed(0)
    res = {}
    for tag, n0, F, stream_slabs, tiled in (("reddit", 5632, 602, 22, (8, 10, 11, 12, 16, 22)), ("unsup", 11484, 602, 25, (12, 16, 23, 24, 32)),
                                            ("rmat_f256", 5632, 256, 22, (8, 16, 22, 28))):
        r = {}
        d0, keep0, X = problems(dev, g, n0, F, stream_slabs, 2)
        idx2 = torch.randint(0, X.rows - 1, ((n0 // 11) * 10 * 25,), generator=g, dtype=torch.int32).to(dev)
        m2 = Mat.zeros((n0 // 11) * 10, F, dev, 32)
        job = [ops.gather_job(X, idx2, (n0 // 11) * 10, 25, m2)]
        head, _ = ops.split_gather_jobs(job, 0.35)
        jn = (_lib.GatherDesc * 1)()
        jh = (_lib.GatherDesc * max(len(head), 1))(*head)
        r["stream_%d_us" % stream_slabs] = timeit(lambda: ops.call("gs_dense_wgrad_grouped_stream", ctypes.addressof(d0), 6, ctypes.addressof(jn), 0, s), s)
        r["stream_%d_cogather35_us" % stream_slabs] = timeit(lambda: ops.call("gs_dense_wgrad_grouped_stream", ctypes.addressof(d0), 6, ctypes.addressof(jh), len(head), s), s)
        for ks in tiled:
            for k1 in (1, 2):
                d1, keep1, _ = problems(dev, g, n0, F, ks, k1, N=1000)
                for i in range(6):           # same operands as the stream run
                    d1[i].A, d1[i].a_idx, d1[i].dZ, d1[i].a_rows = d0[i].A, d0[i].a_idx, d0[i].dZ, d0[i].a_rows
                r["tiled3_%d_%d_us" % (ks, k1)] = timeit(lambda: ops.call("gs_dense_wgrad_grouped_tiled3", ctypes.addressof(d1), 6, ctypes.addressof(jn), 0, s), s)
                if k1 == 2:
                    r["tiled3_%d_%d_cogather35_us" % (ks, k1)] = timeit(lambda: ops.call("gs_dense_wgrad_grouped_tiled3", ctypes.addressof(d1), 6, ctypes.addressof(jh), len(head), s), s)
        res[tag] = r
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
