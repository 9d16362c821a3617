"""Where a workgroup of split16_dma_fwd_kernel spends its cycles (-DS16_TIMELINE build, loaded with GS_LIB):
    bash benchmarks/probes/build_variant.sh s16_tl gs_split16.hip -DS16_TIMELINE
    GS_LIB=benchmarks/probes/_lib/libgs_s16_tl.so python -m benchmarks.timeline_split16"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import _lib, ops  # noqa: E402
from graphsage_amd.ops import Mat  # noqa: E402


def main():
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
    for _ in range(30):                                   # warm clocks; the stamps of the LAST launch are read
        ops.call("gs_dense_fwd_rows_split16", ops.ptr(X2), ops.ptr(rexp), ops.ptr(ids), F, rows, ops.ptr(cnt), ops.ptr(W2), H,
                 ops.ACT_RELU, ops.ptr(b), out.ptr, out.ld, None, 0, s)
    st.sync()
    lib = _lib.load()
    n = 256 * 128
    buf = (ctypes.c_ulonglong * n)()
    lib.gs_debug_s16_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.gs_debug_s16_timeline(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 128).astype(np.int64)
    stages = (F + 31) // 32
    ent, req, bar0, loop_end, done = (t[:, k] for k in range(5))
    st_ = t[:, 8:8 + 4 * stages].reshape(256, stages, 4)
    med = lambda a: float(np.median(a))
    print("shader-clock cycles, median over 256 workgroups (first round of the launch):")
    print("  entry -> first stages requested %.0f | -> first barrier passed %.0f | K loop %.0f | epilogue %.0f | total %.0f"
          % (med(req - ent), med(bar0 - req), med(loop_end - bar0), med(done - loop_end), med(done - ent)))
    first = st_[:, :, 0] - np.concatenate([bar0[:, None], st_[:, :-1, 3]], axis=1)
    wait = st_[:, :, 1] - st_[:, :, 0]
    bar = st_[:, :, 2] - st_[:, :, 1]
    second = st_[:, :, 3] - st_[:, :, 2]
    print("  per stage (median over workgroups and stages): first half issued %.0f | counted wait %.0f | barrier %.0f | second half issued %.0f"
          % (med(first), med(wait), med(bar), med(second)))
    for k in range(stages):
        print("  stage %2d: first %5.0f wait %5.0f barrier %5.0f second %5.0f" % (k, med(first[:, k]), med(wait[:, k]), med(bar[:, k]), med(second[:, k])))


if __name__ == "__main__":
    main()
