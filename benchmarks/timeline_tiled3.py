"""Where a workgroup of sage_tiled3_fwd_kernel spends its cycles (-DF3_TIMELINE build, loaded with GS_LIB):
    bash benchmarks/probes/build_variant.sh f3_tl gs_split.hip -DF3_TIMELINE
    GS_LIB=benchmarks/probes/_lib/libgs_f3_tl.so python -m benchmarks.timeline_tiled3 [F]"""
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
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 602
    N, B, s2, D = 232965, 512, 10, 128
    g = torch.Generator(device="cpu").manual_seed(0)
    X = Mat(torch.randn((N + 1, ((F + 31) // 32) * 32), generator=g).to(dev), F)
    n = B * (1 + s2)
    ids = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
    means = Mat.zeros(n, F, dev, 32)
    means.buf[:, :F].normal_()
    Ws = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    Wn = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    out = Mat.zeros(n, 2 * D, dev)
    e0, e1 = ops.Event(), ops.Event()
    for _ in range(30):
        ops.sage_dense_fwd_tiled3(X, ids, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, [], stream=s)
    e0.record(s)
    for _ in range(20):
        ops.sage_dense_fwd_tiled3(X, ids, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, [], stream=s)
    e1.record(s)
    st.sync()
    print("launch: %.2f us (HIP events, 20 back-to-back launches)" % (e0.elapsed_ms(e1) / 20 * 1e3))
    lib = _lib.load()
    nw = 256 * 2 * 96
    buf = (ctypes.c_ulonglong * nw)()
    lib.gs_debug_f3_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.gs_debug_f3_timeline(buf, nw) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 2, 96).astype(np.int64)
    nwg = ((n + 63) // 64) * 2
    t = t[:min(nwg, 256)]
    stages = ((F + 31) // 32 + 3) // 4 * 4
    med = lambda a: float(np.median(a))
    for w in (0, 1):
        a = t[:, w]
        print("wave %d (K half %d), shader-clock cycles, median over %d workgroups:" % (4 * w, w, len(a)))
        print("  entry -> requested %.0f | -> prologue barrier %.0f | K loop %.0f (%.0f per stage) | K halves summed %.0f | stores %.0f | total %.0f"
              % (med(a[:, 1] - a[:, 0]), med(a[:, 2] - a[:, 1]), med(a[:, 3] - a[:, 2]), med(a[:, 3] - a[:, 2]) / stages,
                 med(a[:, 4] - a[:, 3]), med(a[:, 5] - a[:, 4]) if w == 0 else 0.0, med((a[:, 5] if w == 0 else a[:, 4]) - a[:, 0])))
        if w == 0:
            wall = (a[:, 7] - a[:, 6]) / 100.0
            print("  wall clock per workgroup: median %.2f us (min %.2f, max %.2f); first entry -> last exit %.2f us; entries spread over %.2f us"
                  % (med(wall), wall.min(), wall.max(), (a[:, 7].max() - a[:, 6].min()) / 100.0, (a[:, 6].max() - a[:, 6].min()) / 100.0))
        st_ = a[:, 8:8 + 3 * stages].reshape(len(a), stages, 3)
        prev = np.concatenate([a[:, 2:3], st_[:, :-1, 2]], axis=1)
        issue, wait, bar = st_[:, :, 0] - prev, st_[:, :, 1] - st_[:, :, 0], st_[:, :, 2] - st_[:, :, 1]
        print("  per stage (median): issue %.0f | counted wait %.0f | barrier %.0f" % (med(issue), med(wait), med(bar)))
        print("  stages: " + " ".join("%d:%.0f/%.0f/%.0f" % (k, med(issue[:, k]), med(wait[:, k]), med(bar[:, k])) for k in range(stages)))


if __name__ == "__main__":
    main()
