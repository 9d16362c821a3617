"""Diagnostics: per-wave timeline of the stream weight-gradient kernel (needs a -DGS_TIMELINE build of the library:
    HIPCC_EXTRA=-DGS_TIMELINE python -m graphsage_amd.build --force
Prints, per problem, when its waves start / enter the steady loop / leave it / finish, relative to the first wave."""
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
    N, F, B, s2, D = 232965, 602, 512, 10, 128
    g = torch.Generator(device="cpu").manual_seed(0)
    X = Mat(torch.randn((N + 1, 608), generator=g).to(dev), F)
    n = B + B * s2
    ids_self = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
    means = Mat.zeros(n, F, dev); means.buf[:, :F].normal_()
    dz0 = Mat.zeros(n, 2 * D, dev); dz0.buf.normal_()
    NS = int(os.environ.get("SLICES", 22))
    probs = [(X, ids_self, dz0, 0, F, D, n, NS), (means, None, dz0, D, F, D, n, NS)]
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
    jn = (_lib.GatherDesc * 1)()
    lib = _lib.load()
    big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
    for mode in ("hot", "cold"):
        for _ in range(3):
            if mode == "cold":
                big.add_(1.0)
            torch.cuda.synchronize()
            ops.call("gs_dense_wgrad_grouped_stream", ctypes.addressof(arr), len(probs), ctypes.addressof(jn), 0, s)
            torch.cuda.synchronize()
        items = 2 * 10 * 2 * NS
        buf = (ctypes.c_ulonglong * (items * 8))()
        assert lib.gs_debug_timeline(buf, items * 8) == 0
        raw = np.frombuffer(buf, dtype=np.uint64).reshape(items, 8).astype(np.int64)
        t, cyc = raw[:, :4], raw[:, 4:]
        mhz = (cyc[:, 2] - cyc[:, 1]) / ((t[:, 2] - t[:, 1]) * 0.01)
        print("%s: s_memtime ticks per us inside the loop: mean %.0f (min %.0f max %.0f)" % (mode, mhz.mean(), mhz.min(), mhz.max()))
        t0 = t[:, 0].min()
        t = (t - t0) * 0.01          # us (100 MHz)
        for name, sl_ in (("gathered", slice(0, items // 2)), ("dense", slice(items // 2, items))):
            q = t[sl_]
            print("%s %-8s start %5.1f..%5.1f  loop-enter %5.1f..%5.1f  loop-exit %5.1f..%5.1f  end %5.1f..%5.1f | mean prologue %.1f loop %.1f stores %.1f" % (
                mode, name, q[:, 0].min(), q[:, 0].max(), q[:, 1].min(), q[:, 1].max(), q[:, 2].min(), q[:, 2].max(), q[:, 3].min(), q[:, 3].max(),
                (q[:, 1] - q[:, 0]).mean(), (q[:, 2] - q[:, 1]).mean(), (q[:, 3] - q[:, 2]).mean()))


if __name__ == "__main__":
    main()
