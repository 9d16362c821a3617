"""Diagnostics: per-wave timeline of the stream layer-0 forward kernel (needs a -DGS_TIMELINE build, see
timeline_wgrad.py)."""
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
    F, B, s2, D = 602, 512, 10, 128
    n = B + B * s2
    g = torch.Generator(device="cpu").manual_seed(0)
    selfd, means = Mat.zeros(n, F, dev), Mat.zeros(n, F, dev)
    selfd.buf[:, :F].normal_(); means.buf[:, :F].normal_()
    Ws = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    Wn = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    out = Mat.zeros(n, 2 * D, dev)
    lib = _lib.load()
    big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
    tn = 2
    items = ((n + 31) // 32) * (D // 64) * 2 * 4          # tiles x 4 K-quarter waves
    for mode in ("hot", "cold"):
        for _ in range(3):
            if mode == "cold":
                big.add_(1.0)
            torch.cuda.synchronize()
            ops.sage_dense_fwd_stream(selfd, None, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, [], stream=s)
            torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (items * 8))()
        assert lib.gs_debug_timeline(buf, items * 8) == 0
        raw = np.frombuffer(buf, dtype=np.uint64).reshape(items, 8).astype(np.int64)
        t, cyc = raw[:, :4], raw[:, 4:]
        hw = cyc[:, 0]
        xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
        cu = ((hwid >> 8) & 0xF) | (((hwid >> 12) & 1) << 4) | (((hwid >> 13) & 7) << 5) | (xcc << 8)
        simd = (hwid >> 4) & 3
        per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
        print("%s placement: %d distinct CUs; waves per CU min %d max %d mean %.1f; histogram %s" % (
            mode, len(per_cu), per_cu.min(), per_cu.max(), per_cu.mean(), np.bincount(per_cu).tolist()))
        mhz = (cyc[:, 2] - cyc[:, 1]) / np.maximum((t[:, 2] - t[:, 1]) * 0.01, 1e-9)
        t = (t - t[:, 0].min()) * 0.01
        print("%s TN=%d items %d clock %.0f MHz: start %5.1f..%5.1f  loop-enter %5.1f..%5.1f  loop-exit %5.1f..%5.1f  end %5.1f..%5.1f | mean prologue %.1f loop %.1f stores %.1f" % (
            mode, tn, items, mhz.mean(), t[:, 0].min(), t[:, 0].max(), t[:, 1].min(), t[:, 1].max(), t[:, 2].min(), t[:, 2].max(), t[:, 3].min(), t[:, 3].max(),
            (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean()))


if __name__ == "__main__":
    main()
