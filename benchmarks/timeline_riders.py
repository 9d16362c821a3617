"""Diagnostics: per-wave timeline of the stream layer-0 forward kernel WITH gather riders (-DGS_TIMELINE build loaded
through GS_LIB): when do the rider waves start, how long does an item take, what happens to the host waves."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import _lib, ops  # noqa: E402
from graphsage_amd.ops import Mat  # noqa: E402


def pct(x):
    return " ".join("%5.1f" % v for v in np.percentile(x, [0, 10, 50, 90, 100]))


def main():
    dev = torch.device("cuda:0")
    st = ops.Stream()
    s = st.handle
    N, F, B, s1, s2, D = 232965, 602, 512, 25, 10, 128
    n = B + B * s2
    g = torch.Generator(device="cpu").manual_seed(0)
    X = Mat(torch.randn((N + 1, 608), generator=g).to(dev), F)
    idx2 = torch.randint(0, N, (B * s2 * s1,), generator=g, dtype=torch.int32).to(dev)
    ids_self = torch.randint(0, N, (n,), generator=g, dtype=torch.int32).to(dev)
    means = Mat.zeros(n, F, dev, 32)
    means.buf[:, :F].normal_()
    Ws = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    Wn = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    out = Mat.zeros(n, 2 * D, dev)
    m2 = Mat.zeros(B * s2, F, dev, 32)
    lib = _lib.load()
    form = 2          # the one stream form left in the library (the weight-stationary form 3: benchmarks/variants/)
    hosts = 1024 if form == 3 else ((n + 31) // 32) * (D // 64) * 2 * 4
    print("form", form)
    for frac in (0.0, 0.15, 0.5):
        jobs = ops.split_gather_jobs([ops.gather_job(X, idx2, B * s2, s1, m2)], frac)[0] if frac > 0 else []
        riders = int(sum(j.n for j in jobs)) * 3 if jobs else 0
        items = min(hosts + riders, 32768)
        for _ in range(4):
            ops.sage_dense_fwd_stream(X, ids_self, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, jobs, stream=s)
            torch.cuda.synchronize()
        e0, e1 = ops.Event(), ops.Event()
        e0.record(s)
        ops.sage_dense_fwd_stream(X, ids_self, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, jobs, stream=s)
        e1.record(s)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (items * 8))()
        assert lib.gs_debug_timeline(buf, items * 8) == 0
        raw = np.frombuffer(buf, dtype=np.uint64).reshape(items, 8).astype(np.int64)
        t = (raw[:, :4] - raw[:, 0].min()) * 0.01
        hw = raw[:, 4]
        xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
        cu = ((hwid >> 8) & 0xF) | (((hwid >> 12) & 1) << 4) | (((hwid >> 13) & 7) << 5) | (xcc << 8)
        h, r = t[:hosts], t[hosts:]
        cyc = raw[:hosts, 4:]
        dt = (raw[:hosts, 2] - raw[:hosts, 1]) * 0.01
        mhz = (cyc[:, 2] - cyc[:, 1]) / np.maximum(dt, 1e-9)
        print("   shader clock inside the host waves' loops: [%s] MHz" % pct(mhz[dt > 1.0]))
        print("== riders %.2f (%d rider waves): launch %.1f us by events" % (frac, riders, e0.elapsed_ms(e1) * 1e3))
        print("   host waves: start [%s]  loop-exit [%s]  end [%s]  mean loop %.2f us" % (
            pct(h[:, 0]), pct(h[:, 2]), pct(h[:, 3]), (h[:, 2] - h[:, 1]).mean()))
        if len(r):
            print("   rider waves: start [%s]  end [%s]  item duration [%s]" % (pct(r[:, 0]), pct(r[:, 3]), pct(r[:, 3] - r[:, 0])))
            # resident rider waves over time (1 us bins)
            bins = np.arange(0, max(t[:, 3].max(), 1) + 1, 2.0)
            conc = [(int(((r[:, 0] <= b) & (r[:, 3] > b)).sum()), int(((h[:, 0] <= b) & (h[:, 3] > b)).sum())) for b in bins]
            print("   t(us): resident rider waves / host waves : " + "  ".join("%d:%d/%d" % (b, c[0], c[1]) for b, c in zip(bins, conc)))
            rc = cu[hosts:]
            per_cu = np.bincount(np.unique(rc, return_inverse=True)[1])
            print("   rider waves per CU over the launch: min %d max %d on %d CUs" % (per_cu.min(), per_cu.max(), len(per_cu)))


if __name__ == "__main__":
    main()
