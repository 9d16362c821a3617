"""Diagnostics: K2 (hop-2 gather+mean of the Reddit step) throughput against resident waves per CU and loads in flight
per lane.  Each configuration runs in its own process (the probe knob is read once):
    for c in 8,0 8,40000 8,80000 8,160000 13,80000 25,80000 25,160000; do GS_GATHER_PROBE=$c python benchmarks/probe_gather_occupancy.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import ops  # noqa: E402
from graphsage_amd.ops import Mat  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    N, F, n, s = 232965, 602, 5120, 25
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    X = Mat.zeros(N + 1, F, dev, ld_multiple=32)
    X.buf[:N, :F].normal_(generator=g)
    idx = torch.randint(0, N, (n * s,), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
    out = Mat.zeros(n, F, dev)
    st = ops.Stream()
    for _ in range(5):
        ops.gather_mean_fwd(X, idx, n, s, out=out, stream=st.handle)
    e0, e1 = ops.Event(), ops.Event()
    e0.record(st.handle)
    for _ in range(30):
        ops.gather_mean_fwd(X, idx, n, s, out=out, stream=st.handle)
    e1.record(st.handle)
    us = e0.elapsed_ms(e1) / 30 * 1e3
    bytes_ = n * s * F * 4 + n * s * 4 + n * F * 4
    u, lds = (os.environ.get("GS_GATHER_PROBE", "8,0") + ",0").split(",")[:2]
    lds = int(lds)
    blocks = 8 if lds == 0 else min(8, 160 * 1024 // max(lds, 1))
    print("U=%s lds pad %6d B (<= %d workgroups = %2d waves per CU): %6.1f us  %5.2f TB/s algorithmic" % (u, lds, blocks, 4 * blocks, us, bytes_ / us / 1e6))


if __name__ == "__main__":
    main()
