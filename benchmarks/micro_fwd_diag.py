"""Diagnostic: the layer-0 stream contraction at the Reddit step shapes with the library named by GS_LIB (operand loads
compiled out in the variants of benchmarks/probes/build_variant.sh): alone (hot / cold operands) and with 15 % / 50 % of
the next step's gather riding.     GS_LIB=... python benchmarks/micro_fwd_diag.py <tag>"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import ops  # noqa: E402
from graphsage_amd.ops import Mat  # noqa: E402
from benchmarks.micro_stream import timeit  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "base"
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
    means = Mat.zeros(n, F, dev, 32)
    means.buf[:, :F].normal_()
    Ws = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    Wn = Mat(torch.randn((F, D), generator=g).to(dev) * 0.05, D)
    out = Mat.zeros(n, 2 * D, dev)
    m2, m1 = Mat.zeros(B * s2, F, dev, 32), Mat.zeros(B, F, dev, 32)
    jobs_all = [ops.gather_job(X, idx2, B * s2, s1, m2), ops.gather_job(X, idx1, B, s2, m1)]
    res = {"tag": tag, "form": 2}          # the one stream form left in the library (the weight-stationary form 3: benchmarks/variants/)

    def fwd(jobs):
        return lambda: ops.sage_dense_fwd_stream(X, ids_self, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, jobs, stream=s)
    res["fwd_alone_hot_us"] = timeit(fwd([]), s)
    for frac in (0.15, 0.3, 0.5):
        head, _ = ops.split_gather_jobs(jobs_all, frac)
        res["fwd_riders_%.2f_us" % frac] = timeit(fwd(head), s)
    big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
    tot = 0.0
    for _ in range(8):
        big.add_(1.0)
        torch.cuda.synchronize()
        e0, e1 = ops.Event(), ops.Event()
        e0.record(s); fwd([])(); e1.record(s)
        tot += e0.elapsed_ms(e1) * 1e3
    res["fwd_alone_cold_us"] = tot / 8
    res["gather_alone_us"] = timeit(lambda: [ops.gather_mean_fwd(X, idx2, B * s2, s1, out=m2, stream=s),
                                             ops.gather_mean_fwd(X, idx1, B, s2, out=m1, stream=s)], s)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
