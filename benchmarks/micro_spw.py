"""Micro-benchmark: the sparse (arg-max) weight gradient of the max-pool MLP at the Reddit step's two shapes (5120 groups x 25 and
512 x 10, d = 602, hidden = 512), back-to-back launches, HIP events.  The form is chosen by the environment of the process:
    python benchmarks/micro_spw.py                     # the default (LDS-DMA pipeline)
    GS_SPW_NO_DMA=1 python benchmarks/micro_spw.py      # the register-staged form of rounds 2-4
    GS_SPW_DIAG=1|2|6|14|22 python benchmarks/micro_spw.py   # diagnostics (wrong values): DMA only | compute only | ... without the
                                                          # barrier | ... row reads only | ... FMAs only"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import ops  # noqa: E402
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
    N, F, H = 232965, 602, 512
    g = torch.Generator(device="cpu").manual_seed(0)
    X = Mat(torch.randn((N + 1, 608), generator=g).to(dev), F)
    X.buf[:, F:] = 0
    res = {"env": {k: v for k, v in os.environ.items() if k.startswith("GS_SPW")}}
    for tag, n, ns, k in (("hop2_5120x25", 5120, 25, 48), ("hop1_512x10", 512, 10, 16)):
        ids = torch.randint(0, N, (n * ns,), generator=g, dtype=torch.int32).to(dev)
        arg = torch.randint(0, ns, (n, H), generator=g, dtype=torch.int32).to(dev)
        if os.environ.get("MICRO_ARG_CONST"):          # every column picks row 0: the LDS reads are pure broadcasts
            arg.zero_()
        dpm = Mat(torch.randn((n, H), generator=g).to(dev), H)
        slabs = torch.zeros((k, F, H), device=dev)
        torch.cuda.synchronize()
        us = timeit(lambda: ops.maxpool_sparse_wgrad(X, ids, n, ns, arg, dpm, H, k, slabs.data_ptr(), H, stream=s), s)
        alg = n * ns * F * 4 + 10 * n * H * 8 + k * F * H * 4        # gathered rows once + arg-max / values once per feature block + slabs
        torch.cuda.synchronize()
        import hashlib
        res[tag] = {"us": us, "algorithmic_GB_s": alg / us * 1e-3, "sha": hashlib.sha256(slabs.cpu().numpy().tobytes()).hexdigest()[:12]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
