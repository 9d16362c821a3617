"""Calibration of rocprofv3 FETCH_SIZE for THIS kernel's access pattern (guide: MI355X_MICROARCH.md §HBM):
gather_rows over a random permutation of ALL rows of a 566 MB table (every row exactly once -> no cache
reuse possible beyond the 256 MB Infinity Cache, known byte count = rows*F*4).

    rocprofv3 --pmc FETCH_SIZE -d out -o cal -- python benchmarks/calibrate_fetch.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import ops  # noqa: E402
from graphsage_amd.ops import Mat  # noqa: E402

dev = torch.device("cuda:0")
N, F = 232966, 602
X = Mat(torch.randn((N, 608), device=dev), F)
perm = torch.randperm(N, device=dev).to(torch.int32)
out = Mat.zeros(N, F, dev, 32)
torch.cuda.synchronize()
for _ in range(5):
    ops.gather_rows(X, perm, out=out)
torch.cuda.synchronize()
print("known read bytes per launch:", N * F * 4, "(+ %d B of ids); written: %d" % (N * 4, N * F * 4))
