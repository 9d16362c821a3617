"""Runs the stream / tiled contraction kernels alone a few times (for rocprofv3 --pmc)."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import _lib, ops
from graphsage_amd.ops import Mat
dev = torch.device("cuda:0")
F, B, s2, D = 602, 512, 10, 128
n = B + B * s2
selfd, means = Mat.zeros(n, F, dev), Mat.zeros(n, F, dev)
selfd.buf[:, :F].normal_(); means.buf[:, :F].normal_()
Ws = Mat(torch.randn((F, D), device=dev) * 0.05, D); Wn = Mat(torch.randn((F, D), device=dev) * 0.05, D)
out = Mat.zeros(n, 2 * D, dev)
dz0 = Mat.zeros(n, 2 * D, dev); dz0.buf.normal_()
torch.cuda.synchronize()
s = ops.current_stream()
probs = [(selfd, dz0, 0), (means, dz0, D)]
arr = (_lib.WgradDesc * 2)()
keep = []
for i, (A, Z, col0) in enumerate(probs):
    sl = torch.zeros(22 * F * D, device=dev); keep.append(sl)
    arr[i].A, arr[i].a_idx, arr[i].dZ, arr[i].slabs = A.ptr, None, Z.ptr, sl.data_ptr()
    arr[i].lda, arr[i].ldz, arr[i].ld_slab, arr[i].n = A.ld, Z.ld, D, n
    arr[i].d, arr[i].col0, arr[i].out_dim, arr[i].n_slabs = F, col0, D, 22
jn = (_lib.GatherDesc * 1)()
for _ in range(10):
    ops.sage_dense_fwd_stream(selfd, None, means, n, Ws, Wn, D, ops.ACT_RELU, None, out, [], stream=s)
    ops.sage_dense_fwd(selfd, None, means, None, n, Ws, Wn, D, True, ops.ACT_RELU, None, out, stream=s)
    ops.call("gs_dense_wgrad_grouped_stream", ctypes.addressof(arr), 2, ctypes.addressof(jn), 0, s)
    ops.call("gs_dense_wgrad_grouped", ctypes.addressof(arr), 2, s)
torch.cuda.synchronize()
