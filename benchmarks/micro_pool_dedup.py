import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from graphsage_amd import ops
from graphsage_amd.ops import Mat
from benchmarks.micro_stream import timeit
dev = torch.device("cuda:0")
st = ops.Stream(); s = st.handle
N, F, H, m = 232965, 602, 512, 133120
g = torch.Generator(device="cpu").manual_seed(0)
X = Mat(torch.randn((N + 1, 608), generator=g).to(dev), F)
idx = torch.sort(torch.randint(0, N, (m,), generator=g, dtype=torch.int32))[0].to(dev)
W = Mat(torch.randn((F, H), generator=g).to(dev) * 0.05, H)
b = torch.zeros(H, device=dev)
out = Mat.zeros(m, H, dev)
for frac in (1.0, 0.63, 0.3, 0.05):
    cnt = torch.tensor([int(m * frac)], dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t = timeit(lambda: ops.call("gs_dense_fwd_rows_dev", X.ptr, X.ld, ops.ptr(idx), F, m, ops.ptr(cnt), W.ptr, W.ld, H, ops.ACT_RELU, ops.ptr(b), out.ptr, out.ld, s), s, iters=5, warmup=2)
    print("rows_dev frac %.2f: %.1f us" % (frac, t))
pooled = Mat.zeros(5120, H, dev); arg = torch.zeros((5120, H), dtype=torch.int32, device=dev)
t = timeit(lambda: ops.dense_pool_max_fwd(X, idx[:128000], 5120, 25, W, b, pooled, arg, stream=s), s, iters=5, warmup=2)
print("fused pool (128000 rows): %.1f us" % t)
