"""CPU tests of the host-side logic: the C-ABI library loads and exports every declared symbol, the C++ CSR
builder, adjacency semantics of the minibatch iterator, data-parallel sharding, the reference-format reader."""
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    from graphsage_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "graphsage_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), "header declares %s but the library does not export it" % name
    assert sorted(declared) == _lib.EXPORTED_SYMBOLS, set(declared) ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.gs_abi_version() == _lib.GS_ABI_VERSION


def test_product_path_refuses_cpu_tensors():
    import torch
    from graphsage_amd import _lib, ops
    with pytest.raises(_lib.GraphsageAmdError):
        ops.ptr(torch.zeros(4))
    if not torch.cuda.is_available():
        from graphsage_amd import engine
        with pytest.raises(_lib.GraphsageAmdError):
            engine.Engine()


def test_error_reporting_through_c_abi():
    from graphsage_amd import _lib
    lib = _lib.load()
    rc = lib.gs_reduce_slabs(None, 1, 0, 1, 1, 4, 0.0, None, 0, None, 0, 0, None)
    assert rc == -1 and b"gs_reduce_slabs" in lib.gs_last_error()


def test_cpp_csr_builder_matches_numpy():
    from graphsage_amd.utils import build_csr, build_csr_numpy
    rng = np.random.RandomState(0)
    n = 500
    src = rng.randint(0, n, size=4000).astype(np.int32)
    dst = rng.randint(0, n, size=4000).astype(np.int32)
    keep = rng.rand(4000) < 0.7
    for k in (None, keep):
        r1, c1 = build_csr(n, src, dst, k)
        r2, c2 = build_csr_numpy(n, src, dst, k)
        assert np.array_equal(r1, r2) and np.array_equal(c1, c2)
    r, c = build_csr(3, np.array([0], np.int32), np.array([0], np.int32))
    assert r.tolist() == [0, 1, 1, 1] and c.tolist() == [0]          # self loop stored once
    r, c = build_csr(3, np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert r.tolist() == [0, 0, 0, 0] and c.size == 0                 # empty graph


def test_minibatch_iterator_reference_semantics():
    from graphsage_amd.minibatch import NodeMinibatchIterator
    from graphsage_amd.utils import synthetic_graph
    G = synthetic_graph(n_nodes=600, feat_dim=8, num_classes=5, avg_degree=8, seed=1)
    ph = {k: k for k in ("batch", "labels", "batch_size", "dropout")}
    it = NodeMinibatchIterator(G, None, ph, None, G.num_classes, batch_size=64, max_degree=6)
    N = G.n_nodes
    no_train = G.val_mask | G.test_mask
    assert it.adj.shape == (N + 1, 6) and (it.adj[N] == N).all()
    assert (it.adj[:N][no_train] == N).all()                          # minibatch.py:232-233
    rp, col = it.train_csr
    for i in np.where(~no_train)[0][:100]:
        nb = col[rp[i]:rp[i + 1]]
        assert not no_train[nb].any()                                 # train_removed edges are filtered (:234-236)
        assert it.deg[i] == len(nb)
        row = it.adj[i]
        assert (row == N).all() if len(nb) == 0 else set(row).issubset(set(nb))
        if len(nb) > 6:
            assert len(set(row)) == 6
    rpt, colt = it.test_csr
    assert rpt[-1] > rp[-1]                                           # test adjacency has all edges
    assert set(it.train_nodes).isdisjoint(set(np.where(no_train)[0]))
    assert (it.deg[it.train_nodes] > 0).all()                         # :214-215
    it.shuffle()
    seen = []
    while not it.end():
        fd, labels = it.next_minibatch_feed_dict()
        assert labels.shape[1] == G.num_classes and fd["batch_size"] == len(fd["batch"])
        seen.extend(fd["batch"].tolist())
    assert sorted(seen) == sorted(it.train_nodes.tolist())            # ragged last batch included (:302-307)
    fd, lab, done, subset = it.incremental_node_val_feed_dict(64, 0)
    assert len(subset) == min(64, len(it.val_nodes))


def test_shard_order_covers_global_batches():
    from graphsage_amd.distributed import shard_order
    order = np.arange(1000)
    parts = [shard_order(order, r, 4, 32) for r in range(4)]
    assert all(len(p) == len(parts[0]) for p in parts) and len(parts[0]) % 32 == 0
    step0 = np.concatenate([p[:32] for p in parts])
    assert np.array_equal(step0, order[:128])                         # global batch i = concat of rank batches


def test_load_data_reference_format(tmp_path):
    from graphsage_amd.utils import load_data
    prefix = str(tmp_path / "toy")
    nodes = [{"id": "n%d" % i, "val": i == 3, "test": i == 4} for i in range(5)]
    links = [{"source": 0, "target": 1}, {"source": 1, "target": 2}, {"source": 2, "target": 3}, {"source": 3, "target": 4}]
    json.dump({"directed": False, "graph": {}, "nodes": nodes, "links": links, "multigraph": False}, open(prefix + "-G.json", "w"))
    json.dump({"n%d" % i: i for i in range(5)}, open(prefix + "-id_map.json", "w"))
    json.dump({"n%d" % i: i % 2 for i in range(5)}, open(prefix + "-class_map.json", "w"))
    np.save(prefix + "-feats.npy", np.arange(15, dtype=np.float32).reshape(5, 3))
    G = load_data(prefix)
    assert G.n_nodes == 5 and G.src.tolist() == [0, 1, 2, 3] and G.dst.tolist() == [1, 2, 3, 4]
    assert G.train_removed.tolist() == [False, False, True, True]      # utils.py:55-60
    assert abs(G.feats[:3].mean()) < 1e-6                              # scaler fit on train rows only (:62-68)
    assert G.label_matrix().shape == (6, 2)
    # no feature file -> identity features only (utils.py:41-43): feats is None, the model then needs identity_dim > 0
    os.remove(prefix + "-feats.npy")
    G2 = load_data(prefix)
    assert G2.feats is None and G2.padded_features() is None and G2.n_nodes == 5


def test_load_data_id_map_walks_and_removed_nodes(tmp_path):
    """A non-identity id map, string node ids, a node without val/test annotations (removed, utils.py:43-50) and a
    walks file in ORIGINAL ids (mapped through id_map, minibatch.py:116-118; pairs naming removed/unknown nodes are
    dropped, :64-66)."""
    from graphsage_amd.minibatch import EdgeMinibatchIterator, NodeMinibatchIterator
    from graphsage_amd.utils import load_data
    prefix = str(tmp_path / "toy2")
    names = ["a", "b", "c", "d", "e", "f"]
    rows = {"a": 4, "b": 0, "c": 5, "d": 1, "e": 3, "f": 2}                       # NOT the identity
    nodes = [{"id": nm, "val": nm == "d", "test": nm == "e"} for nm in names]
    del nodes[5]["val"]                                                           # "f" lacks an annotation -> removed
    links = [{"source": 0, "target": 1}, {"source": 1, "target": 2}, {"source": 2, "target": 3},
             {"source": 3, "target": 4}, {"source": 4, "target": 5}, {"source": 0, "target": 5}]
    json.dump({"directed": False, "graph": {}, "nodes": nodes, "links": links, "multigraph": False}, open(prefix + "-G.json", "w"))
    json.dump(rows, open(prefix + "-id_map.json", "w"))
    json.dump({nm: i % 2 for i, nm in enumerate(names)}, open(prefix + "-class_map.json", "w"))
    np.save(prefix + "-feats.npy", np.arange(18, dtype=np.float32).reshape(6, 3))
    with open(prefix + "-walks.txt", "w") as fp:
        fp.write("a\tb\nb\tc\na\tf\nzz\ta\nc\ta\n")
    G = load_data(prefix, load_walks=True)
    assert G.n_nodes == 6 and G.present.tolist() == [True, True, False, True, True, True]
    assert G.val_mask.tolist() == [False, True, False, False, False, False] and G.test_mask[3]
    # edges touching the removed node "f" (row 2) are gone; the others are in id-map rows
    assert sorted(zip(G.src.tolist(), G.dst.tolist())) == [(0, 5), (1, 3), (4, 0), (5, 1)]
    assert G.walks.tolist() == [[4, 0], [0, 5], [5, 4]]                            # a-f and zz-a dropped
    assert G.node_ids[4] == "a" and G.id_map["c"] == 5
    ph = {}
    it = NodeMinibatchIterator(G, None, ph, None, G.num_classes, batch_size=2, max_degree=3)
    assert sorted(it.train_nodes.tolist()) == [0, 4, 5] and 2 not in it.nodes.tolist()   # b, a, c; "f" is nowhere
    assert (it.adj[2] == 6).all() and (it.test_adj[2] == 6).all()
    from graphsage_amd.models import Placeholder
    ph = {'batch1': Placeholder('b1'), 'batch2': Placeholder('b2'), 'batch_size': Placeholder('bs')}
    eit = EdgeMinibatchIterator(G, None, ph, context_pairs=G.walks, batch_size=2, max_degree=3)
    assert sorted(map(tuple, eit.train_edges.tolist())) == [(0, 5), (4, 0), (5, 4)] and 2 not in eit.nodes.tolist()


def test_bench_json_strings_format():
    """bench.py's metric / workload strings build for every mode (a formatting slip there would lose the round's
    bench line; the GPU part of bench.py cannot run in the CPU suite)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for argv in ([], ["--model", "graphsage_maxpool"], ["--unsupervised"], ["--gpus", "8"]):
        args = bench.parse_args(argv)
        metric, workload = bench.describe(args, args.feat_dim, args.samples_1, args.samples_2, args.batch_size, args.gpus)
        assert "25x10" in metric and ("unsupervised" in metric) == args.unsupervised
        assert "N=232965" in workload and args.model in workload
        assert ("RCCL" in workload) == (args.gpus > 1)
    args = bench.parse_args(["--workload", "rmat"])                       # BASELINE configs[4]
    assert (args.nodes, args.feat_dim, args.classes, args.samples_1, args.samples_2) == (10000000, 256, 64, 15, 10)
    metric, workload = bench.describe(args, args.feat_dim, args.samples_1, args.samples_2, args.batch_size, 1)
    assert "RMAT 10M-node/200M-edge" in metric and "15x10" in metric and "E=200000000" in workload


def test_single_hip_runtime_mapped():
    """Loading the C-ABI library before torch must not map a second libamdhip64 (torch bundles its own copy; two HIP
    runtimes in one process cannot share streams or memory).  _lib.load() imports torch first for that reason."""
    import subprocess
    import sys
    code = ("from graphsage_amd import _lib; _lib.load(build_if_missing=False); import torch; "
            "paths = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}); print(paths)")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         cwd=os.path.join(os.path.dirname(__file__), ".."))
    assert out.returncode == 0, out.stderr[-2000:]
    paths = eval(out.stdout.strip().splitlines()[-1])
    assert len(paths) == 1, paths


def test_split_gather_jobs_pointer_arithmetic():
    """ops.split_gather_jobs: the tail job covers the last rows of the largest job with idx / out / self pointers
    advanced by whole rows; head + tail partition the rows (pure host logic on descriptor structs)."""
    from graphsage_amd import _lib, ops

    def job(n, s, idx=0x1000, out=0x9000, self_src=None, self_idx=None, ldo=608, ld_self=608):
        j = _lib.GatherDesc()
        j.X, j.idx, j.out, j.self_src, j.self_idx = 0x100, idx, out, self_src, self_idx
        j.ldx, j.ld_self, j.ldo, j.n, j.s, j.d = 608, ld_self, ldo, n, s, 602
        return j

    small, big = job(512, 10, idx=0x2000, out=0x5000), job(5120, 25)
    head, tail = ops.split_gather_jobs([small, big], 0.7)
    assert len(head) == 2 and len(tail) == 1
    n_head = int(5120 * 0.7)
    assert head[0].n == 512 and head[1].n == n_head and tail[0].n == 5120 - n_head
    assert tail[0].idx == 0x1000 + 4 * n_head * 25 and tail[0].out == 0x9000 + 4 * n_head * 608
    assert head[1].idx == 0x1000 and head[1].out == 0x9000 and tail[0].s == 25 and tail[0].d == 602
    # GCN jobs carry self rows: through an index vector (advance the indices) or dense (advance the rows)
    _, t = ops.split_gather_jobs([job(100, 5, self_src=0x7000, self_idx=0x8000)], 0.25)
    assert t[0].self_idx == 0x8000 + 4 * 25 and t[0].self_src == 0x7000 and t[0].n == 75
    _, t = ops.split_gather_jobs([job(100, 5, self_src=0x7000, ld_self=256)], 0.25)
    assert t[0].self_src == 0x7000 + 4 * 25 * 256 and not t[0].self_idx
    # degenerate fractions
    h, t = ops.split_gather_jobs([big], 1.0)
    assert len(h) == 1 and not t
    h, t = ops.split_gather_jobs([small, big], 0.0)
    assert len(h) == 1 and h[0].n == 512 and t[0].n == 5120 and t[0].idx == 0x1000
    assert ops.split_gather_jobs([], 0.5) == ([], [])


def test_rmat_generator_on_cpu():
    """The R-MAT generator is plain torch: its CSR invariants hold on the CPU device too (BASELINE configs[4])."""
    import torch
    from graphsage_amd.utils import rmat_csr_device
    rowptr, col = rmat_csr_device(3000, 60000, torch.device("cpu"), seed=9)
    rp, c = rowptr.numpy(), col.numpy()
    assert rp[0] == 0 and rp[-1] == 60000 and np.all(np.diff(rp) >= 0) and c.min() >= 0 and c.max() < 3000
    deg = np.diff(rp)
    assert deg.max() > 10 * deg.mean() and (deg == 0).any()       # skewed: hubs and isolated nodes


def test_driver_f1_matches_sklearn():
    """supervised_train.calc_f1's NumPy micro / macro F1 == sklearn.metrics.f1_score (what the reference calls)."""
    from sklearn import metrics
    from graphsage_amd.supervised_train import f1_micro_macro
    rng = np.random.default_rng(4)
    for C, n in ((41, 512), (7, 33), (121, 256)):
        t = rng.integers(0, C, n)
        p = np.where(rng.random(n) < 0.6, t, rng.integers(0, C, n))
        p[p == 3] = 5                                            # a class that is never predicted, one never true
        mic, mac = f1_micro_macro(t, p, False)
        assert abs(mic - metrics.f1_score(t, p, average="micro")) < 1e-12
        assert abs(mac - metrics.f1_score(t, p, average="macro")) < 1e-12
        yt = (rng.random((n, C)) > 0.7).astype(np.float32)
        yp = np.where(rng.random((n, C)) < 0.8, yt, 1 - yt)
        yp[:, 0] = 0; yt[:, 1] = 0; yp[:, 1] = 0                 # never-predicted and all-absent columns
        mic, mac = f1_micro_macro(yt, yp, True)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert abs(mic - metrics.f1_score(yt, yp, average="micro")) < 1e-12
            assert abs(mac - metrics.f1_score(yt, yp, average="macro")) < 1e-12


def test_descriptor_struct_layouts_match_the_library():
    """Every ctypes mirror of a descriptor struct of include/graphsage_amd.h has the library's sizeof() (checked at load
    time too: _lib.load() raises on a mismatch instead of letting a kernel read a shifted struct)."""
    import ctypes
    from graphsage_amd import _lib
    lib = _lib.load()
    sizes = (ctypes.c_int32 * 16)()
    n = lib.gs_abi_struct_sizes(sizes, 16)
    mirrors = [_lib.GatherDesc, _lib.WgradDesc, _lib.VarDesc, _lib.FanoutDesc, _lib.TailDesc, _lib.Dropout, _lib.PullDesc,
               _lib.LpTailDesc]
    assert n == len(mirrors)
    assert [ctypes.sizeof(c) for c in mirrors] == list(sizes[:n])
    # a drifted mirror is caught by load()
    saved, _lib._lib = _lib._lib, None
    fields = _lib.WgradDesc
    try:
        class Drifted(ctypes.Structure):
            """struct gs_wgrad_desc (drifted)"""
            _fields_ = fields._fields_[:-1]
        _lib.WgradDesc = Drifted
        with pytest.raises(_lib.GraphsageAmdError, match="struct layout mismatch"):
            _lib.load()
    finally:
        _lib.WgradDesc = fields
        _lib._lib = saved


def test_tiled3_slab_policy_cuts_a_pass_into_one_round():
    """engine.tiled3_slab_policy (host logic of gs_dense_wgrad_grouped_tiled3's launch): one round of workgroups where the pass fits,
    slices of <= 1024 rows always, slab capacity respected."""
    from graphsage_amd.engine import tiled3_slab_policy
    # the Reddit step: two 602 x 128 problems over 5632 rows (10 tiles each), layer 1 (4 tiles x 2), head (4), bias (1) over 512 rows
    probs = [(10, 5632, 64), (10, 5632, 64), (4, 512, 64), (4, 512, 64), (4, 512, 64), (1, 512, 64)]
    ks = tiled3_slab_policy(probs, 256)
    assert ks == [11, 11, 1, 1, 1, 1] and sum(t * k for (t, _, _), k in zip(probs, ks)) == 233
    # the unsupervised step: 11,484 rows -> 12 slices of 960 rows (<= 1024), one round
    ks = tiled3_slab_policy([(10, 11484, 64), (10, 11484, 64), (4, 1044, 64), (4, 1044, 64)], 256)
    assert ks[0] == ks[1] == 12 and all((n + k - 1) // k <= 1024 for (_, n, _), k in zip([(10, 11484, 64)] * 2, ks[:2]))
    assert sum(t * k for t, k in zip((10, 10, 4, 4), ks)) <= 256
    # more work than one round of 1024-row slices holds: the row limit wins over the round
    ks = tiled3_slab_policy([(40, 16000, 64)], 256)
    assert ks == [16] and (16000 + 15) // 16 <= 1024
    # a nearly full slab arena caps the count (the caller has checked cap >= ceil(n / 1024))
    assert tiled3_slab_policy([(10, 5632, 6), (10, 5632, 64)], 256) [0] == 6
    # tiny problems: one slice each
    assert tiled3_slab_policy([(1, 33, 64), (1, 7, 64)], 256) == [1, 1]
    # the RMAT step (F = 256: 4 tiles per layer-0 problem): at most 24 slabs -- what the optimizer launch sums in ONE round trip
    ks = tiled3_slab_policy([(4, 5632, 64), (4, 5632, 64), (4, 512, 64), (4, 512, 64), (2, 512, 64), (1, 512, 64)], 256)
    assert max(ks) <= 24 and ks[0] == ks[1] == 24
