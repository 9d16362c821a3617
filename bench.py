#!/usr/bin/env python
"""Headline benchmark: sampled-edges/sec of the full GraphSAGE-mean TRAINING step
(sample -> gather+mean -> dense -> loss -> backward -> [all-reduce] -> clip+Adam) on a synthetic
Reddit-shaped graph (N=232,965, F=602, C=41, fan-out 25x10, batch 512 per GPU) -- BASELINE.json configs[1].

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One JSON line on rank 0.  `value` = (edges all ranks sampled in K steps) / (max-over-ranks wall time),
features/CSR/labels/epoch order resident in HBM before the timed region.  See DESIGN.md "Measurement".
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from graphsage_amd import distributed as gsd  # noqa: E402


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(G, it, args, world, rank):
    from graphsage_amd import engine as eng
    from graphsage_amd.models import Placeholder, SAGEInfo
    from graphsage_amd.neigh_samplers import AdjInfo, CSRAdjacency, UniformNeighborSampler
    from graphsage_amd.supervised_models import SupervisedGraphsage
    eng.reset_engine()
    e = eng.get_engine()
    ph = {'labels': Placeholder('labels'), 'batch': Placeholder('batch1'), 'dropout': Placeholder('dropout', 0.),
          'batch_size': Placeholder('batch_size')}
    adj_info = AdjInfo(CSRAdjacency(it.train_csr[0], it.train_csr[1], G.n_nodes, e.device))
    sampler = UniformNeighborSampler(adj_info, seed=123)
    if args.unsupervised:
        from graphsage_amd.models import SampleAndAggregate
        ph = {'batch1': Placeholder('batch1'), 'batch2': Placeholder('batch2'), 'neg_samples': Placeholder('neg'),
              'dropout': Placeholder('dropout', 0.), 'batch_size': Placeholder('batch_size')}
        layer_infos = [SAGEInfo("node", sampler, args.samples_1, args.dim_1), SAGEInfo("node", sampler, args.samples_2, args.dim_2)]
        model = SampleAndAggregate(ph, G.padded_features(), adj_info, it.deg, layer_infos, concat=True, aggregator_type="mean",
                                   learning_rate=0.00001, weight_decay=0.0, neg_sample_size=20, world_size=world, rank=rank)
        model.row_offset = rank * (2 * args.batch_size + 20)
        return e, model, ph
    agg = {"graphsage_mean": "mean", "gcn": "gcn", "graphsage_maxpool": "maxpool", "graphsage_meanpool": "meanpool"}[args.model]
    mult = 2 if agg == "gcn" else 1          # supervised_train.py:175-176
    layer_infos = [SAGEInfo("node", sampler, args.samples_1, mult * args.dim_1),
                   SAGEInfo("node", sampler, args.samples_2, mult * args.dim_2)]
    model = SupervisedGraphsage(G.num_classes, ph, G.padded_features(), adj_info, it.deg, layer_infos,
                                concat=(agg != "gcn"), aggregator_type=agg, sigmoid_loss=False,
                                learning_rate=0.01, weight_decay=0.0, world_size=world, rank=rank)
    model.row_offset = rank * args.batch_size
    return e, model, ph


def describe(args, F, s1, s2, B, world):
    """(metric, config.workload) strings of the JSON line."""
    mode = "unsupervised" if args.unsupervised else "supervised"
    if args.workload == "rmat":
        graph = "RMAT synthetic graph (N=%d, E=%d directed, a/b/c/d=0.57/0.19/0.19/0.05, F=%d U(-1,1), C=%d random labels)" % (
            args.nodes, args.rmat_edges, F, args.classes)
        size = lambda v: ("%dM" % (v // 1000000)) if v >= 1000000 else str(v)
        shape = "RMAT %s-node/%s-edge" % (size(args.nodes), size(args.rmat_edges))
    else:
        graph = "Reddit-shaped synthetic graph (N=%d, F=%d, C=%d, avg_degree=%d)" % (args.nodes, F, args.classes, args.avg_degree)
        shape = "Reddit-shaped"
    fmt = ("%s, " + mode + " %s, fan-out %dx%d, batch %d "
           "per GPU, dims %d/%d, full training step (sample+gather+fwd+bwd%s+clip+Adam), hipGraph replay, next-step "
           "gather co-scheduled with the layer-0 contraction and the weight-gradient launch (horizontal fusion)")
    workload = fmt % (graph, args.model, s1, s2, B, args.dim_1, args.dim_2, "+RCCL all-reduce" if world > 1 else "")
    metric = "sampled-edges/sec, %s %s %s fan-out %dx%d" % (shape, mode, args.model, s1, s2)
    return metric, workload


def build_rmat(args, world, rank):
    """BASELINE configs[4]: RMAT graph + U(-1,1) features + random labels generated directly in HBM (replicated per
    GPU), supervised graphsage_mean.  Returns (engine, model, placeholders, epoch order)."""
    from graphsage_amd import engine as eng
    from graphsage_amd.models import Placeholder, SAGEInfo
    from graphsage_amd.neigh_samplers import AdjInfo, CSRAdjacency, UniformNeighborSampler
    from graphsage_amd.ops import Mat
    from graphsage_amd.supervised_models import SupervisedGraphsage
    from graphsage_amd.utils import rmat_csr_device
    eng.reset_engine()
    e = eng.get_engine()
    N, F, C = args.nodes, args.feat_dim, args.classes
    rowptr, col = rmat_csr_device(N, args.rmat_edges, e.device, seed=123)
    g = torch.Generator(device=e.device)
    g.manual_seed(123)
    feats = Mat.zeros(N + 1, F, e.device, ld_multiple=32)            # row N = zero pad row
    feats.buf[:N, :F].uniform_(-1.0, 1.0, generator=g)
    labels = Mat.zeros(N + 1, C, e.device)
    cls = torch.randint(0, C, (N,), device=e.device, generator=g)
    labels.buf[torch.arange(N, device=e.device), cls] = 1.0
    order = torch.randperm(N, device=e.device, generator=g).to(torch.int32).cpu().numpy()
    torch.cuda.synchronize()
    ph = {'labels': Placeholder('labels'), 'batch': Placeholder('batch1'), 'dropout': Placeholder('dropout', 0.),
          'batch_size': Placeholder('batch_size')}
    adj_info = AdjInfo(CSRAdjacency.from_device(rowptr, col, N))
    sampler = UniformNeighborSampler(adj_info, seed=123)
    layer_infos = [SAGEInfo("node", sampler, args.samples_1, args.dim_1), SAGEInfo("node", sampler, args.samples_2, args.dim_2)]
    model = SupervisedGraphsage(C, ph, feats, adj_info, None, layer_infos, concat=True, aggregator_type="mean",
                                sigmoid_loss=False, learning_rate=0.01, weight_decay=0.0, world_size=world, rank=rank)
    model.row_offset = rank * args.batch_size
    return e, model, ph, order, labels, int(col.numel())


def parse_args(argv=None):
    if os.environ.get("GS_FAULT_DUMP_S"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["GS_FAULT_DUMP_S"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch_size", type=int, default=512)
    ap.add_argument("--samples_1", type=int, default=25)
    ap.add_argument("--samples_2", type=int, default=10)
    ap.add_argument("--dim_1", type=int, default=128)
    ap.add_argument("--dim_2", type=int, default=128)
    ap.add_argument("--nodes", type=int, default=232965)
    ap.add_argument("--feat_dim", type=int, default=602)
    ap.add_argument("--classes", type=int, default=41)
    ap.add_argument("--avg_degree", type=int, default=50)
    ap.add_argument("--model", default="graphsage_mean",
                    help="graphsage_mean (headline, BASELINE configs[1]) | graphsage_maxpool (configs[2]) | gcn | graphsage_meanpool")
    ap.add_argument("--workload", default="reddit", choices=["reddit", "rmat"],
                    help="reddit: BASELINE configs[1-3] (default, the metric's configuration); rmat: configs[4] "
                         "(N=10^7, E=2*10^8, F=256, C=64, fan-out 15x10; sets --nodes/--feat_dim/--classes/--samples_1)")
    ap.add_argument("--rmat-edges", dest="rmat_edges", type=int, default=200000000)
    ap.add_argument("--unsupervised", action="store_true",
                    help="BASELINE configs[3]: unsupervised graphsage_mean on random-walk pairs (20 negatives, xent, MRR)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    ap.add_argument("--steps-per-launch", dest="steps_per_launch", type=int, default=8,
                    help="consecutive training steps replayed per hipGraph launch (single GPU)")
    args = ap.parse_args(argv)
    if args.workload == "rmat":
        given = set(a.split("=")[0] for a in (argv if argv is not None else sys.argv[1:]))
        for flag, val in (("--nodes", 10000000), ("--feat_dim", 256), ("--classes", 64), ("--samples_1", 15)):
            if flag not in given:
                setattr(args, flag[2:], val)
        if args.unsupervised or args.model != "graphsage_mean":
            ap.error("--workload rmat is the supervised graphsage_mean configuration")
    return args


def main():
    args = parse_args()

    rank, local_rank, world = gsd.init_from_env()
    if world != args.gpus and world > 1:
        log("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())

    from graphsage_amd import ops
    from graphsage_amd.minibatch import NodeMinibatchIterator
    from graphsage_amd.utils import reddit_shaped

    t0 = time.time()
    B, s1, s2, F = args.batch_size, args.samples_1, args.samples_2, args.feat_dim
    G = it = None
    if args.workload == "rmat":
        e, model, ph, epoch, label_table, n_edges = build_rmat(args, world, rank)
        if rank == 0:
            log("RMAT graph+model ready in %.1fs: N=%d edges=%d params=%d" % (time.time() - t0, args.nodes, n_edges, e.n_trainable()))
        order = gsd.shard_order(epoch, rank, world, B) if world > 1 else epoch
        model.attach_device_epoch(order, label_table)
    else:
        G = reddit_shaped(avg_degree=args.avg_degree, seed=123, n_nodes=args.nodes, feat_dim=args.feat_dim,
                          num_classes=args.classes)
        it = NodeMinibatchIterator(G, None, {}, None, G.num_classes, batch_size=args.batch_size, max_degree=128,
                                   build_padded=False)
        e, model, ph = build_model(G, it, args, world, rank)
        if rank == 0:
            log("graph+model ready in %.1fs: N=%d edges=%d train=%d params=%d" %
                (time.time() - t0, G.n_nodes, len(G.src), len(it.train_nodes), e.n_trainable()))

    if args.workload == "rmat":
        pass
    elif args.unsupervised:
        from graphsage_amd.utils import run_random_walks
        pairs = run_random_walks(it.train_csr[0], it.train_csr[1], it.train_nodes, max_pairs=2000000, seed=123)
        pairs = np.random.RandomState(123).permutation(pairs)
        if world > 1:
            n_steps = len(pairs) // (B * world)
            pairs = pairs[: n_steps * B * world].reshape(n_steps, world, B, 2)[:, rank].reshape(-1, 2)
        model.attach_device_pairs(pairs)
        args.model = "graphsage_mean"
    else:
        epoch = np.random.RandomState(123).permutation(it.train_nodes)
        order = gsd.shard_order(epoch, rank, world, B) if world > 1 else epoch
        model.attach_device_epoch(order, it.label_matrix)
    if world > 1:
        model.grad_hook = gsd.GradAllReduce(e)
    elif os.environ.get("GS_PROBE_DP_SCHEDULE"):
        # diagnostic: run the data-parallel step schedule (backward graph | hook | optimizer graph, one step per launch)
        # on one GPU with a no-op hook, to see its host-side cost without RCCL
        model.grad_hook = lambda m: None

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    spl = args.steps_per_launch

    def run_steps(k):
        model.train_steps_device(B, k, steps_per_launch=spl)

    # the first two executions of a graph key are eager + capture: warm both the k-step and the 1-step graphs
    run_steps(max(args.warmup, 2 * spl + 6))
    barrier()
    t0 = time.time()
    run_steps(args.steps)
    e.sync()
    torch.cuda.synchronize()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=e.device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        torch.distributed.barrier()
    loss_after = model._fetch_unsup(B)[0] if args.unsupervised else model._fetch(B)[0]

    edges_per_step = ((2 * B + 20) if args.unsupervised else B) * (s2 + s2 * s1)
    value = edges_per_step * world * args.steps / dt

    metric, workload = describe(args, F, s1, s2, B, world)
    result = {
        "metric": metric,
        "value": value, "unit": "sampled-edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "global_batch": B * world, "parallelism": "dp%d" % world, "loss_after": loss_after},
    }

    # ---------------- roofline of the dominant kernel (K2 hop-2 gather+mean), HIP events on the engine stream.
    # Every rank runs the region (the interleaved training steps all-reduce under N>1); rank 0 reports.
    if True:
        n2 = ((2 * B + 20) if args.unsupervised else B) * s2
        idx2 = model.samples1[2]
        mean2 = ops.Mat.zeros(n2, F, e.device)
        torch.cuda.synchronize()
        iters = max(20, min(args.steps, 200))
        evs = [(ops.Event(), ops.Event()) for _ in range(iters)]
        for a, b in evs:                     # interleave K2 launches with full steps: same cache state as training
            run_steps(1)
            a.record(e.stream)
            ops.gather_mean_fwd(model.features, idx2, n2, s1, out=mean2, stream=e.stream)
            b.record(e.stream)
        e.sync()
        k2_us = float(np.mean([a.elapsed_ms(b) for a, b in evs])) * 1e3
        alg_bytes = n2 * s1 * F * 4 + n2 * s1 * 4 + n2 * F * 4    # rows*F*4 + ids + mean write
        achieved = alg_bytes / (k2_us * 1e-6) / 1e9
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_k2_pmc.json")
        if os.path.exists(pmc) and (B, s1, s2, F) == (512, 25, 10, 602) and not args.unsupervised:
            with open(pmc) as fpm:     # HBM bytes per launch from the committed rocprofv3 --pmc passes of this command
                traffic = json.load(fpm)["traffic_bytes_per_launch"]
            traffic_src = "profiles/r01_k2_pmc.json (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes; FETCH_SIZE x1.974, calibrated)"
        result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                              "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                              "kernel": "gather_mean_kernel<8> (K2, hop-2: [%d x %d] rows of %d fp32)" % (n2, s1, F),
                              "avg_launch_us": k2_us, "algorithmic_bytes_per_launch": alg_bytes}
        if args.model in ("graphsage_maxpool", "graphsage_meanpool") and not args.unsupervised:
            # pooling aggregators: the dominant kernel is the MLP contraction over every gathered neighbor row
            # ([n2*s1 + B*s2 rows, F] x [F, hidden], fp32 MFMA) -> report ITS roofline; the K2 numbers stay as "gather"
            agg0 = model.aggregators[0]
            mlp = agg0.mlp_layers[0]
            rows_all = n2 * s1 + B * s2
            ids_all = torch.as_strided(model.samples1[1], (rows_all,), (1,))     # hop-1 and hop-2 ids are adjacent
            H = ops.Mat.zeros(rows_all, agg0.hidden_dim, e.device)
            torch.cuda.synchronize()
            evs = [(ops.Event(), ops.Event()) for _ in range(max(10, min(args.steps, 50)))]
            for a, b in evs:
                run_steps(1)
                a.record(e.stream)
                ops.sage_dense_fwd(None, None, model.features, ids_all, rows_all, None, mlp.vars['weights'].value,
                                   agg0.hidden_dim, False, ops.ACT_RELU, mlp.vars['bias'].value.buf, H, stream=e.stream)
                b.record(e.stream)
            e.sync()
            mlp_us = float(np.mean([a.elapsed_ms(b) for a, b in evs])) * 1e3
            flops = 2.0 * rows_all * F * agg0.hidden_dim
            tf = flops / (mlp_us * 1e-6) / 1e12
            result["roofline_gather"] = result["roofline"]
            result["roofline"] = {"bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
                                  "traffic": None,
                                  "kernel": "gemm_f32_mfma_kernel<128,128> (pooling MLP forward: [%d gathered rows x %d] . "
                                            "[%d x %d], v_mfma_f32_32x32x2_f32)" % (rows_all, F, F, agg0.hidden_dim),
                                  "avg_launch_us": mlp_us, "algorithmic_flops_per_launch": flops}
        if rank == 0 and not args.no_cpu_baseline and world == 1 and args.model == "graphsage_mean" and G is not None:
            from oracle.cpu_baseline import time_cpu_baseline
            from graphsage_amd.utils import padded_from_csr
            tc = time.time()
            adj, _ = padded_from_csr(it.train_csr[0], it.train_csr[1], G.n_nodes, 128, np.random.RandomState(123))
            cb = time_cpu_baseline(G.padded_features(), adj, it.label_matrix, it.train_nodes, G.num_classes,
                                   batch_size=B, num_samples=(s1, s2), dims=(F, args.dim_1, args.dim_2),
                                   budget_s=args.cpu_budget_s)
            cb.pop("s_per_step", None)
            result["cpu_baseline"] = cb
            log("cpu baseline took %.1fs" % (time.time() - tc))
        if rank == 0:
            print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
