#!/usr/bin/env python
"""Headline benchmark: sampled-edges/sec of the full GraphSAGE-mean TRAINING step
(sample -> gather+mean -> dense -> loss -> backward -> [all-reduce] -> clip+Adam) on a synthetic
Reddit-shaped graph (N=232,965, F=602, C=41, average degree 492, fan-out 25x10, batch 512 per GPU) --
BASELINE.json configs[1] -- plus the micro-F1 half of the metric and short driver-visible runs of configs[2..4].

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

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TF = 157.3    # v_mfma_f32_32x32x2_f32 dense peak
MFMA_16BIT_PEAK_TF = 2500.0  # v_mfma_f32_32x32x16_{f16,bf16} dense peak (MI355X_MICROARCH.md; the 5 PF headline figure is 2:1 sparse)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def placeholders(unsup=False):
    from graphsage_amd.models import Placeholder
    if unsup:
        return {'batch1': Placeholder('batch1'), 'batch2': Placeholder('batch2'), 'neg_samples': Placeholder('neg'),
                'dropout': Placeholder('dropout', 0.), 'batch_size': Placeholder('batch_size')}
    return {'labels': Placeholder('labels'), 'batch': Placeholder('batch1'), 'dropout': Placeholder('dropout', 0.),
            'batch_size': Placeholder('batch_size')}


def build_model(DG, args, world, rank, model_name, unsupervised=False, sampler_seed=123, sampler_law=None, sigmoid=False):
    """A fresh engine + model on the device-resident graph DG (features / CSR / labels are shared, not copied)."""
    from graphsage_amd import engine as eng
    from graphsage_amd.models import SAGEInfo, SampleAndAggregate
    from graphsage_amd.neigh_samplers import AdjInfo, CSRAdjacency, UniformNeighborSampler
    from graphsage_amd.supervised_models import SupervisedGraphsage
    eng.reset_engine()
    e = eng.get_engine()
    train_adj = CSRAdjacency.from_device(DG.train_csr[0], DG.train_csr[1], DG.n_nodes)
    adj_info = AdjInfo(train_adj)
    sampler = UniformNeighborSampler(adj_info, seed=sampler_seed, law=sampler_law or args.sampler_law, max_degree=args.max_degree)
    ph = placeholders(unsupervised)
    if unsupervised:
        layer_infos = [SAGEInfo("node", sampler, args.samples_1, args.dim_1), SAGEInfo("node", sampler, args.samples_2, args.dim_2)]
        model = SampleAndAggregate(ph, DG.feats, adj_info, DG.deg, layer_infos, concat=True, aggregator_type="mean",
                                   learning_rate=0.00001, weight_decay=0.0, neg_sample_size=20, world_size=world, rank=rank)
        model.row_offset = rank * (2 * args.batch_size + 20)
        return e, model, ph, adj_info
    agg = {"graphsage_mean": "mean", "gcn": "gcn", "graphsage_maxpool": "maxpool", "graphsage_meanpool": "meanpool"}[model_name]
    mult = 2 if agg == "gcn" else 1          # supervised_train.py:175-176
    layer_infos = [SAGEInfo("node", sampler, args.samples_1, mult * args.dim_1),
                   SAGEInfo("node", sampler, args.samples_2, mult * args.dim_2)]
    model = SupervisedGraphsage(DG.num_classes, ph, DG.feats, adj_info, DG.deg, layer_infos,
                                concat=(agg != "gcn"), aggregator_type=agg, sigmoid_loss=sigmoid,
                                learning_rate=0.01, weight_decay=0.0, world_size=world, rank=rank)
    model.row_offset = rank * args.batch_size
    return e, model, ph, adj_info


def describe(args, F, s1, s2, B, world):
    """(metric, config.workload) strings of the JSON line."""
    mode = "unsupervised" if args.unsupervised else "supervised"
    if args.workload == "rmat":
        graph = "RMAT synthetic graph (N=%d, E=%d directed, a/b/c/d=0.57/0.19/0.19/0.05, F=%d U(-1,1), C=%d random labels)" % (
            args.nodes, args.rmat_edges, F, args.classes)
        size = lambda v: ("%dM" % (v // 1000000)) if v >= 1000000 else str(v)
        shape = "RMAT %s-node/%s-edge" % (size(args.nodes), size(args.rmat_edges))
    else:
        graph = "Reddit-shaped synthetic graph (N=%d, F=%d, C=%d, avg_degree=%d, planted-community labels, feat_signal=%g)" % (
            args.nodes, F, args.classes, args.avg_degree, args.feat_signal)
        shape = "Reddit-shaped"
    fmt = ("%s, " + mode + " %s, fan-out %dx%d, batch %d "
           "per GPU, dims %d/%d, full training step (sample+gather+fwd+bwd%s+clip+Adam), hipGraph replay, the next step's "
           "gather+mean and the sampler of the step after ride as extra workgroups in the step's own launches "
           "(layer-0 contraction / fused tail / weight gradients / optimizer: horizontal fusion)")
    workload = fmt % (graph, args.model, s1, s2, B, args.dim_1, args.dim_2, "+RCCL all-reduce" if world > 1 else "")
    metric = "sampled-edges/sec, %s %s %s fan-out %dx%d" % (shape, mode, args.model, s1, s2)
    return metric, workload


def build_rmat(args, world, rank):
    """BASELINE configs[4]: RMAT graph + U(-1,1) features + random labels generated directly in HBM (replicated per
    GPU), supervised graphsage_mean.  Returns (engine, model, placeholders, epoch order, label table, n_edges)."""
    from graphsage_amd import engine as eng
    from graphsage_amd.models import SAGEInfo
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
    ph = placeholders()
    adj_info = AdjInfo(CSRAdjacency.from_device(rowptr, col, N))
    sampler = UniformNeighborSampler(adj_info, seed=123, law=args.sampler_law, max_degree=args.max_degree)
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
    ap.add_argument("--avg_degree", type=int, default=492, help="Reddit's real average degree (SURVEY §8d)")
    ap.add_argument("--feat_signal", type=float, default=0.02,
                    help="class-centroid scale in the synthetic features (0.5 = trivially separable; 0.02 keeps micro-F1 informative)")
    ap.add_argument("--model", default="graphsage_mean",
                    help="graphsage_mean (headline, BASELINE configs[1]) | graphsage_maxpool (configs[2]) | gcn | graphsage_meanpool")
    ap.add_argument("--workload", default="reddit", choices=["reddit", "rmat"],
                    help="reddit: BASELINE configs[1-3] (default, the metric's configuration); rmat: configs[4] "
                         "(N=10^7, E=2*10^8, F=256, C=64, fan-out 15x10; sets --nodes/--feat_dim/--classes/--samples_1)")
    ap.add_argument("--rmat-edges", dest="rmat_edges", type=int, default=200000000)
    ap.add_argument("--unsupervised", action="store_true",
                    help="BASELINE configs[3]: unsupervised graphsage_mean on random-walk pairs (20 negatives, xent, MRR)")
    ap.add_argument("--sampler_law", default=os.environ.get("GS_SAMPLER_LAW", "reference"), choices=["reference", "iid", "distinct"],
                    help="sampling law of the CSR sampler (reference = the reference's joint law on a virtual padded "
                         "[N+1, max_degree] table; iid = independent draws with replacement from the full list)")
    ap.add_argument("--max_degree", type=int, default=128, help="FLAGS.max_degree (supervised_train.py:40): width of the padded table")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU port legs (cpu_baseline and micro_f1)")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    ap.add_argument("--f1-steps", dest="f1_steps", type=int, default=100,
                    help="training steps of the micro-F1 leg (MI355X engine and CPU port, same graph / order / steps)")
    ap.add_argument("--f1-val-nodes", dest="f1_val_nodes", type=int, default=4096)
    ap.add_argument("--f1-seeds", dest="f1_seeds", type=int, default=5,
                    help="seeds (weights / epoch order / sampler stream) of the micro-F1 legs; mean +- std are reported")
    ap.add_argument("--no-aux", action="store_true", help="skip the short configs[2..4] runs reported under `aux`")
    ap.add_argument("--aux-steps", dest="aux_steps", type=int, default=40)
    ap.add_argument("--steps-per-launch", dest="steps_per_launch", type=int, default=None,
                    help="consecutive training steps replayed per hipGraph launch (default: 32 on one GPU -- measured "
                         "110.4 | 109.5 | 108.9 us/step at 8 | 16 | 32 --, 8 with the all-reduce recorded in the graph)")
    args = ap.parse_args(argv)
    if args.workload == "rmat":
        given = set(a.split("=")[0] for a in (argv if argv is not None else sys.argv[1:]))
        for flag, val in (("--nodes", 10000000), ("--feat_dim", 256), ("--classes", 64), ("--samples_1", 15)):
            if flag not in given:
                setattr(args, flag[2:], val)
        if args.unsupervised or args.model != "graphsage_mean":
            ap.error("--workload rmat is the supervised graphsage_mean configuration")
    return args


def roots_of(args):
    return (2 * args.batch_size + 20) if args.unsupervised else args.batch_size


def pmc_profile_path(args):
    """Committed rocprofv3 --pmc summary (FETCH_SIZE / WRITE_SIZE passes of THIS command) for this configuration."""
    tag = "deg%d_b%d_%dx%d_f%d" % (args.avg_degree, args.batch_size, args.samples_1, args.samples_2, args.feat_dim)
    return os.path.join(ROOT, "profiles", "k2_pmc_%s.json" % tag)


def lib_digest():
    """Source digest of the loaded library (graphsage_amd/_C/build.stamp, written by graphsage_amd.build)."""
    p = os.path.join(ROOT, "graphsage_amd", "_C", "build.stamp")
    return open(p).read().strip() if os.path.exists(p) else None


def step_traffic_profile(args):
    """Per-step fabric-side bytes from the newest committed whole-step PMC profile (profiles/r*_step_traffic.json,
    benchmarks/profile_step_traffic.sh) -- only for the headline configuration it was taken on."""
    import glob
    if args.workload != "reddit" or args.unsupervised or args.model != "graphsage_mean" or args.avg_degree != 492 \
            or (args.batch_size, args.samples_1, args.samples_2, args.feat_dim) != (512, 25, 10, 602):
        return None
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_step_traffic.json")))
    if not paths:
        return None
    with open(paths[-1]) as f:
        d = json.load(f)
    by = {}
    for l in d["launches"]:
        # the steady-state launch of a kernel is the most frequent grid; the sampler rides in the optimizer launch
        k = l["kernel"]
        if k == "sample_fanout_kernel":
            continue
        if k not in by or l["launches_seen"] > by[k]["launches_seen"]:
            by[k] = l
    total = sum((l["hbm_read_MB"] + l["hbm_write_MB"]) * 1e6 for l in by.values())
    return {"bytes_per_step": total, "path": os.path.relpath(paths[-1], ROOT), "lib_digest": d.get("lib_digest")}


def aux_dominant_profile():
    """{config: {kernel, avg_us, share}} of the aux configurations from the newest committed kernel traces
    (profiles/r*_aux_dominant.json, written by benchmarks/aux_dominant.py from the rocprofv3 --kernel-trace summaries)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_aux_dominant.json")))
    if not paths:
        return None
    with open(paths[-1]) as f:
        d = json.load(f)
    d["path"] = os.path.relpath(paths[-1], ROOT)
    return d


def ref_on_shim_profile():
    """cpu_baseline kind "reference-on-shim": the reference's OWN supervised training step (/root/reference/graphsage/
    supervised_models.py, unmodified, on the TF 1.x stand-in of the test suite) timed by benchmarks/ref_on_shim_cpu.py in the BUILD container -- the
    reference cannot travel to the GPU box, so the committed record is attached (PROFILE-SOURCED), with the torch-CPU port timed
    on the same cores beside it."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ref_on_shim_cpu.json")))
    if not paths:
        return None
    with open(paths[-1]) as f:
        d = json.load(f)
    return {"kind": "reference-on-shim", "value": d["reference_on_shim"]["value"], "unit": d["reference_on_shim"]["unit"],
            "s_per_step": d["reference_on_shim"]["s_per_step"], "cores": d["cores"], "host": d["host"],
            "port_same_cores": d["port_same_cores"], "config": d["config"], "what": d["what"], "note": d["note"],
            "source": "PROFILE-SOURCED, not measured by this run: %s (python benchmarks/ref_on_shim_cpu.py)" % os.path.relpath(paths[-1], ROOT)}


def timed_events(e, fn, iters, between=None):
    """Average duration (us) of fn() measured with HIP events on the ENGINE stream (torch.cuda.Event would only see
    torch's current stream); `between()` runs before every sample so the caches are in the training state."""
    from graphsage_amd import ops
    evs = [(ops.Event(), ops.Event()) for _ in range(iters)]
    for a, b in evs:
        if between is not None:
            between()
        a.record(e.stream)
        fn()
        b.record(e.stream)
    e.sync()
    return float(np.mean([a.elapsed_ms(b) for a, b in evs])) * 1e3


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves -- one process per GPU, the
    contract of `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` -- and hand their output through.  (Before
    round 5 this command silently ran ONE rank and printed n_gpus: 1.)"""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < args.gpus and os.environ.get("GS_DIST_BACKEND", "nccl") != "gloo":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (GS_DIST_BACKEND=gloo lets ranks share a device: a "
                         "functional check of the N > 1 path, not a measurement)" % (args.gpus, n_dev))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py --gpus %d without WORLD_SIZE: launching the ranks: %s" % (args.gpus, " ".join(cmd)))
    raise SystemExit(subprocess.call(cmd))


def main():
    args = parse_args()
    t_begin = time.time()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    rank, local_rank, world = gsd.init_from_env()
    if world != args.gpus:
        # a line whose n_gpus differs from what was asked for would be recorded under the wrong N
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- launch with --nproc-per-node equal to --gpus" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    device = torch.device("cuda:%d" % torch.cuda.current_device())

    from graphsage_amd import ops
    from graphsage_amd.utils import random_walk_pairs_device, reddit_shaped_device

    t0 = time.time()
    B, s1, s2, F = args.batch_size, args.samples_1, args.samples_2, args.feat_dim
    DG = None
    if args.workload == "rmat":
        e, model, ph, epoch, label_table, n_edges = build_rmat(args, world, rank)
        if rank == 0:
            log("RMAT graph+model ready in %.1fs: N=%d edges=%d params=%d" % (time.time() - t0, args.nodes, n_edges, e.n_trainable()))
        order = gsd.shard_order(epoch, rank, world, B) if world > 1 else epoch
        model.attach_device_epoch(order, label_table)
    else:
        DG = reddit_shaped_device(device, n_nodes=args.nodes, feat_dim=F, num_classes=args.classes,
                                  avg_degree=args.avg_degree, seed=123, feat_signal=args.feat_signal)
        e, model, ph, adj_info = build_model(DG, args, world, rank, args.model, args.unsupervised)
        if rank == 0:
            log("graph+model ready in %.1fs: N=%d undirected edges=%d train nodes=%d (median train degree %d) params=%d" %
                (time.time() - t0, DG.n_nodes, DG.n_edges_undirected, len(DG.train_nodes),
                 int(np.median(DG.deg[DG.train_nodes])), e.n_trainable()))
        if args.unsupervised:
            pairs = random_walk_pairs_device(DG.train_csr[0], DG.train_csr[1], DG.train_nodes, max_pairs=2000000, seed=123)
            pairs = pairs.cpu().numpy()
            if world > 1:
                n_steps = len(pairs) // (B * world)
                pairs = pairs[: n_steps * B * world].reshape(n_steps, world, B, 2)[:, rank].reshape(-1, 2)
            model.attach_device_pairs(pairs)
            args.model = "graphsage_mean"
        else:
            epoch = np.random.RandomState(123).permutation(DG.train_nodes)
            order = gsd.shard_order(epoch, rank, world, B) if world > 1 else epoch
            model.attach_device_epoch(order, DG.label_table)
    dp_info = None
    if world > 1:
        model.grad_hook = gsd.make_grad_hook(e, log=log)
        if rank == 0:
            log("gradient all-reduce: %s" % type(model.grad_hook).__name__)
    elif os.environ.get("GS_PROBE_DP_SCHEDULE"):
        # diagnostic: the data-parallel step schedule on ONE GPU, the collective replaced by a wave that sleeps for the given
        # number of microseconds (0 = no-op hook)
        model.grad_hook = gsd.SpinHook(e, float(os.environ["GS_PROBE_DP_SCHEDULE"]))
    if model.grad_hook is not None:
        dp_info = model.measure_dp_allreduce(log=log if rank == 0 else None) or {}
        dp_info["allreduce"] = type(model.grad_hook).__name__
        dp_info["in_graph"] = bool(model._dp_in_graph())
        if hasattr(model.grad_hook, "ranks"):
            dp_info["rccl_ranks"] = model.grad_hook.ranks()          # ncclCommCount: what RCCL itself says it spans

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.steps_per_launch is None:
        args.steps_per_launch = 32 if world == 1 else 8
    spl = args.steps_per_launch

    def run_steps(k):
        model.train_steps_device(B, k, steps_per_launch=spl)

    # The first two executions of a graph key are eager + capture: warm the full-length and the 1-step graphs, then the
    # exact call that is timed, twice -- a K that is not a multiple of the graph length ends in a shorter graph of its own
    # (K = 20 at 8 steps per launch: 8 + 8 + 4), which would otherwise run eagerly INSIDE the timed region.
    warm_steps = max(args.warmup, 2 * min(spl, 8) + 6)
    run_steps(warm_steps)
    run_steps(args.steps)
    run_steps(args.steps)
    warm_steps += 2 * args.steps
    # The two calls above spend tens of ms on the HOST (the second one walks the whole K-step Python chain under stream
    # capture) while the device idles and drops its clocks: a timed call right behind them ran 3-4 % slow, and so did every
    # first call after an idle stretch (profiles/r05_launch_probe.txt: 2259 | 2234 | 2215 | 2199 | 2198 us for consecutive
    # launch + sync calls of the same 20-step graph).  The warm-up that counts is the one right before the timed region:
    # replay the exact timed call (launch + sync, as it is timed) for >= 100 steps.
    warm_calls = int(os.environ.get("GS_BENCH_WARM_CALLS", min(16, max(2, -(-100 // max(1, args.steps))))))
    for _ in range(warm_calls):
        run_steps(args.steps)
        e.sync()
        warm_steps += args.steps
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
    # HIP events on the engine stream around the timed call, repeated -- the device-side duration of the same K
    # steps without the host's launch / synchronisation cost that the wall clock above carries (about 7 us per step at
    # K = 20); `value` stays the wall-clock figure.  All ranks run it (the steps all-reduce under N > 1); rank 0 reports.
    ev_reps = 12
    evs = [(ops.Event(), ops.Event()) for _ in range(ev_reps)]
    for a, b in evs:
        a.record(e.stream)
        run_steps(args.steps)
        b.record(e.stream)
    e.sync()
    ev_ms = np.asarray([a.elapsed_ms(b) for a, b in evs][1:]) / args.steps
    events = {"ms_per_step_median": float(np.median(ev_ms)), "ms_per_step_p10": float(np.percentile(ev_ms, 10)),
              "ms_per_step_p90": float(np.percentile(ev_ms, 90)), "launches": int(len(ev_ms)), "steps_per_sample": args.steps,
              "basis": "hipEventElapsedTime on the engine stream around each call of the timed %d-step region" % args.steps}

    launch_probe = None
    if os.environ.get("GS_BENCH_LAUNCH_PROBE") and world == 1:
        # diagnostic (benchmarks/r5_launch_probe.sh): where the wall clock of the timed call goes on the host -- the call itself
        # (Python + hipGraphLaunch until it returns), then the wait for the stream, each from an idle device as in the timed region
        t_call, t_wait, t_sync2 = [], [], []
        for _ in range(15):
            barrier()
            ta = time.perf_counter()
            run_steps(args.steps)
            tb = time.perf_counter()
            e.sync()
            tc = time.perf_counter()
            torch.cuda.synchronize()
            td = time.perf_counter()
            t_call.append(tb - ta); t_wait.append(tc - tb); t_sync2.append(td - tc)
        med = lambda v: float(np.median(v[2:])) * 1e6
        launch_probe = {"call_us": med(t_call), "wait_us": med(t_wait), "second_sync_us": med(t_sync2),
                        "total_us": med([a + b + c for a, b, c in zip(t_call, t_wait, t_sync2)]),
                        "events_us": events["ms_per_step_median"] * args.steps * 1e3,
                        "totals_in_order_us": [round((a + b + c) * 1e6, 1) for a, b, c in zip(t_call, t_wait, t_sync2)]}
        log("launch probe (K = %d): %s" % (args.steps, json.dumps(launch_probe)))

    roots = (2 * B + 20) if args.unsupervised else B
    edges_per_step = roots * (s2 + s2 * s1)
    value = edges_per_step * world * args.steps / dt

    metric, workload = describe(args, F, s1, s2, B, world)
    result = {
        "metric": metric,
        "value": value, "unit": "sampled-edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "warmup_steps_run": warm_steps,
        "ms_per_step": dt / args.steps * 1e3, "ms_per_step_events": events, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "global_batch": B * world, "parallelism": "dp%d" % world, "loss_after": loss_after,
                   "steps_per_graph_launch": spl, "sampler_law": args.sampler_law, "max_degree": args.max_degree,
                   "allreduce": dp_info["allreduce"] if dp_info else None,
                   "rccl_ranks": dp_info.get("rccl_ranks") if dp_info else None},
    }
    if launch_probe:
        result["launch_probe"] = launch_probe
    if dp_info:
        # how the collective sits in the step: its stand-alone duration, the gather share forked beside it, and the
        # EXPOSED time = this step minus the same schedule with a no-op hook in the collective's place (all ranks swap
        # together, so no rank waits for a peer)
        real_hook = model.grad_hook
        model.grad_hook = gsd.SpinHook(e, 0.0) if getattr(real_hook, "capturable", False) else (lambda m: None)
        run_steps(2 * min(spl, 8) + 6)
        run_steps(args.steps)
        run_steps(args.steps)
        for _ in range(warm_calls):
            run_steps(args.steps)
            e.sync()
        barrier()
        t1 = time.time()
        run_steps(args.steps)
        e.sync()
        torch.cuda.synchronize()
        dt0 = time.time() - t1
        if world > 1:
            t = torch.tensor([dt0], dtype=torch.float64, device=e.device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt0 = float(t.item())
        model.grad_hook = real_hook
        dp_info["ms_per_step_without_collective"] = dt0 / args.steps * 1e3
        dp_info["exposed_allreduce_us_per_step"] = (dt - dt0) / args.steps * 1e6
        result["dp_schedule"] = dp_info

    # ---------------- roofline of the gather (K2 hop-2 gather+mean), HIP events on the engine stream.
    # Every rank runs the region (the interleaved training steps all-reduce under N>1); rank 0 reports.
    n2 = roots * s2
    idx2 = model.samples1[2]
    mean2 = ops.Mat.zeros(n2, F, e.device)
    torch.cuda.synchronize()
    iters = max(20, min(args.steps, 100))
    uniq = []

    def between():
        run_steps(1)
        if len(uniq) < 8:
            e.sync()
            uniq.append(int(torch.unique(model.samples1[2]).numel()))

    k2_us = timed_events(e, lambda: ops.gather_mean_fwd(model.features, idx2, n2, s1, out=mean2, stream=e.stream), iters, between)
    alg_bytes = n2 * s1 * F * 4 + n2 * s1 * 4 + n2 * F * 4    # rows*F*4 + ids + mean write
    achieved = alg_bytes / (k2_us * 1e-6) / 1e9
    unique_bytes = float(np.mean(uniq)) * F * 4 + n2 * s1 * 4 + n2 * F * 4
    traffic = traffic_src = None
    traffic_stale = None
    pmc = pmc_profile_path(args)
    if os.path.exists(pmc) and args.workload == "reddit" and not args.unsupervised:
        with open(pmc) as fpm:
            prof = json.load(fpm)
        traffic = prof["traffic_bytes_per_launch"]
        # the counters were collected on the library whose source digest the profile carries: a different library running
        # now (kernels edited since) makes them stale -> `frac` falls back to the live unique-row bound below
        traffic_stale = prof.get("lib_digest") != lib_digest()
        traffic_src = ("PROFILE-SOURCED, not measured by this run: %s (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of "
                       "this command on this graph; FETCH_SIZE calibrated on a known-byte gather, MI355X_MICROARCH.md HBM). "
                       "FETCH_SIZE counts at the L2's FABRIC side: reads served by the Infinity Cache (MALL) are included, so "
                       "this is an UPPER bound on the bytes that reached HBM; the unique-row bytes are the lower bound."
                       % os.path.relpath(pmc, ROOT))
    unique_gbs = unique_bytes / (k2_us * 1e-6) / 1e9
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "kernel": "gather_mean_kernel<8> (K2, hop-2: [%d x %d] rows of %d fp32), standalone launch interleaved with "
                      "training steps" % (n2, s1, F),
            "avg_launch_us": k2_us, "algorithmic_bytes_per_launch": alg_bytes,
            "frac_algorithmic": achieved / HBM_PEAK_GBS,
            "unique_row_bytes_per_launch": unique_bytes,
            "unique_rows_frac": float(np.mean(uniq)) / float(n2 * s1),
            "frac_unique_lower": min(achieved, unique_gbs) / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
            "lib_digest": lib_digest(),
            "traffic_basis": "L2 fabric-side bytes (Infinity-Cache hits included): upper bound on HBM bytes"}
    if traffic is not None and not traffic_stale:
        roof["frac"] = traffic / (k2_us * 1e-6) / 1e9 / HBM_PEAK_GBS
        roof["frac_basis"] = ("L2 fabric-side bytes (traffic, upper bound on HBM bytes) / avg_launch_us / peak; the true HBM "
                              "fraction lies in [frac_unique_lower, frac]")
    else:
        roof["frac"] = min(achieved, unique_gbs) / HBM_PEAK_GBS
        roof["frac_basis"] = ("unique-row bytes (live lower bound of the HBM-side traffic: duplicate rows of a launch can be "
                              "served by L2/MALL) / avg_launch_us / peak; no committed PMC profile of THIS library "
                              "(source digest) for this configuration")
    result["roofline"] = roof
    # ---------------- the WHOLE step against the same roof: algorithmic bytes of a step (rows*F*4 + ids +
    # mean writes, all hops) / ms_per_step, and the fabric-side counter bytes of the step's launches from the committed
    # whole-step PMC profile (their ratio = traffic the step moves beyond its algorithmic bytes)
    rows_step = roots * (1 + s2 + s2 * s1)
    alg_step = rows_step * F * 4 + roots * (s2 + s2 * s1) * 4 + roots * (1 + s2) * F * 4
    step = {"algorithmic_bytes_per_step": alg_step, "ms_per_step": dt / args.steps * 1e3,
            "achieved": alg_step / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac_algorithmic": alg_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, "counter_bytes_per_step": None}
    stp = step_traffic_profile(args)
    if stp is not None:
        step["counter_bytes_per_step"] = stp["bytes_per_step"]
        step["frac_counter"] = stp["bytes_per_step"] / (dt / args.steps) / 1e9 / HBM_PEAK_GBS
        step["wasted_traffic_ratio"] = stp["bytes_per_step"] / alg_step
        step["counter_source"] = "PROFILE-SOURCED: %s (L2 fabric-side bytes, upper bound on HBM bytes)" % stp["path"]
        step["counter_stale"] = stp.get("lib_digest") != lib_digest()
    result["roofline_step"] = step
    # the driver's parser keeps `roofline` and drops the other objects: the STEP's own fractions ride inside it too.  `frac`
    # above belongs to the stand-alone K2 launch (the kernel the north star names: at the chip's copy rate); the timed region
    # as a whole sits far lower -- these three fields say how far.
    roof["scope"] = ("frac / achieved / traffic: the stand-alone K2 launch (not part of the timed region); in_step_*: the whole "
                     "timed training step against the same 8 TB/s roof")
    roof["in_step_frac_algorithmic"] = step["frac_algorithmic"]
    roof["in_step_frac_counter"] = step.get("frac_counter")
    roof["wasted_traffic_ratio"] = step.get("wasted_traffic_ratio")
    roof["in_step_counter_stale"] = step.get("counter_stale")

    # ---------------- roofline of the launch that dominates the step: the layer-0 contraction with the next step's
    # gather+mean co-scheduled in it (both roofs at once).  The exact launch of the step is re-issued between events.
    agg0 = model.aggregators[0]
    replay = getattr(agg0, "last_fused_launch", None)
    if replay is not None:
        fn, info = replay
        us = timed_events(e, fn, iters, lambda: run_steps(1))
        gb = info["gather_bytes"] + info["gemm_bytes"]
        result["roofline_step_kernel"] = {
            "kernel": info["kernel"], "avg_launch_us": us,
            "hbm": {"algorithmic_bytes_per_launch": gb, "achieved": gb / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac_algorithmic": gb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "gather_share_of_step": info["gather_share"]},
            "mfma": {"algorithmic_flops_per_launch": info["flops"], "achieved": info["flops"] / (us * 1e-6) / 1e12,
                     "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": info["flops"] / (us * 1e-6) / 1e12 / MFMA_F32_PEAK_TF}}
        pp = info.get("piece_products", 1)
        if pp > 1:
            # the kernel computes in fp32-equivalent arithmetic on the bf16 pipe: `frac` above prices the algorithmic fp32 flops
            # against the fp32 MFMA peak (what an fp32 kernel could reach at best); these price what is ISSUED against its own pipe
            m = result["roofline_step_kernel"]["mfma"]
            m["basis"] = "algorithmic fp32 flops / fp32 MFMA peak (the kernel itself issues bf16 MFMAs: see pipe_*)"
            m["pipe"] = "bf16 (fp32 operands as three bf16 pieces, %d MFMAs per product tile)" % pp
            m["pipe_flops_issued_per_launch"] = pp * info["flops"]
            m["pipe_peak"] = MFMA_16BIT_PEAK_TF
            m["pipe_frac"] = pp * info["flops"] / (us * 1e-6) / 1e12 / MFMA_16BIT_PEAK_TF

    if args.model in ("graphsage_maxpool", "graphsage_meanpool") and not args.unsupervised:
        # pooling aggregators: the dominant kernel is the MLP contraction over every gathered neighbor row
        # ([n2*s1 + B*s2 rows, F] x [F, hidden], fp32 MFMA) -> report ITS roofline; the K2 numbers stay as "gather"
        mlp = agg0.mlp_layers[0]
        rows_all = n2 * s1 + B * s2
        ids_all = torch.as_strided(model.samples1[1], (rows_all,), (1,))     # hop-1 and hop-2 ids are adjacent
        H = ops.Mat.zeros(rows_all, agg0.hidden_dim, e.device)
        torch.cuda.synchronize()
        mlp_us = timed_events(e, lambda: ops.sage_dense_fwd(None, None, model.features, ids_all, rows_all, None,
                                                            mlp.vars['weights'].value, agg0.hidden_dim, False, ops.ACT_RELU,
                                                            mlp.vars['bias'].value.buf, H, stream=e.stream),
                              max(10, min(args.steps, 50)), lambda: run_steps(1))
        flops = 2.0 * rows_all * F * agg0.hidden_dim
        tf = flops / (mlp_us * 1e-6) / 1e12
        result["roofline_gather"] = result["roofline"]
        result["roofline"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                              "frac": tf / MFMA_F32_PEAK_TF, "traffic": None,
                              "kernel": "gemm_f32_mfma_kernel<128,128> (pooling MLP forward: [%d gathered rows x %d] . "
                                        "[%d x %d], v_mfma_f32_32x32x2_f32)" % (rows_all, F, F, agg0.hidden_dim),
                              "avg_launch_us": mlp_us, "algorithmic_flops_per_launch": flops}

    headline = (rank == 0 and world == 1 and args.model == "graphsage_mean" and DG is not None and not args.unsupervised)
    # ---------------- micro-F1 half of the metric + the CPU baseline (see f1_legs)
    if headline and not args.no_cpu_baseline:
        tc = time.time()
        del model                                   # its workspaces are not needed any more (the legs build their own)
        f1, cb = f1_legs(DG, args, B, s1, s2, F, spl, seeds=args.f1_seeds, steps=args.f1_steps, n_val=args.f1_val_nodes)
        result["cpu_baseline"] = cb
        ros = ref_on_shim_profile()
        if ros is not None:
            cb["reference_on_shim"] = ros
        result["micro_f1"] = f1
        log("cpu baseline + micro-F1 legs took %.1fs: %s" % (time.time() - tc, json.dumps(
            {k: (round(v["mean"], 4), round(v["std"], 4)) for k, v in f1["legs"].items()})))

    # ---------------- driver-visible short runs of BASELINE configs[2], [3], [4] (own models, same process)
    if headline and not args.no_aux:
        result["aux"] = run_aux(DG, args, B, s1, s2)

    if rank == 0:
        result["bench_wall_s"] = time.time() - t_begin
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def f1_legs(DG, args, B, s1, s2, F, spl, seeds=5, steps=100, n_val=4096):
    """micro-F1 half of the metric as a MEASUREMENT: for each of `seeds` seeds (weights, epoch order, sampler stream)
    four models train for `steps` steps on the same graph / order / learning rate FROM THE SAME INITIAL WEIGHTS and are
    validated on the same held-out nodes on the full (test) adjacency (supervised_train.py:280):
      cpu_port                   torch-CPU port of the reference graph: padded table (minibatch.py:227-259), one shared
                                 column permutation per sampler call (neigh_samplers.py:24-29)
      mi355x_padded_same_draws   the MI355X engine on THE SAME padded tables with THE SAME permutations injected
                                 (gs_sample_padded): isolates the kernels' numerics from the sampler law
      mi355x_csr_reference_law   the MI355X engine, native CSR sampler with law="reference" (the reference's joint law
                                 on a virtual padded table, device-epoch hipGraph path)
      mi355x_csr_iid             the MI355X engine, law="iid" (independent draws with replacement from the full list)
      mi355x_csr_distinct        the MI355X engine, law="distinct" (per-row draws without replacement, max_degree cap)
    "mi355x" in the result is the leg of the law the timed path runs (--sampler_law, default reference).
    Returns (micro_f1 dict with per-leg mean/std/values and paired differences vs cpu_port, cpu_baseline dict)."""
    from graphsage_amd.neigh_samplers import AdjInfo, CSRAdjacency, PaddedAdjacency
    from oracle import graphsage_oracle as orc
    from oracle.cpu_baseline import port_micro_f1, time_cpu_baseline
    feats_h, adj_h, test_adj_h, labels_h = DG.host_view(max_degree=args.max_degree)
    val = DG.val_nodes[:n_val].astype(np.int32)
    n_vb = (len(val) + B - 1) // B
    legs = {"cpu_port": [], "mi355x_padded_same_draws": [], "mi355x_csr_reference_law": [], "mi355x_csr_iid": [],
            "mi355x_csr_distinct": []}
    cb = None
    wall = {}

    def eval_gpu(model, ph, adj_info, test_adj, perms=None):
        train_adj = adj_info.current
        adj_info.assign(test_adj)                                   # supervised_train.py:280
        preds = []
        for i, a in enumerate(range(0, len(val), B)):
            b = val[a:a + B]
            if perms is not None:
                model.layer_infos[0].neigh_sampler.inject_perms(perms[i])
            _, p = model.eval_step({ph['batch']: b, ph['labels']: labels_h[b], ph['batch_size']: len(b)})
            preds.append(p)
        adj_info.assign(train_adj)                                  # supervised_train.py:285
        return orc.calc_f1_micro(labels_h[val], np.vstack(preds), False)

    def init_from_port(e, model, port):
        for agg, (wn, ws) in zip(model.aggregators, port.params):
            agg.vars['neigh_weights'].assign(wn.detach().numpy())
            agg.vars['self_weights'].assign(ws.detach().numpy())
        model.node_pred.vars['weights'].assign(port.W.detach().numpy())
        model.node_pred.vars['bias'].assign(port.b.detach().numpy())
        e.snapshot_initial_parameters()

    for sd in range(seeds):
        seed = 123 + sd
        epoch = np.random.RandomState(seed).permutation(DG.train_nodes)
        prng = np.random.RandomState(1000 + seed)
        perms_train = [[prng.permutation(args.max_degree) for _ in range(2)] for _ in range(steps)]
        perms_val = [[prng.permutation(args.max_degree) for _ in range(2)] for _ in range(n_vb)]
        # ---- the CPU port (timed: the first seed's run is the cpu_baseline leg)
        from oracle.cpu_baseline import CpuSupervisedMean
        init = CpuSupervisedMean(feats_h[:1], adj_h[:1], [F, args.dim_1, args.dim_2], DG.num_classes, [s1, s2], seed=seed)
        t0 = time.time()
        c, port = time_cpu_baseline(feats_h, adj_h, labels_h, DG.train_nodes, DG.num_classes, batch_size=B,
                                    num_samples=(s1, s2), dims=(F, args.dim_1, args.dim_2), order=epoch, fixed_steps=steps,
                                    return_model=True, seed=seed, perms=perms_train)
        legs["cpu_port"].append(port_micro_f1(port, test_adj_h, labels_h, val, batch_size=B, perms=perms_val))
        wall.setdefault("cpu_port", []).append(time.time() - t0)
        if cb is None:
            c.pop("s_per_step", None)
            c.pop("steps_trained", None)
            cb = c
        # ---- MI355X, padded tables + the same injected permutations (host-fed eager steps: the permutation is a host input)
        t0 = time.time()
        e, model, ph, adj_info = build_model(DG, args, 1, 0, "graphsage_mean", sampler_seed=seed)
        init_from_port(e, model, init)
        adj_info.assign(PaddedAdjacency(adj_h, e.device))
        sampler = model.layer_infos[0].neigh_sampler
        for t in range(steps):
            b = epoch[t * B:(t + 1) * B]
            sampler.inject_perms(perms_train[t])
            model.train_step({ph['batch']: b, ph['labels']: labels_h[b], ph['batch_size']: len(b)}, fetch=False)
        legs["mi355x_padded_same_draws"].append(eval_gpu(model, ph, adj_info, PaddedAdjacency(test_adj_h, e.device), perms_val))
        wall.setdefault("mi355x_padded_same_draws", []).append(time.time() - t0)
        del model
        # ---- MI355X, CSR sampler, device epoch + hipGraphs (the timed path), both laws
        for law, leg in (("reference", "mi355x_csr_reference_law"), ("iid", "mi355x_csr_iid"), ("distinct", "mi355x_csr_distinct")):
            t0 = time.time()
            e, model, ph, adj_info = build_model(DG, args, 1, 0, "graphsage_mean", sampler_seed=seed, sampler_law=law)
            init_from_port(e, model, init)
            model.attach_device_epoch(epoch, DG.label_table)
            model.train_steps_device(B, steps, steps_per_launch=spl)
            e.sync()
            test_adj = CSRAdjacency.from_device(DG.test_csr[0], DG.test_csr[1], DG.n_nodes)
            legs[leg].append(eval_gpu(model, ph, adj_info, test_adj))
            wall.setdefault(leg, []).append(time.time() - t0)
            del model
    out_legs, paired = {}, {}
    ref = np.asarray(legs["cpu_port"])
    for k, v in legs.items():
        v = np.asarray(v)
        out_legs[k] = {"mean": float(v.mean()), "std": float(v.std(ddof=1)) if len(v) > 1 else 0.0, "values": [float(x) for x in v],
                       "wall_s_per_seed": float(np.mean(wall[k]))}
        if k != "cpu_port":
            d = v - ref
            paired[k] = {"mean": float(d.mean()), "stderr": float(d.std(ddof=1) / np.sqrt(len(d))) if len(d) > 1 else None}
    timed = {"reference": "mi355x_csr_reference_law", "iid": "mi355x_csr_iid", "distinct": "mi355x_csr_distinct"}[args.sampler_law]
    f1 = {"mi355x": out_legs[timed]["mean"], "mi355x_leg": timed, "cpu_port": out_legs["cpu_port"]["mean"], "seeds": seeds,
          "train_steps": steps, "val_nodes": int(len(val)), "legs": out_legs, "paired_delta_vs_cpu_port": paired,
          "note": "per seed: same synthetic graph, epoch order, steps, lr and INITIAL WEIGHTS for every leg; validation on the "
                  "full (test) adjacency; mean / sample std over seeds; paired_delta = leg - cpu_port per seed (mean, standard "
                  "error).  mi355x_padded_same_draws isolates numerics (same tables, same permutations); the two CSR legs "
                  "differ from it only in the sampler law."}
    return f1, cb


def toy_ppi_leg(args):
    """BASELINE configs[0]: `python -m graphsage.supervised_train --train_prefix ./example_data/ppi --model graphsage_mean
    --sigmoid` (example_supervised.sh:1) on a toy-PPI-SHAPED synthetic graph (N = 14,755 from example_data/toy-ppi-id_map.json,
    F = 50, C = 121 multi-hot; the reference's G / feats / class_map blobs are stripped), the driver's defaults: 10 epochs,
    B = 512, fan-out 25x10, dims 128/128, max_degree 128, lr 0.01.  Two legs FROM THE SAME INITIAL WEIGHTS on the same epoch
    orders: the torch-CPU port (the configuration's own device in BASELINE.json: 'reference TF-CPU path') and the MI355X engine
    (CSR sampler, reference law, device epoch + hipGraphs); validation micro-F1 (sigmoid, threshold 0.5) on the test adjacency."""
    from graphsage_amd import engine as eng
    from graphsage_amd.models import SAGEInfo
    from graphsage_amd.neigh_samplers import AdjInfo, CSRAdjacency, UniformNeighborSampler
    from graphsage_amd.supervised_models import SupervisedGraphsage
    from graphsage_amd.utils import build_csr, padded_from_csr, synthetic_graph
    from oracle import graphsage_oracle as orc
    from oracle.cpu_baseline import CpuSupervisedMean, port_micro_f1
    N, F, C, B, s1, s2, D, MAXD, EPOCHS = 14755, 50, 121, 512, 25, 10, 128, 128, 10
    G = synthetic_graph(n_nodes=N, feat_dim=F, num_classes=C, avg_degree=28, seed=123, multilabel=True)
    nt = G.val_mask | G.test_mask
    rp, col = build_csr(N, G.src, G.dst, keep=~(nt[G.src] | nt[G.dst]))
    rp_t, col_t = build_csr(N, G.src, G.dst)
    rng = np.random.RandomState(123)
    adj, deg = padded_from_csr(rp, col, N, MAXD, rng)
    test_adj, _ = padded_from_csr(rp_t, col_t, N, MAXD, rng)
    feats, labels = G.padded_features(), G.label_matrix()
    train_nodes = np.nonzero(~nt & (deg[:N] > 0))[0].astype(np.int32)
    val = np.nonzero(G.val_mask)[0].astype(np.int32)
    per_epoch = len(train_nodes) // B
    order = np.concatenate([rng.permutation(train_nodes)[: per_epoch * B] for _ in range(EPOCHS)]).astype(np.int32)
    steps = per_epoch * EPOCHS
    # ---- CPU port: timed while it trains
    t0 = time.time()
    init = CpuSupervisedMean(feats[:1], adj[:1], [F, D, D], C, [s1, s2], seed=123)        # the port's initial weights
    port = CpuSupervisedMean(feats, adj, [F, D, D], C, [s1, s2], sigmoid_loss=True, seed=123)
    ts = []
    for i in range(steps):
        b = order[i * B:(i + 1) * B]
        t1 = time.time()
        port.train_step(b, labels[b])
        ts.append(time.time() - t1)
    cpu_f1 = port_micro_f1(port, test_adj, labels, val, batch_size=B, sigmoid=True)
    cpu_s = float(np.median(ts[2:]))
    cpu_wall = time.time() - t0
    # ---- MI355X
    eng.reset_engine()
    e = eng.get_engine()
    adj_info = AdjInfo(CSRAdjacency(rp, col, N, e.device))
    sampler = UniformNeighborSampler(adj_info, seed=123, law=args.sampler_law, max_degree=MAXD)
    ph = placeholders()
    layer_infos = [SAGEInfo("node", sampler, s1, D), SAGEInfo("node", sampler, s2, D)]
    model = SupervisedGraphsage(C, ph, feats, adj_info, deg, layer_infos, concat=True, aggregator_type="mean",
                                sigmoid_loss=True, learning_rate=0.01, weight_decay=0.0)
    for agg, (wn, ws) in zip(model.aggregators, init.params):
        agg.vars['neigh_weights'].assign(wn.detach().numpy())
        agg.vars['self_weights'].assign(ws.detach().numpy())
    model.node_pred.vars['weights'].assign(init.W.detach().numpy())
    model.node_pred.vars['bias'].assign(init.b.detach().numpy())
    e.snapshot_initial_parameters()
    model.attach_device_epoch(order, labels)
    spl = min(args.steps_per_launch or 32, 32)
    model.train_steps_device(B, steps, steps_per_launch=spl)
    e.sync()
    train_adj = adj_info.current
    adj_info.assign(CSRAdjacency(rp_t, col_t, N, e.device))                 # supervised_train.py:280
    preds = []
    for a in range(0, len(val), B):
        b = val[a:a + B]
        _, p = model.eval_step({ph['batch']: b, ph['labels']: labels[b], ph['batch_size']: len(b)})
        preds.append(p)
    adj_info.assign(train_adj)
    gpu_f1 = orc.calc_f1_micro(labels[val], np.vstack(preds), True)
    K = 64
    for _ in range(2):
        model.train_steps_device(B, K, steps_per_launch=spl)
    e.sync()
    t0 = time.time()
    model.train_steps_device(B, K, steps_per_launch=spl)
    e.sync()
    gpu_s = (time.time() - t0) / K
    edges = B * (s2 + s2 * s1)
    return {"config": "configs[0]: toy-PPI-shaped (N=%d, F=%d, C=%d multi-hot, avg degree 28) supervised graphsage_mean --sigmoid, "
                      "B=%d, fan-out %dx%d, dims %d/%d, %d epochs = %d steps (example_supervised.sh:1 semantics; synthetic "
                      "data: the reference's toy-ppi blobs are stripped)" % (N, F, C, B, s1, s2, D, D, EPOCHS, steps),
            "cpu_port": {"s_per_step": cpu_s, "value": edges / cpu_s, "unit": "sampled-edges/s", "cores": int(port.threads),
                         "kind": "port", "val_micro_f1": float(cpu_f1), "wall_s": cpu_wall},
            "mi355x": {"ms_per_step": gpu_s * 1e3, "value": edges / gpu_s, "unit": "sampled-edges/s",
                       "val_micro_f1": float(gpu_f1), "sampler_law": args.sampler_law},
            "val_nodes": int(len(val)), "train_steps": int(steps)}


def run_aux(DG, args, B, s1, s2):
    """Short timed runs of the other BASELINE configurations so that the driver's record carries them:
    configs[2] maxpool, configs[3] unsupervised (one GPU's share), configs[4] RMAT (one GPU's share)."""
    from graphsage_amd.utils import random_walk_pairs_device
    out = {}
    K, spl = args.aux_steps, args.steps_per_launch
    warm = 2 * min(spl, 8) + 6

    dominant = aux_dominant_profile()

    def timed(model, e, n_roots, fan1, feat_dim, key, flops_fwd=None):
        from graphsage_amd import ops
        model.train_steps_device(B, warm, steps_per_launch=spl)
        for _ in range(2):                  # every graph length of the timed call: eager, then captured
            model.train_steps_device(B, K, steps_per_launch=spl)
        for _ in range(min(16, max(2, -(-100 // max(1, K))))):     # device warm right before the timed call (see main())
            model.train_steps_device(B, K, steps_per_launch=spl)
            e.sync()
        e.sync()
        t0 = time.time()
        model.train_steps_device(B, K, steps_per_launch=spl)
        e.sync()
        dt = time.time() - t0
        evs = [(ops.Event(), ops.Event()) for _ in range(6)]
        for a, b in evs:
            a.record(e.stream)
            model.train_steps_device(B, K, steps_per_launch=spl)
            b.record(e.stream)
        e.sync()
        ev = np.asarray([a.elapsed_ms(b) for a, b in evs][1:]) / K
        r = {"ms_per_step": dt / K * 1e3, "ms_per_step_events_median": float(np.median(ev)),
             "value": n_roots * (s2 + s2 * fan1) * K / dt, "unit": "sampled-edges/s", "steps": K}
        # the configuration's own roofline: algorithmic bytes of a step = rows * F * 4 + ids + mean writes
        rows = n_roots * (1 + s2 + s2 * fan1)
        alg = rows * feat_dim * 4 + n_roots * (s2 + s2 * fan1) * 4 + n_roots * (1 + s2) * feat_dim * 4
        r["roofline"] = {"bound": "hbm", "algorithmic_bytes_per_step": alg, "achieved": alg / (dt / K) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac_algorithmic": alg / (dt / K) / 1e9 / HBM_PEAK_GBS}
        if flops_fwd is not None:
            # pooling aggregators are bound by the MLP contraction.  The figure that says how well the DEVICE's dominant kernel
            # uses its pipe: the flops that kernel issues (every 128 x 256 tile of the step's distinct rows x 32-k stages x the
            # arithmetic's piece products) / its average duration in the committed kernel trace / the dense fp16|bf16 MFMA peak --
            # and, beside it, the same against the peak at the clock the board's power management leaves a loop of that
            # kernel (profiles/r05_pool_clock_probe.txt).  (Rounds 4-5 reported the REFERENCE graph's forward flops -- every
            # gathered row -- over the fp32-MFMA peak here: 0.84, a number about a kernel the device does not run.)
            a0 = model.aggregators[0]
            products = 3 if e.pool_f16 else 6
            derated_ghz = 1.92 if e.pool_f16 else 2.02
            r["roofline"].update({"bound": "mfma", "peak_tflops": MFMA_16BIT_PEAK_TF, "reference_graph_flops_fwd_per_step": flops_fwd})
            dk = (dominant or {}).get("configs", {}).get(key)
            lu = getattr(a0, "last_unique", None)
            if dk and lu is not None and "split" in dk["kernel"]:
                rows_u = int(lu[0].item())
                tiles_m, hidden, stages = -(-rows_u // 128), a0.hidden_dim, -(-feat_dim // 32)
                issued = 2.0 * (tiles_m * 128) * (-(-hidden // 256) * 256) * (stages * 32) * products
                tf = issued / (dk["avg_us"] * 1e-6) / 1e12
                r["roofline"].update({
                    "dominant_kernel_flops_issued": issued, "dominant_kernel_distinct_rows": rows_u,
                    "dominant_kernel_piece_products": products, "achieved_tflops": tf,
                    "frac": tf / MFMA_16BIT_PEAK_TF,
                    "frac_at_power_managed_clock": tf / (MFMA_16BIT_PEAK_TF * derated_ghz / 2.4),
                    "fp32_equivalent_tflops": issued / products / (dk["avg_us"] * 1e-6) / 1e12,
                    "frac_basis": "flops ISSUED by %s (distinct rows padded to 128-row tiles x %d columns x %d k x %d piece products) / "
                                  "its avg_us in the committed kernel trace / 2.5 PFLOP/s dense 16-bit MFMA peak; "
                                  "frac_at_power_managed_clock: the same against the peak at the %.2f GHz the board holds in a loop "
                                  "of this kernel (2.4 GHz nominal)" % (dk["kernel"].split(" [")[0], -(-hidden // 256) * 256,
                                                                        stages * 32, products, derated_ghz)})
        if dominant and key in dominant.get("configs", {}):
            r["roofline"]["dominant_kernel"] = dominant["configs"][key]
            r["roofline"]["dominant_kernel_source"] = "PROFILE-SOURCED: %s" % dominant["path"]
            r["roofline"]["dominant_kernel_stale"] = dominant.get("lib_digest") != lib_digest()
        return r

    try:
        from graphsage_amd import inits
        inits.set_seed(123)                # both max-pool legs below start from the SAME initial weights
        e, model, ph, _ = build_model(DG, args, 1, 0, "graphsage_maxpool")
        model.attach_device_epoch(np.random.RandomState(123).permutation(DG.train_nodes), DG.label_table)
        F_, H_, D1, D2, C_ = args.feat_dim, 512, args.dim_1, args.dim_2, DG.num_classes
        n0, n1 = B, B * s2
        flops = (2.0 * (n0 * s2 + n1 * s1) * F_ * H_ + 2.0 * (n0 + n1) * (F_ + H_) * D1          # layer 0: MLP + both matmuls
                 + 2.0 * n0 * s2 * (2 * D1) * H_ + 2.0 * n0 * (2 * D1 + H_) * D2 + 2.0 * n0 * 2 * D2 * C_)
        r = timed(model, e, B, s1, args.feat_dim, "graphsage_maxpool", flops_fwd=flops)
        r["config"] = "configs[2]: Reddit-shaped supervised graphsage_maxpool, fan-out %dx%d, batch %d" % (s1, s2, B)
        r["loss_after"] = model._fetch(B)[0]
        r["pooling_mlp_arithmetic"] = ("fp32 operands as two fp16 pieces each under row / column scales, three exact products per "
                                       "element pair on the fp16 matrix pipe, fp32 accumulation (csrc/gs_split16.hip)"
                                       if e.pool_f16 else "fp32 operands as three bf16 pieces each, six products (csrc/gs_split.hip)")
        out["graphsage_maxpool"] = r
        if e.pool_f16:
            # the same step with the pooling MLP in the three-piece bf16 arithmetic of round 4 (no operand bit dropped, six products;
            # power-bound): reported beside the default so that the arithmetic change is visible in the driver's own record
            del model
            os.environ["GS_POOL_F16"] = "0"
            try:
                inits.set_seed(123)
                e, model, ph, _ = build_model(DG, args, 1, 0, "graphsage_maxpool")
                model.attach_device_epoch(np.random.RandomState(123).permutation(DG.train_nodes), DG.label_table)
                r3 = timed(model, e, B, s1, args.feat_dim, "graphsage_maxpool_bf16x3", flops_fwd=flops)
                r["three_bf16_pieces_roofline"] = r3["roofline"]
                r["three_bf16_pieces"] = {"ms_per_step": r3["ms_per_step"], "ms_per_step_events_median": r3["ms_per_step_events_median"],
                                          "loss_after": model._fetch(B)[0],
                                          "note": "same seed, data and step count as the default leg.  The two arithmetics (and the plain "
                                                  "fp32-MFMA kernel) agree per step to 2e-5 in the loss and drift apart along a trajectory "
                                                  "like any two fp32 summation orders do (max-pool arg-max choices are discontinuous): "
                                                  "tests/test_bench_parity_gpu.py::test_maxpool_two_fp16_pieces_train_like_three_bf16_pieces"}
            finally:
                os.environ["GS_POOL_F16"] = "1"
    except Exception as ex:            # an aux failure must not lose the headline line
        out["graphsage_maxpool"] = {"error": repr(ex)}
    try:
        e, model, ph, _ = build_model(DG, args, 1, 0, "gcn")
        model.attach_device_epoch(np.random.RandomState(123).permutation(DG.train_nodes), DG.label_table)
        r = timed(model, e, B, s1, args.feat_dim, "gcn")
        r["config"] = ("configs[1]-shaped: Reddit-shaped supervised gcn (GCNAggregator, dims 2x%d, concat off: "
                       "supervised_train.py:175-185), fan-out %dx%d, batch %d" % (args.dim_1, s1, s2, B))
        r["loss_after"] = model._fetch(B)[0]
        out["gcn"] = r
        del model
    except Exception as ex:
        out["gcn"] = {"error": repr(ex)}
    try:
        e, model, ph, _ = build_model(DG, args, 1, 0, "graphsage_mean", unsupervised=True)
        pairs = random_walk_pairs_device(DG.train_csr[0], DG.train_csr[1], DG.train_nodes, max_pairs=1000000, seed=123)
        model.attach_device_pairs(pairs.cpu().numpy())
        r = timed(model, e, 2 * B + 20, s1, args.feat_dim, "unsupervised")
        r["config"] = ("configs[3]: Reddit-shaped unsupervised graphsage_mean (random-walk pairs, 20 negatives), "
                       "fan-out %dx%d, batch %d, ONE GPU's share of the 8-GPU configuration" % (s1, s2, B))
        r["loss_after"] = model._fetch_unsup(B)[0]
        out["unsupervised"] = r
    except Exception as ex:
        out["unsupervised"] = {"error": repr(ex)}
    try:
        a2 = argparse.Namespace(**vars(args))
        a2.nodes, a2.feat_dim, a2.classes, a2.samples_1, a2.workload = 10000000, 256, 64, 15, "rmat"
        del e, model
        e, model, ph, order, labels, n_edges = build_rmat(a2, 1, 0)
        model.attach_device_epoch(order, labels)
        r = timed(model, e, B, 15, 256, "rmat")
        r["config"] = ("configs[4]: RMAT N=10^7 / E=%d directed, F=256, supervised graphsage_mean, fan-out 15x%d, batch %d, "
                       "ONE GPU's share of the 8-GPU configuration" % (n_edges, s2, B))
        out["rmat"] = r
    except Exception as ex:
        out["rmat"] = {"error": repr(ex)}
    try:
        del e, model
    except Exception:
        pass
    try:
        out["toy_ppi"] = toy_ppi_leg(args)
    except Exception as ex:
        out["toy_ppi"] = {"error": repr(ex)}
    return out


if __name__ == "__main__":
    main()
