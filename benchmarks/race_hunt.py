"""Diagnostics: hunt a rare run-to-run difference INSIDE one process.  Two identical models (same seeds, same graph, own engines
and streams) train the bench's headline step in lockstep, CHUNK steps per call; after every chunk every step-state buffer of the
two is compared bit for bit.  The first chunk that differs names the stage that broke (ids / labels -> sampler, means -> gather
riders, h0 -> layer-0 forward, z / dz / d_h0 -> fused tail, grads -> weight gradients, params -> optimizer).
    python benchmarks/race_hunt.py [chunks] [chunk_steps] [bench args...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from graphsage_amd import inits  # noqa: E402
from graphsage_amd.utils import reddit_shaped_device  # noqa: E402


def state(model, e, B):
    out = {"params": e.params, "grads": e.grads, "adam_m": e.adam_m}
    for par in (0, 1):
        out["ids%d" % par] = model.ids_buffer(B, parity=par)[0]
        lab = e._ws.get(("m", ("labels", par), B, model.num_classes)) if hasattr(e, "_ws") else None
        if lab is not None:
            out["labels%d" % par] = lab.buf if hasattr(lab, "buf") else lab
    for k, v in e._ws.items():
        name = str(k)
        if isinstance(v, torch.Tensor):
            out["ws:" + name] = v
        elif hasattr(v, "buf") and isinstance(v.buf, torch.Tensor):
            out["ws:" + name] = v.buf
    return out


SERIAL = os.environ.get("RACE_SERIAL", "1") == "1"


def main():
    chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    cs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    args = bench.parse_args(sys.argv[3:] + ["--no-cpu-baseline", "--no-aux"])
    dev = torch.device("cuda:0")
    B = args.batch_size
    DG = reddit_shaped_device(dev, n_nodes=args.nodes, feat_dim=args.feat_dim, num_classes=args.classes, avg_degree=args.avg_degree,
                              seed=123, feat_signal=args.feat_signal)
    pair = []
    for _ in range(2):
        inits.set_seed(5)
        e, model, ph, adj = bench.build_model(DG, args, 1, 0, args.model, False)
        epoch = np.random.RandomState(123).permutation(DG.train_nodes)
        model.attach_device_epoch(epoch, DG.label_table)
        pair.append((e, model))
    (eA, mA), (eB, mB) = pair
    for e, m in pair:                               # eager + capture of the chunk graph
        m.train_steps_device(B, 3 * cs, steps_per_launch=cs)
        e.sync()
    torch.cuda.synchronize()
    bad = 0
    for c in range(chunks):
        mA.train_steps_device(B, cs, steps_per_launch=cs)
        if SERIAL:
            eA.sync()                                    # (one model on the chip at a time, as in bench.py)
        mB.train_steps_device(B, cs, steps_per_launch=cs)
        eA.sync(); eB.sync()
        torch.cuda.synchronize()
        if not torch.equal(eA.params, eB.params) or c % 50 == 49:
            sa, sb = state(mA, eA, B), state(mB, eB, B)
            diff = [k for k in sa if k in sb and sa[k].shape == sb[k].shape and not torch.equal(sa[k], sb[k])]
            if diff:
                bad += 1
                print("chunk %d (steps %d..%d): DIFFER: %s" % (c, c * cs, (c + 1) * cs - 1, ", ".join(sorted(diff)[:40])), flush=True)
                for k in sorted(diff):
                    a, b = sa[k], sb[k]
                    d = (a != b)
                    idx = d.flatten().nonzero().flatten()
                    if a.dtype.is_floating_point:
                        mx = float((a.double() - b.double()).abs().max())
                    else:
                        mx = float((a.long() - b.long()).abs().max())
                    ld = a.shape[-1] if a.dim() > 1 else 1
                    rows = sorted(set((idx[:4000] // ld).tolist()))
                    print("    %-66s %7d of %7d differ, max |a-b| %.3g, rows %s%s" % (k[:66], int(d.sum()), d.numel(), mx, rows[:12],
                                                                                 " ..." if len(rows) > 12 else ""), flush=True)
                break
    print("done: %d chunks of %d steps, %d differing" % (c + 1, cs, bad))


if __name__ == "__main__":
    main()
