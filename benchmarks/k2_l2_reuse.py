"""Round 5: can an XCD-affine order of K2's work turn the duplicate rows of a step (37 % of the hop-2 gather at Reddit's degree)
into L2 hits?  Takes the ids a real step of the headline configuration samples and simulates, per XCD, an LRU cache of the L2's
size over the row stream in K2's issue order (work item = (parent row, 64-float4 chunk), workgroup b = 4 consecutive items on
XCD b % 8), then the same for orders that give a parent's items to the XCD of ... (a) its index block, (b) its own id hash, and
for the bound no parent order can beat: every duplicate of a row on one XCD with an infinite cache.
    python benchmarks/k2_l2_reuse.py [out.json]"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def lru_hits(stream, capacity):
    cache, hits = OrderedDict(), 0
    for r in stream:
        if r in cache:
            cache.move_to_end(r)
            hits += 1
        else:
            cache[r] = True
            if len(cache) > capacity:
                cache.popitem(last=False)
    return hits


def main():
    args = bench.parse_args(["--no-cpu-baseline", "--no-aux"])
    from graphsage_amd.utils import reddit_shaped_device
    dev = torch.device("cuda:0")
    DG = reddit_shaped_device(dev, n_nodes=args.nodes, feat_dim=args.feat_dim, num_classes=args.classes, avg_degree=args.avg_degree,
                              seed=123, feat_signal=args.feat_signal)
    e, model, ph, adj_info = bench.build_model(DG, args, 1, 0, "graphsage_mean")
    model.attach_device_epoch(np.random.RandomState(123).permutation(DG.train_nodes), DG.label_table)
    model.train_steps_device(512, 6, steps_per_launch=2)
    e.sync()
    ids2 = model.samples1[2].cpu().numpy().reshape(5120, 25)          # hop-2 ids of the last step: parent row -> its 25 neighbors
    row_bytes = 608 * 4
    cap = (4 << 20) // row_bytes                                       # rows an XCD's 4 MB L2 holds if it held nothing else
    n_rows, s = ids2.shape
    total = n_rows * s
    uniq = len(np.unique(ids2))
    res = {"parents": n_rows, "fan_out": s, "row_fetches": total, "distinct_rows": int(uniq), "duplicate_frac": 1.0 - uniq / total,
           "l2_rows_per_xcd": cap, "orders": {}}

    def simulate(xcd_of_parent):
        hits = 0
        for x in range(8):
            rows = np.flatnonzero(xcd_of_parent == x)
            hits += lru_hits(ids2[rows].reshape(-1).tolist(), cap)     # a parent's three chunk items fetch disjoint column ranges:
        return hits / total                                            # one row stream per XCD stands for each of them

    # K2 today: item w = 3 * row + chunk, workgroup = 4 consecutive items -> parent rows are dealt round-robin-ish over the XCDs
    today = (np.arange(n_rows) * 3 // 4) % 8
    res["orders"]["today (4 consecutive items per workgroup, workgroup b on XCD b % 8)"] = simulate(today)
    res["orders"]["contiguous blocks of 640 parents per XCD"] = simulate(np.arange(n_rows) * 8 // n_rows)
    par = model.samples1[1].cpu().numpy()                               # the parents' own ids
    res["orders"]["parent id % 8 (the same node sampled twice meets itself)"] = simulate(par % 8)
    # the bound: rows assigned to XCDs so that ALL duplicates of a row meet, infinite cache -> every duplicate is a hit
    res["orders"]["bound: every duplicate of a row on one XCD, infinite L2"] = 1.0 - uniq / total
    # ... with the real L2: neighbors dealt by neighbor id % 8 (needs a cross-XCD sum of 8 partial means per parent), stream order kept
    hits = 0
    flat = ids2.reshape(-1)
    for x in range(8):
        hits += lru_hits(flat[flat % 8 == x].tolist(), cap)
    res["orders"]["neighbor id % 8 per XCD (8 partial means per parent), 4 MB LRU"] = hits / total
    # distance between the two fetches of a duplicated row, in row fetches of the whole launch
    first, dist = {}, []
    for i, r in enumerate(flat.tolist()):
        if r in first:
            dist.append(i - first[r])
        first[r] = i
    dist = np.asarray(dist)
    res["duplicate_distance_fetches"] = {"median": float(np.median(dist)), "p10": float(np.percentile(dist, 10)),
                                         "within_one_xcd_l2_of_the_stream": float((dist < 8 * cap).mean())}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
