"""PCIe-inclusive rate of the reference-style boundary: model.train_step(feed_dict) with HOST batch ids + label matrix
per step (what graphsage_amd.supervised_train does, = sess.run(..., feed_dict) of supervised_train.py:275), against the
device-resident epoch that bench.py times.  Reddit-shaped workload, one GPU.

    python -m benchmarks.feed_path
"""
import importlib.util
import json
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from graphsage_amd.minibatch import NodeMinibatchIterator
    from graphsage_amd.utils import reddit_shaped
    args = bench.parse_args([])
    G = reddit_shaped(avg_degree=args.avg_degree, seed=123, n_nodes=args.nodes, feat_dim=args.feat_dim, num_classes=args.classes)
    it = NodeMinibatchIterator(G, None, {}, None, G.num_classes, batch_size=512, max_degree=128, build_padded=False)
    e, model, ph = bench.build_model(G, it, args, 1, 0)
    order = np.random.RandomState(123).permutation(it.train_nodes)
    B, steps = 512, 200
    feeds = []
    for i in range(steps + 20):
        b = order[(i * B) % (len(order) - B):][:B].astype(np.int32)
        feeds.append({ph['batch']: b, ph['labels']: it.label_matrix[b], ph['batch_size']: B, ph['dropout']: 0.0})
    res = {}
    for fetch in (True, False):
        for f in feeds[:20]:
            model.train_step(f, fetch=fetch)
        e.sync()
        torch.cuda.synchronize()
        t0 = time.time()
        for f in feeds[20:]:
            model.train_step(f, fetch=fetch)
        e.sync()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / steps
        res["fetch_loss_preds_every_step" if fetch else "no_fetch"] = {"ms_per_step": dt * 1e3,
                                                                         "sampled_edges_per_s": B * 260 / dt}
    res["host_bytes_per_step"] = B * 4 + B * G.num_classes * 4
    print(json.dumps(res))


if __name__ == "__main__":
    main()
