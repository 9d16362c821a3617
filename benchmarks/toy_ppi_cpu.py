"""BASELINE configs[0]: toy-PPI-shaped supervised graphsage_mean on the CPU restatement of the reference path
(example_supervised.sh semantics: --sigmoid, B=512, fan-out 25x10, dims 128/128, max_degree 128).  CPU only."""
import json
import time

import numpy as np

from graphsage_amd.utils import padded_from_csr, synthetic_graph, build_csr
from oracle.cpu_baseline import CpuSupervisedMean, time_cpu_baseline
from oracle.graphsage_oracle import calc_f1_micro


def main():
    G = synthetic_graph(n_nodes=14755, feat_dim=50, num_classes=121, avg_degree=28, seed=123, multilabel=True)
    train = ~(G.val_mask | G.test_mask)
    keep = train[G.src] & train[G.dst]
    rp, col = build_csr(G.n_nodes, G.src, G.dst, keep=keep)
    adj, deg = padded_from_csr(rp, col, G.n_nodes, 128, np.random.RandomState(123))
    labels = G.label_matrix()
    train_nodes = np.nonzero(train & (deg[:G.n_nodes] > 0))[0]
    res = time_cpu_baseline(G.padded_features(), adj, labels, train_nodes, 121, batch_size=512, num_samples=(25, 10),
                            dims=(50, 128, 128), budget_s=20.0)
    # quality: 10 epochs of the same port, micro-F1 on the validation nodes (sigmoid / multi-label)
    model = CpuSupervisedMean(G.padded_features(), adj, [50, 128, 128], 121, [25, 10], sigmoid_loss=True)
    rng = np.random.RandomState(123)
    t0 = time.time()
    for ep in range(10):
        order = rng.permutation(train_nodes)
        for i in range(0, len(order) - 511, 512):
            b = order[i:i + 512]
            model.train_step(b, labels[b])
    val = np.nonzero(G.val_mask)[0][:2048]
    rp2, col2 = build_csr(G.n_nodes, G.src, G.dst)
    import torch
    model.adj = torch.as_tensor(padded_from_csr(rp2, col2, G.n_nodes, 128, np.random.RandomState(1))[0], dtype=torch.int64)
    samples, sizes = model.sample(val)
    with torch.no_grad():
        _, logits = model.forward(samples, sizes, labels[val])
    f1 = calc_f1_micro(labels[val], torch.sigmoid(logits).numpy(), True)
    res["val_f1_micro_10_epochs"] = float(f1)
    res["train_s"] = time.time() - t0
    print(json.dumps(res))


if __name__ == "__main__":
    main()
