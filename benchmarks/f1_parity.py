"""micro-F1 half of the metric: train the MI355X engine (native CSR sampler) and the torch-CPU port of the reference
graph (padded-table sampler) for the same number of steps on the same Reddit-shaped synthetic graph and report
validation micro-F1 of both (statistical parity: different sampler joint law and different init streams).

    python benchmarks/f1_parity.py [--steps 150] [--nodes 232965]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--nodes", type=int, default=232965)
    ap.add_argument("--val_nodes", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=None)
    ap.add_argument("--signal", type=float, default=0.04, help="class-centroid scale in the features (0.5 = trivially separable)")
    args = ap.parse_args()
    from graphsage_amd import engine as eng
    from graphsage_amd.minibatch import NodeMinibatchIterator
    from graphsage_amd.models import Placeholder, SAGEInfo
    from graphsage_amd.neigh_samplers import AdjInfo, CSRAdjacency, UniformNeighborSampler
    from graphsage_amd.supervised_models import SupervisedGraphsage
    from graphsage_amd.utils import padded_from_csr, reddit_shaped
    from oracle import graphsage_oracle as orc
    from oracle.cpu_baseline import CpuSupervisedMean

    G = reddit_shaped(avg_degree=50, seed=123, n_nodes=args.nodes, feat_signal=args.signal)
    it = NodeMinibatchIterator(G, None, {}, None, G.num_classes, batch_size=512, max_degree=128, build_padded=False)
    B, s1, s2 = 512, 25, 10
    val = it.val_nodes[: args.val_nodes].astype(np.int32)
    order = np.random.RandomState(123).permutation(it.train_nodes)

    # ---------------- MI355X engine
    eng.reset_engine()
    e = eng.get_engine()
    ph = {'labels': Placeholder('labels'), 'batch': Placeholder('batch1'), 'dropout': Placeholder('dropout', 0.),
          'batch_size': Placeholder('batch_size')}
    train_adj = CSRAdjacency(it.train_csr[0], it.train_csr[1], G.n_nodes, e.device)
    test_adj = CSRAdjacency(it.test_csr[0], it.test_csr[1], G.n_nodes, e.device)
    adj_info = AdjInfo(train_adj)
    sampler = UniformNeighborSampler(adj_info)
    layer_infos = [SAGEInfo("node", sampler, s1, 128), SAGEInfo("node", sampler, s2, 128)]
    model = SupervisedGraphsage(G.num_classes, ph, G.padded_features(), adj_info, it.deg, layer_infos, learning_rate=0.01)
    model.attach_device_epoch(order, it.label_matrix)
    t0 = time.time()
    model.train_steps_device(B, args.steps)
    e.sync()
    t_gpu = time.time() - t0
    adj_info.assign(test_adj)
    preds = []
    for a in range(0, len(val), B):
        b = val[a:a + B]
        _, p = model.eval_step({ph['batch']: b, ph['labels']: it.label_matrix[b], ph['batch_size']: len(b)})
        preds.append(p)
    f1_gpu = orc.calc_f1_micro(it.label_matrix[val], np.vstack(preds), False)

    # ---------------- torch-CPU port of the reference graph (padded sampler, one shared column permutation per call)
    rng = np.random.RandomState(123)
    adj, _ = padded_from_csr(it.train_csr[0], it.train_csr[1], G.n_nodes, 128, rng)
    test_adj_p, _ = padded_from_csr(it.test_csr[0], it.test_csr[1], G.n_nodes, 128, rng)
    cpu = CpuSupervisedMean(G.padded_features(), adj, [G.feats.shape[1], 128, 128], G.num_classes, [s1, s2], lr=0.01)
    cpu_steps = args.cpu_steps or args.steps
    t0 = time.time()
    for i in range(cpu_steps):
        b = order[i * B:(i + 1) * B]
        cpu.train_step(b, it.label_matrix[b])
    t_cpu = time.time() - t0
    cpu.adj = torch.from_numpy(np.ascontiguousarray(test_adj_p, dtype=np.int64))
    preds = []
    with torch.no_grad():
        for a in range(0, len(val), B):
            b = val[a:a + B]
            samples, sizes = cpu.sample(b)
            _, logits = cpu.forward(samples, sizes, it.label_matrix[b])
            preds.append(torch.softmax(logits, dim=1).numpy())
    f1_cpu = orc.calc_f1_micro(it.label_matrix[val], np.vstack(preds), False)
    res = {"feat_signal": args.signal, "steps": args.steps, "cpu_steps": cpu_steps, "val_nodes": int(len(val)), "micro_f1_mi355x": f1_gpu,
           "micro_f1_cpu_port": f1_cpu, "train_wall_s_mi355x": t_gpu, "train_wall_s_cpu_port": t_cpu,
           "cpu_threads": torch.get_num_threads()}
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "f1_parity.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
