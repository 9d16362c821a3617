import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_model_gpu import build
from graphsage_amd import engine as eng
from graphsage_amd.neigh_samplers import CSRAdjacency
dev = torch.device("cuda:0")
use_graphs = os.environ.get("USE_GRAPHS", "1") == "1"
G, it, ph, sampler, model, ns = build(dev, "mean", True, False, csr=True, n_nodes=3000, dim=32)
model.use_graphs = use_graphs
e = eng.get_engine()
model.attach_device_epoch(it.train_nodes, it.label_matrix)
B = 128
for epoch in range(2):
    model.set_epoch_order(np.random.RandomState(epoch).permutation(it.train_nodes))
    for i in range(len(it.train_nodes) // B):
        r = model.train_step_device(B, fetch=True)
        print("epoch", epoch, "step", i, "loss", r[0], flush=True)
print("train ok", flush=True)
model.adj_info.assign(CSRAdjacency(it.test_csr[0], it.test_csr[1], G.n_nodes, e.device))
val = it.val_nodes.astype(np.int32)
print("n val", len(val), flush=True)
loss, preds = model.eval_step({ph['batch']: val, ph['labels']: it.label_matrix[val], ph['batch_size']: len(val)})
print("eval ok", loss, flush=True)
