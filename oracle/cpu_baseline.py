"""CPU baseline: the reference's TF graph restated op-for-op with torch-CPU tensors (MKL, all host
cores), used ONLY by bench.py's `cpu_baseline` leg and by tests (see graphsage_oracle.py header).

TensorFlow 1.x cannot be installed here, so this is a PORT ("kind": "port"), not the reference:
  padded-table sampler with one shared column permutation   neigh_samplers.py:24-29
  fully MATERIALISED per-hop feature gather                 models.py:299
  mean(axis=1) + two matmuls + concat + relu                aggregators.py:43-64
  l2_normalize + dense head + softmax/sigmoid CE            supervised_models.py:85-118
  backward (autograd = tf.gradients), clip +-5, TF Adam     supervised_models.py:95-99
It is cross-checked against the NumPy oracle in tests/test_oracle.py.
"""
import time

import numpy as np
import torch


class CpuSupervisedMean(object):
    def __init__(self, features, adj, dims, num_classes, num_samples, concat=True, sigmoid_loss=False, lr=0.01,
                 weight_decay=0.0, seed=123, threads=None):
        if threads:
            torch.set_num_threads(threads)
        self.threads = torch.get_num_threads()
        g = torch.Generator().manual_seed(seed)
        self.X = torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32))
        self.adj = torch.from_numpy(np.ascontiguousarray(adj, dtype=np.int64))
        self.num_samples = list(num_samples)
        self.concat, self.sigmoid_loss, self.lr, self.wd = concat, sigmoid_loss, lr, weight_decay
        self.dims = list(dims)
        K = len(num_samples)
        self.params = []
        for layer in range(K):
            dm = 2 if (concat and layer != 0) else 1
            din, dout = dm * dims[layer], dims[layer + 1]
            r = float(np.sqrt(6.0 / (din + dout)))
            self.params.append((torch.empty(din, dout).uniform_(-r, r, generator=g).requires_grad_(),   # neigh_weights
                                torch.empty(din, dout).uniform_(-r, r, generator=g).requires_grad_()))  # self_weights
        dm = 2 if concat else 1
        r = float(np.sqrt(6.0 / (dm * dims[-1] + num_classes)))
        self.W = torch.empty(dm * dims[-1], num_classes).uniform_(-r, r, generator=g).requires_grad_()
        self.b = torch.zeros(num_classes, requires_grad=True)
        self.all = [p for pair in self.params for p in pair] + [self.W, self.b]
        self.m = [torch.zeros_like(p) for p in self.all]
        self.v = [torch.zeros_like(p) for p in self.all]
        self.t = 0
        self.rng = np.random.RandomState(seed)

    def set_params_from_oracle(self, params):
        with torch.no_grad():
            for (wn, ws), p in zip(self.params, params["agg"]):
                wn.copy_(torch.from_numpy(p["neigh_weights"]))
                ws.copy_(torch.from_numpy(p["self_weights"]))
            self.W.copy_(torch.from_numpy(params["node_pred"]["weights"]))
            self.b.copy_(torch.from_numpy(params["node_pred"]["bias"]))

    def sample(self, batch, perms=None):
        K = len(self.num_samples)
        samples = [torch.as_tensor(np.asarray(batch), dtype=torch.int64)]
        support, sizes = 1, [1]
        for k in range(K):
            t = K - k - 1
            s = self.num_samples[t]
            support *= s
            rows = self.adj.index_select(0, samples[k])                       # embedding_lookup (:26)
            perm = perms[k] if perms is not None else self.rng.permutation(self.adj.shape[1])
            cols = torch.as_tensor(np.asarray(perm[:s]), dtype=torch.int64)  # shuffled columns, sliced (:27-28)
            samples.append(rows.index_select(1, cols).reshape(-1))
            sizes.append(support)
        return samples, sizes

    def forward(self, samples, sizes, labels):
        K = len(self.num_samples)
        B = samples[0].shape[0]
        hidden = [self.X.index_select(0, s) for s in samples]                 # materialised gathers (models.py:299)
        for layer in range(K):
            wn, ws = self.params[layer]
            dm = 2 if (self.concat and layer != 0) else 1
            nxt = []
            for hop in range(K - layer):
                neigh = hidden[hop + 1].reshape(B * sizes[hop], self.num_samples[K - hop - 1], dm * self.dims[layer])
                means = neigh.mean(dim=1)
                fn, fs = means @ wn, hidden[hop] @ ws
                out = torch.cat([fs, fn], dim=1) if self.concat else fs + fn
                nxt.append(torch.relu(out) if layer != K - 1 else out)
            hidden = nxt
        out = hidden[0]
        out = out * torch.rsqrt(torch.clamp((out * out).sum(dim=1, keepdim=True), min=1e-12))
        logits = out @ self.W + self.b
        labels = torch.as_tensor(labels)
        if self.sigmoid_loss:
            loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, labels)
        else:
            loss = -(labels * torch.log_softmax(logits, dim=1)).sum(dim=1).mean()
        if self.wd:
            for p in self.all:
                loss = loss + self.wd * (p * p).sum() / 2
        return loss, logits

    def train_step(self, batch, labels, perms=None):
        samples, sizes = self.sample(batch, perms)
        loss, logits = self.forward(samples, sizes, labels)
        grads = torch.autograd.grad(loss, self.all)
        self.t += 1
        lr_t = self.lr * np.sqrt(1 - 0.999 ** self.t) / (1 - 0.9 ** self.t)
        with torch.no_grad():
            for p, g, m, v in zip(self.all, grads, self.m, self.v):
                g = g.clamp(-5.0, 5.0)
                m.mul_(0.9).add_(g, alpha=0.1)
                v.mul_(0.999).addcmul_(g, g, value=0.001)
                p.sub_(lr_t * m / (v.sqrt() + 1e-8))
        return float(loss.detach()), logits.detach().numpy(), grads


def time_cpu_baseline(features, adj, label_matrix, train_nodes, num_classes, batch_size=512, num_samples=(25, 10),
                      dims=(602, 128, 128), budget_s=15.0, warmup=2, max_steps=100, order=None, fixed_steps=None,
                      return_model=False, seed=123, perms=None):
    """Times full training steps of the port on a bounded sample of the workload (~budget_s of CPU time).
    `order` (epoch order of root nodes) and `fixed_steps` make the run reproducible step for step (bench.py's micro-F1
    leg trains the MI355X engine on the same order for the same number of steps)."""
    model = CpuSupervisedMean(features, adj, list(dims), num_classes, list(num_samples), seed=seed)
    if order is None:
        order = np.random.RandomState(123).permutation(train_nodes)
    times, i = [], 0
    t_start = time.time()
    while True:
        if fixed_steps is not None:
            b = order[i * batch_size:(i + 1) * batch_size]
        else:
            b = order[(i * batch_size) % max(1, len(order) - batch_size):][:batch_size]
        t0 = time.time()
        model.train_step(b, label_matrix[b], perms[i] if perms is not None else None)   # perms: injected column draws
        dt = time.time() - t0
        if i >= warmup:
            times.append(dt)
        i += 1
        if fixed_steps is not None:
            if i >= fixed_steps:
                break
        elif (time.time() - t_start > budget_s and len(times) >= 3) or len(times) >= max_steps:
            break
    med = float(np.median(times))
    edges = batch_size * (num_samples[1] + num_samples[1] * num_samples[0])
    res = {"value": edges / med, "unit": "sampled-edges/s", "cores": int(model.threads), "kind": "port",
           "sample": "%d full training steps (B=%d, fan-out %dx%d, F=%d) of the torch-CPU restatement of the "
                     "reference TF graph (TF 1.x not installable), median step %.1f ms" %
                     (len(times), batch_size, num_samples[0], num_samples[1], dims[0], med * 1e3),
           "s_per_step": med, "steps_trained": i}
    return (res, model) if return_model else res


def port_micro_f1(model, test_adj, label_matrix, val_nodes, batch_size=512, sigmoid=False, perms=None):
    """Validation micro-F1 of the port (supervised_train.py:63-70, 73-79): forward on the TEST adjacency
    (supervised_train.py:280), argmax / 0.5-threshold, micro average.  perms[i]: injected column draws of batch i."""
    from . import graphsage_oracle as orc
    model.adj = torch.from_numpy(np.ascontiguousarray(test_adj, dtype=np.int64))
    preds = []
    with torch.no_grad():
        for a in range(0, len(val_nodes), batch_size):
            b = val_nodes[a:a + batch_size]
            samples, sizes = model.sample(b, perms[a // batch_size] if perms is not None else None)
            _, logits = model.forward(samples, sizes, label_matrix[b])
            preds.append((torch.sigmoid(logits) if sigmoid else torch.softmax(logits, dim=1)).numpy())
    return orc.calc_f1_micro(label_matrix[val_nodes], np.vstack(preds), sigmoid)
