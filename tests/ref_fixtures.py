"""Loader for tests/golden/ref_*.npz -- outputs of the REFERENCE'S OWN code (tests/golden/make_ref_fixtures.py runs
/root/reference/graphsage/*.py unmodified on the torch-backed TF1 stand-in of tests/tf1_shim).  Shared by the CPU suite
(oracle == reference) and the `-m gpu` suite (HIP == reference)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SUP = ["sup_mean", "sup_mean_add_sigmoid", "sup_gcn", "sup_maxpool", "sup_meanpool_sigmoid", "sup_mean_3layer",
       "sup_mean_full_degree", "sup_mean_identity", "sup_mean_tail",
       "sup_maxpool_big",             # model_size = "big": hidden 1024 (aggregators.py:139-142)
       "sup_gcn_tail"]                # the GCN model at widths the device's fused tail takes (128 per layer)
SUP_DROPOUT = ["sup_mean_dropout", "sup_maxpool_dropout"]
UNSUP = ["unsup_mean", "unsup_gcn", "unsup_maxpool", "unsup_meanpool",
         "unsup_mean_tail"]            # widths the device's fused layer-1 + link-prediction launches take (128 per layer)
SUP_CPU = []
UNSUP_CPU = []


class Fixture(object):
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, "ref_%s.npz" % name))
        self.cfg = json.loads(str(self.z["cfg"])) if "cfg" in self.z.files else {}
        c = self.cfg
        if c:
            self.K = len(c["num_samples"])
            self.agg = c["aggregator_type"]
            self.out_dim = c["dim"] * (2 if self.agg == "gcn" else 1)     # supervised_train.py:175-176
            self.identity_dim = c.get("identity_dim", 0)
            self.dims = [self.z["graph/feats"].shape[1] + self.identity_dim] + [self.out_dim] * self.K
            self.n_steps = int(self.z["n_steps"])
            self.n_nodes = self.z["graph/feats"].shape[0] - 1

    def __getitem__(self, k):
        return self.z[k]

    def has(self, k):
        return k in self.z.files

    def lists(self, which):
        rp, col = self.z["graph/%s_rowptr" % which], self.z["graph/%s_col" % which]
        return [col[rp[i]:rp[i + 1]] for i in range(len(rp) - 1)]

    def params(self, prefix, dtype, supervised=True):
        """{"agg": [per-layer dicts], "node_pred": {...}, ("embeds": ...)} in the oracle's layout from `prefix + name`."""
        agg = []
        for i in range(self.K):
            p = {}
            for k in ("neigh_weights", "self_weights", "weights", "mlp_weights", "mlp_bias"):
                key = "%sagg%d/%s" % (prefix, i, k)
                if key in self.z.files:
                    p[k] = self.z[key].astype(dtype)
            agg.append(p)
        out = {"agg": agg}
        if supervised:
            out["node_pred"] = {"weights": self.z[prefix + "node_pred/weights"].astype(dtype),
                                "bias": self.z[prefix + "node_pred/bias"].astype(dtype)}
        if prefix + "embeds" in self.z.files:
            out["embeds"] = self.z[prefix + "embeds"].astype(dtype)
        return out

    def perms(self, step_prefix, n_calls):
        return [self.z["%sperm%d" % (step_prefix, k)] for k in range(n_calls)]

    def sampled(self, step_prefix, k0, K):
        """Flat id vectors of sampler calls k0 .. k0+K-1 (models.py:273)."""
        return [self.z["%ssampled%d" % (step_prefix, k0 + k)].reshape(-1) for k in range(K)]


def flat_items(params):
    """(name, array) pairs in a fixed order over the layout of Fixture.params."""
    items = []
    for i, p in enumerate(params["agg"]):
        for k in sorted(p):
            items.append(("agg%d/%s" % (i, k), p[k]))
    if "node_pred" in params:
        items.append(("node_pred/weights", params["node_pred"]["weights"]))
        items.append(("node_pred/bias", params["node_pred"]["bias"]))
    if "embeds" in params:
        items.append(("embeds", params["embeds"]))
    return items
