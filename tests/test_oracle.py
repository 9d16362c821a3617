"""CPU tests: the oracle against the hand-computed golden fixture, finite differences, the CPU-baseline
port and the reference's documented semantics."""
import os

import numpy as np
import pytest

from oracle import graphsage_oracle as orc
from oracle import sampler_hash

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_mean.npz")


def test_golden_tiny_mean_exact():
    g = np.load(GOLD)
    ns = list(g["num_samples"])
    samples, support = orc.sample(g["adj"], g["batch"], ns, [g["perm0"], g["perm1"]])
    assert support == [1, 2, 4]
    assert np.array_equal(samples[1], g["samples1"]) and np.array_equal(samples[2], g["samples2"])
    params = [{"self_weights": g["W0_self"], "neigh_weights": g["W0_neigh"]},
              {"self_weights": g["W1_self"], "neigh_weights": g["W1_neigh"]}]
    out, tape = orc.aggregate_fwd(samples, g["feats"], list(g["dims"]), ns, support, len(g["batch"]), params, "mean", True)
    assert np.array_equal(out, g["out"])                         # integer-valued: exact in fp32
    assert np.array_equal(tape[0][0][3], g["l0_hop0"]) and np.array_equal(tape[0][1][3], g["l0_hop1"])


def test_sampler_semantics():
    rng = np.random.default_rng(0)
    N, md = 50, 8
    neigh = [list(rng.choice(N, size=rng.integers(0, 14), replace=False)) for _ in range(N)]
    skip = rng.random(N) < 0.2
    adj, deg = orc.construct_adj(neigh, md, rng, skip_mask=skip)
    assert adj.shape == (N + 1, md) and (adj[N] == N).all()
    for i in range(N):
        if skip[i] or len(neigh[i]) == 0:
            assert (adj[i] == N).all()                           # val/test rows and isolated rows stay all-pad
        else:
            assert set(adj[i]).issubset(set(neigh[i]))
            if len(neigh[i]) >= md:
                assert len(set(adj[i])) == md                    # down-sampled WITHOUT replacement
    ids = np.array([0, 1, N, 3])
    perm = rng.permutation(md)
    out = orc.uniform_neighbor_sampler(adj, ids, 5, perm)
    assert out.shape == (4, 5) and (out[2] == N).all()
    assert np.array_equal(out, adj[ids][:, perm[:5]])            # ONE column permutation shared by all rows


@pytest.mark.parametrize("agg,concat", [("mean", True), ("mean", False), ("gcn", False), ("maxpool", True), ("meanpool", True)])
@pytest.mark.parametrize("sig", [False, True])
def test_backward_finite_differences(agg, concat, sig):
    rng = np.random.default_rng(1)
    N, F, C, B = 40, 6, 5, 4
    feat = np.vstack([rng.normal(size=(N, F)), np.zeros((1, F))])
    neigh = [list(rng.choice(N, size=rng.integers(0, 8), replace=False)) for _ in range(N)]
    adj, _ = orc.construct_adj(neigh, 8, rng)
    ns, dims = [3, 2], [F, 4, 4]
    params = orc.make_supervised_params(agg, dims, C, concat, rng, dtype=np.float64)
    if agg in ("maxpool", "meanpool"):
        for p in params["agg"]:
            p["mlp_weights"] = p["mlp_weights"][:, :7].copy()
            p["mlp_bias"] = rng.normal(size=7) * 0.1
            p["neigh_weights"] = orc.glorot((7, p["neigh_weights"].shape[1]), rng, np.float64)
    params["node_pred"]["bias"] = rng.normal(size=C) * 0.1
    perms = [rng.permutation(8), rng.permutation(8)]
    batch = rng.choice(N, B, replace=False)
    samples, ss = orc.sample(adj, batch, ns, perms)
    labels = (rng.random((B, C)) > 0.5).astype(np.float64) if sig else np.eye(C)[rng.integers(0, C, B)]
    f = lambda: orc.supervised_fwd_bwd(params, feat, samples, ss, labels, dims, ns, B, agg, concat, sig, weight_decay=0.01)
    r = f()
    for (name, p), (_, g) in zip(orc.flat_param_items(params, agg), orc.flat_param_items(r["grads"], agg)):
        for _ in range(5):
            idx = tuple(rng.integers(0, s) for s in p.shape)
            old = p[idx]
            p[idx] = old + 1e-6; lp = f()["loss"]
            p[idx] = old - 1e-6; lm = f()["loss"]
            p[idx] = old
            fd = (lp - lm) / 2e-6
            assert abs(fd - g[idx]) <= 1e-6 + 1e-5 * (abs(fd) + abs(g[idx])), (name, idx, fd, g[idx])


@pytest.mark.parametrize("agg,concat", [("mean", True), ("gcn", False), ("maxpool", True)])
def test_identity_and_dropout_finite_differences(agg, concat):
    """The oracle's embedding gradient (identity_dim > 0) and its injected-dropout backward, checked by central
    differences in fp64 (the device parity tests for these options rest on this)."""
    rng = np.random.default_rng(5)
    N, F, idim, C, B = 30, 4, 3, 4, 5
    fixed = np.vstack([rng.normal(size=(N, F)), np.zeros((1, F))])
    emb = rng.normal(size=(N + 1, idim)) * 0.5
    neigh = [list(rng.choice(N, size=rng.integers(1, 8), replace=False)) for _ in range(N)]
    adj, _ = orc.construct_adj(neigh, 8, rng)
    ns, dims = [3, 2], [F + idim, 4, 4]
    params = orc.make_supervised_params(agg, dims, C, concat, rng, dtype=np.float64)
    if agg == "maxpool":
        for p in params["agg"]:
            p["mlp_weights"] = p["mlp_weights"][:, :6].copy()
            p["mlp_bias"] = rng.normal(size=6) * 0.1
            p["neigh_weights"] = orc.glorot((6, p["neigh_weights"].shape[1]), rng, np.float64)
    perms = [rng.permutation(8), rng.permutation(8)]
    batch = rng.choice(N, B, replace=False)
    samples, ss = orc.sample(adj, batch, ns, perms)
    labels = np.eye(C)[rng.integers(0, C, B)]
    keep = 0.7
    cache = {}

    def masks(layer, hop, role, n_rows, d):      # fixed random masks, scaled by 1/keep like tf.nn.dropout
        if role == "self" and agg == "maxpool":
            return None
        key = (layer, hop, role)
        if key not in cache:
            cache[key] = (rng.random((n_rows, d)) < keep) / keep
        return cache[key]

    out_dim = dims[-1] * (2 if concat else 1)
    head_mask = (rng.random((B, out_dim)) < keep) / keep

    def f():
        feats = np.concatenate([emb, fixed], axis=1)
        return orc.supervised_fwd_bwd(params, feats, samples, ss, labels, dims, ns, B, agg, concat, False,
                                      weight_decay=0.0, identity_dim=idim, masks=masks, head_mask=head_mask)
    r = f()
    g_emb = r["grads"]["embeds"]
    touched = np.unique(np.concatenate(samples))
    assert np.count_nonzero(g_emb) > 0 and not g_emb[np.setdiff1d(np.arange(N + 1), touched)].any()
    for _ in range(12):
        idx = (int(rng.choice(touched)), int(rng.integers(0, idim)))
        old = emb[idx]
        emb[idx] = old + 1e-6; lp = f()["loss"]
        emb[idx] = old - 1e-6; lm = f()["loss"]
        emb[idx] = old
        fd = (lp - lm) / 2e-6
        assert abs(fd - g_emb[idx]) <= 1e-6 + 1e-5 * (abs(fd) + abs(g_emb[idx])), (idx, fd, g_emb[idx])
    for (name, p), (_, g) in zip(orc.flat_param_items(params, agg), orc.flat_param_items(r["grads"], agg)):
        for _ in range(3):
            idx = tuple(rng.integers(0, s) for s in p.shape)
            old = p[idx]
            p[idx] = old + 1e-6; lp = f()["loss"]
            p[idx] = old - 1e-6; lm = f()["loss"]
            p[idx] = old
            fd = (lp - lm) / 2e-6
            assert abs(fd - g[idx]) <= 1e-6 + 1e-5 * (abs(fd) + abs(g[idx])), (name, idx, fd, g[idx])


def test_reference_schedule_semantics():
    """Appendix A: support sizes, concat order, identity act on the last layer, pad row in the mean."""
    rng = np.random.default_rng(2)
    N, F = 30, 5
    feat = np.vstack([rng.normal(size=(N, F)), np.zeros((1, F))]).astype(np.float32)
    adj = rng.integers(0, N + 1, size=(N + 1, 6)).astype(np.int32)
    samples, ss = orc.sample(adj, np.arange(4), [3, 2], [np.arange(6), np.arange(6)])
    assert ss == [1, 2, 6] and [len(s) for s in samples] == [4, 8, 24]   # samples_2 is the hop next to the batch
    params = orc.make_aggregator_params("mean", [F, 4, 4], True, rng)
    assert params[1]["self_weights"].shape == (8, 4)                     # dim_mult = 2 for layer != 0
    out, tape = orc.aggregate_fwd(samples, feat, [F, 4, 4], [3, 2], ss, 4, params, "mean", True)
    assert out.shape == (4, 8) and (out < 0).any()                       # identity act on the last layer
    y0 = tape[0][0][3]
    assert (y0 >= 0).all()                                               # relu on layer 0
    self0, means0 = tape[0][0][0], tape[0][0][1]
    np.testing.assert_allclose(y0[:, :4], np.maximum(self0 @ params[0]["self_weights"], 0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(means0, feat[samples[1]].reshape(4, 2, F).mean(1), rtol=1e-6)  # divisor is always s


def test_adam_matches_closed_form_first_step():
    p = np.array([1.0, -2.0], dtype=np.float32)
    g = np.array([0.5, -7.0], dtype=np.float32)
    m, v = np.zeros(2, np.float32), np.zeros(2, np.float32)
    orc.adam_tf_update(p, orc.clip_by_value(g), m, v, 1, 0.01)
    # t=1: m=0.1g, v=0.001g^2, lr_t = lr*sqrt(0.001)/0.1 -> step = lr * g/|g| (up to eps)
    np.testing.assert_allclose(p, [1.0 - 0.01, -2.0 + 0.01], rtol=1e-5)


def test_f1_matches_sklearn():
    from sklearn import metrics
    rng = np.random.default_rng(3)
    y = (rng.random((200, 9)) > 0.6).astype(np.float32)
    p = rng.random((200, 9)).astype(np.float32)
    want = metrics.f1_score(y, (p > 0.5).astype(int), average="micro")
    assert abs(orc.calc_f1_micro(y, p, True) - want) < 1e-9
    yo = np.eye(9)[rng.integers(0, 9, 200)]
    want = metrics.f1_score(yo.argmax(1), p.argmax(1), average="micro")
    assert abs(orc.calc_f1_micro(yo, p, False) - want) < 1e-9


def test_cpu_baseline_port_matches_oracle():
    from oracle.cpu_baseline import CpuSupervisedMean
    rng = np.random.default_rng(0)
    N, F, C = 300, 20, 5
    feat = np.vstack([rng.normal(size=(N, F)), np.zeros((1, F))]).astype(np.float32)
    neigh = [list(rng.choice(N, size=rng.integers(0, 12), replace=False)) for _ in range(N)]
    adj, _ = orc.construct_adj(neigh, 8, rng)
    dims, ns = [F, 8, 8], [4, 3]
    m = CpuSupervisedMean(feat, adj, dims, C, ns, lr=0.01, weight_decay=0.01)
    params = orc.make_supervised_params("mean", dims, C, True, rng)
    m.set_params_from_oracle(params)
    batch = rng.choice(N, 16, replace=False)
    perms = [rng.permutation(8), rng.permutation(8)]
    labels = np.eye(C, dtype=np.float32)[rng.integers(0, C, 16)]
    samples, ss = orc.sample(adj, batch, ns, perms)
    res = orc.supervised_fwd_bwd(params, feat, samples, ss, labels, dims, ns, 16, "mean", True, False, weight_decay=0.01)
    loss, logits, grads = m.train_step(batch, labels, perms)
    assert abs(loss - res["loss"]) < 1e-5
    np.testing.assert_allclose(logits, res["node_preds"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(grads[0].numpy(), res["grads"]["agg"][0]["neigh_weights"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(grads[-2].numpy(), res["grads"]["node_pred"]["weights"], rtol=1e-4, atol=1e-6)


def test_csr_sampler_hash_properties():
    rng = np.random.default_rng(5)
    N = 100
    deg = rng.integers(0, 9, size=N)
    rowptr = np.zeros(N + 1, dtype=np.int64); rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, N, size=int(rowptr[-1])).astype(np.int32)
    ids = rng.integers(0, N + 1, size=64)
    a = sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids, 7, 123, 3, 1)
    b = sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids, 7, 123, 3, 1)
    assert np.array_equal(a, b)                                           # counter-based: pure function
    c = sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids, 7, 123, 4, 1)
    assert not np.array_equal(a, c)                                       # different step -> different draws
    half = sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids[32:], 7, 123, 3, 1, global_row_offset=32)
    assert np.array_equal(a[32:], half)                                   # sharding invariance
    for i, node in enumerate(ids):
        if node == N or deg[node] == 0:
            assert (a[i] == N).all()
        else:
            assert np.isin(a[i], col[rowptr[node]:rowptr[node + 1]]).all()


GOLD_DIR = os.path.dirname(GOLD)


def test_golden_gcn_and_maxpool_calls():
    """One GCN and one MaxPool aggregator call on the tiny graph vs plain-loop arithmetic (make_golden_more.py)."""
    g, m = np.load(GOLD), np.load(os.path.join(GOLD_DIR, "tiny_gcn_maxpool.npz"))
    X, batch, s = g["feats"], g["batch"], 2
    self_vecs = X[batch]
    neigh = X[g["samples1"]].reshape(len(batch), s, X.shape[1])
    y, _ = orc.gcn_aggregator_fwd(self_vecs, neigh, m["W_gcn"], "relu")
    np.testing.assert_allclose(y, m["gcn_out"], rtol=1e-6, atol=1e-6)     # thirds: not exact in fp32
    y, _ = orc.maxpool_aggregator_fwd(self_vecs, neigh, m["W_mlp"], m["b_mlp"], m["W_self"], m["W_neigh"], True, "relu",
                                      pool="max")
    assert np.array_equal(y, m["maxpool_out"])                            # integers: exact


def test_hash_known_answers():
    """mix64 / CSR sampler / dropout mask of oracle/sampler_hash.py vs arbitrary-precision Python ints."""
    k = np.load(os.path.join(GOLD_DIR, "hash_kat.npz"))
    assert np.array_equal(sampler_hash.mix64(k["mix_in"]), k["mix_out"])
    ns, seed, step, hop, row_off, pad = [int(v) for v in k["csr_args"]]
    got = sampler_hash.sample_uniform_csr(k["rowptr"], k["col"], 4, pad, k["ids"], ns, seed, step, hop, row_off)
    assert np.array_equal(got, k["picked"])
    dseed, clock, site, row0, n_rows, d = [int(v) for v in k["drop_args"]]
    rate = float(k["drop_rate"])
    m = sampler_hash.dropout_mask(dseed, clock, site, row0, n_rows, d, rate)
    assert np.array_equal(m > 0, k["keep"] == 1)
    assert np.allclose(m[m > 0], 1.0 / (1.0 - rate))


def test_sampling_law_known_answers():
    """GS_LAW_REFERENCE / GS_LAW_DISTINCT restated in oracle/sampler_hash.py vs the arbitrary-precision plain-loop
    answers of tests/golden/make_law_kat.py (keyed Feistel permutation, virtual padded table, per-call columns)."""
    k = np.load(os.path.join(GOLD_DIR, "law_kat.npz"))
    for name in k.files:
        if name.startswith("perm_") and name != "perm_key":
            n = int(name[5:])
            got = sampler_hash.perm_index(k["perm_key"], np.arange(n, dtype=np.uint64), np.uint64(n))
            assert np.array_equal(got.astype(np.int64), k[name]), name
    M, s, seed, step, hop, row_off, pad = [int(v) for v in k["args"]]
    for law, cap in ((1, M), (2, M), (2, 0)):
        got = sampler_hash.sample_uniform_csr(k["rowptr"], k["col"], 5, pad, k["ids"], s, seed, step, hop, row_off,
                                              law=law, max_degree=cap)
        assert np.array_equal(got, k["picked_law%d_cap%d" % (law, cap)]), (law, cap)


def test_reference_law_is_the_padded_table_law():
    """GS_LAW_REFERENCE == the reference sampler (neigh_samplers.py:24-29: adj[ids][:, perm[:s]]) applied to the VIRTUAL
    padded table, and that table obeys construct_adj (minibatch.py:227-245): rows of nodes with deg > max_degree hold
    max_degree DISTINCT neighbors, deg == max_degree the list itself, deg < max_degree only members of the list,
    deg 0 all pad; frozen across steps; columns distinct within a call and fresh per call."""
    rng = np.random.default_rng(11)
    N, M, s = 300, 16, 5
    deg = rng.integers(0, 60, size=N)
    deg[:5] = [0, M, M + 1, 1, 59]
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = np.concatenate([rng.permutation(N)[:d] for d in deg]).astype(np.int32)     # distinct neighbors per node
    table = sampler_hash.virtual_padded_table(rowptr, col, N, N, 123, M)
    assert table.shape == (N, M)
    for v in range(N):
        nb = col[rowptr[v]:rowptr[v + 1]]
        if deg[v] == 0:
            assert (table[v] == N).all()
        elif deg[v] > M:
            assert len(set(table[v].tolist())) == M and np.isin(table[v], nb).all()
        elif deg[v] == M:
            assert np.array_equal(table[v], nb)
        else:
            assert np.isin(table[v], nb).all()
    ids = rng.integers(0, N + 1, size=200)
    padded = np.vstack([table, np.full((1, M), N, np.int32)])
    seen_cols = set()
    for step in range(4):
        for hop in range(2):
            cols = sampler_hash.call_columns(123, step, hop, s, M)
            assert len(set(cols.tolist())) == s and cols.max() < M
            seen_cols.add(tuple(cols.tolist()))
            got = sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids, s, 123, step, hop, 77, law=1, max_degree=M)
            want = orc.uniform_neighbor_sampler(padded, ids, s, np.concatenate([cols, np.setdiff1d(np.arange(M), cols)]))
            assert np.array_equal(got, want)
            # sharding invariance (the law ignores the global row: shared columns, per-node table)
            assert np.array_equal(got[50:], sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids[50:], s, 123, step, hop,
                                                                            127, law=1, max_degree=M))
    assert len(seen_cols) == 8


def test_reference_law_unsupervised_pass_is_three_reference_sample_calls():
    """models.py:347-357: sample(batch1), sample(batch2), sample(neg_samples) -- three calls of models.py:254-275, i.e.
    2 sampler calls each = SIX independent column permutations per step.  The one-pass restatement over the roots
    [batch1 | batch2 | negatives] with segment call ids g * K + hop equals three orc.sample() runs on the padded table,
    each fed its own permutations, and the six permutations are pairwise different."""
    rng = np.random.default_rng(21)
    N, M, fans, B, NEG = 250, 12, [4, 3], 9, 5
    deg = rng.integers(0, 40, size=N)
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = np.concatenate([rng.permutation(N)[:d] for d in deg]).astype(np.int32)
    padded = np.vstack([sampler_hash.virtual_padded_table(rowptr, col, N, N, 123, M), np.full((1, M), N, np.int32)])
    roots = [rng.integers(0, N, size=B), rng.integers(0, N, size=B), rng.integers(0, N, size=NEG)]
    allroots = np.concatenate(roots).astype(np.int32)
    step, K = 3, len(fans)
    # one pass, segmented call ids
    prev, rows_per_root, got = allroots, 1, []
    for hop, f in enumerate(fans):
        seg = (B * rows_per_root, 2 * B * rows_per_root)
        out = sampler_hash.sample_uniform_csr_segments(rowptr, col, N, N, prev, f, 123, step, hop, K, seg, law=1, max_degree=M)
        got.append(out.reshape(-1))
        prev, rows_per_root = out.reshape(-1), rows_per_root * f
    # three reference sample() calls, each with its own permutations
    perms = {}
    per_group = []
    for g, r in enumerate(roots):
        pg = []
        for hop, f in enumerate(fans):
            cols = sampler_hash.call_columns(123, step, g * K + hop, f, M)
            perms[(g, hop)] = tuple(cols.tolist())
            pg.append(np.concatenate([cols, np.setdiff1d(np.arange(M), cols)]))
        smp, _ = orc.sample(padded, r, fans[::-1], pg)        # num_samples_per_layer: call k uses layer K-1-k
        per_group.append(smp)
    for hop in range(K):
        want = np.concatenate([per_group[g][hop + 1] for g in range(3)])
        assert np.array_equal(got[hop], want), hop
    assert len(set(perms.values())) == 6


def test_distinct_law_properties():
    rng = np.random.default_rng(12)
    N, s = 200, 6
    deg = rng.integers(0, 40, size=N)
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = np.concatenate([rng.permutation(N)[:d] for d in deg]).astype(np.int32)
    ids = np.arange(N)
    for cap in (0, 10):
        a = sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids, s, 5, 2, 0, law=2, max_degree=cap)
        b = sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids, s, 5, 3, 0, law=2, max_degree=cap)
        for v in range(N):
            nb = col[rowptr[v]:rowptr[v + 1]]
            if deg[v] == 0:
                assert (a[v] == N).all()
                continue
            assert np.isin(a[v], nb).all()
            if deg[v] >= s:
                assert len(set(a[v].tolist())) == s                  # without replacement
        assert not np.array_equal(a, b)
        if cap:
            # the frozen max_degree subset: over many steps a capped node only ever shows <= cap distinct neighbors
            v = int(np.argmax(deg))
            seen = set()
            for step in range(60):
                seen.update(sampler_hash.sample_uniform_csr(rowptr, col, N, N, [v], s, 5, step, 0, law=2, max_degree=cap)[0].tolist())
            assert len(seen) == cap < deg[v]
    # below s entries the law falls back to the iid draw
    low = np.where((deg > 0) & (deg < s))[0]
    a0 = sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids, s, 5, 2, 0)
    a2 = sampler_hash.sample_uniform_csr(rowptr, col, N, N, ids, s, 5, 2, 0, law=2)
    assert np.array_equal(a0[low], a2[low])
