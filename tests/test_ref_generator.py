"""The committed tests/golden/ref_*.npz ARE what the generator writes when it executes /root/reference (skipped where the
reference is absent, e.g. on the GPU box), and the TF1 stand-in the generator runs on behaves like TensorFlow 1.x on the
documented points the reference relies on."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GRAPHSAGE_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "graphsage")), reason="reference sources not present")
def test_generator_reproduces_committed_fixtures(tmp_path):
    gen = os.path.join(HERE, "golden", "make_ref_fixtures.py")
    subprocess.run([sys.executable, gen], check=True, env=dict(os.environ, REF_FIXTURE_DIR=str(tmp_path)),
                   stdout=subprocess.DEVNULL)
    names = sorted(f for f in os.listdir(tmp_path) if f.startswith("ref_"))
    committed = sorted(f for f in os.listdir(os.path.join(HERE, "golden")) if f.startswith("ref_"))
    assert names == committed and len(names) >= 13
    for f in names:
        a, b = np.load(os.path.join(tmp_path, f)), np.load(os.path.join(HERE, "golden", f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            if a[k].dtype.kind == "f":        # BLAS thread counts may differ between runs of the generator
                np.testing.assert_allclose(a[k], b[k], rtol=1e-5 if a[k].dtype == np.float32 else 1e-12,
                                           atol=1e-6 if a[k].dtype == np.float32 else 1e-13, err_msg=f + ":" + k)
            else:
                assert np.array_equal(a[k], b[k]), f + ":" + k


def test_nothing_in_the_product_imports_the_shim_or_the_fixtures():
    root = os.path.dirname(HERE)
    for d, _, files in os.walk(os.path.join(root, "graphsage_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert "tf1_shim" not in src and "import tensorflow" not in src and "ref_fixtures" not in src, f
    src = open(os.path.join(root, "bench.py")).read()
    assert "tf1_shim" not in src and "import tensorflow" not in src


@pytest.fixture
def tf():
    sys.path.insert(0, os.path.join(HERE, "tf1_shim"))
    try:
        import tensorflow as tf
        tf.reset_default_graph()
        tf.shim.set_real("float64")
        yield tf
        tf.shim.set_real("float32")
    finally:
        sys.path.remove(os.path.join(HERE, "tf1_shim"))


def test_shim_session_semantics_and_adam(tf):
    """One run evaluates every node once; variable updates land after the fetches; TF's Adam:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t), var -= lr_t*m/(sqrt(v)+eps) -- first step moves by ~lr*sign(g)."""
    x = tf.placeholder(tf.float32, shape=(None,))
    w = tf.Variable(np.asarray([1.0, -2.0, 0.5]), name="w")
    loss = tf.reduce_sum(w * w * x)
    opt = tf.train.AdamOptimizer(learning_rate=0.1)
    gv = opt.compute_gradients(loss)
    step = opt.apply_gradients(gv)
    sess = tf.Session()
    _, l0, w0, g0 = sess.run([step, loss, w, gv[0][0]], feed_dict={x: [1.0, 2.0, 3.0]})
    assert np.allclose(w0, [1.0, -2.0, 0.5]) and np.isclose(l0, 1 + 8 + 0.75)        # pre-update values
    assert np.allclose(g0, [2.0, -8.0, 3.0])
    w1 = sess.run(w)
    g = np.asarray([2.0, -8.0, 3.0])
    m, v = 0.1 * g, 0.001 * g * g
    want = np.asarray([1.0, -2.0, 0.5]) - 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9) * m / (np.sqrt(v) + 1e-8)
    assert np.allclose(w1, want, rtol=1e-12)
    assert np.allclose(w1, np.asarray([1.0, -2.0, 0.5]) - 0.1 * np.sign(g), atol=1e-6)


def test_shim_documented_op_semantics(tf):
    sess = tf.Session()
    x = tf.constant(np.asarray([[-3.0, 0.0, 2.0]]))
    z = tf.constant(np.asarray([[1.0, 0.0, 1.0]]))
    got = sess.run(tf.nn.sigmoid_cross_entropy_with_logits(labels=z, logits=x))
    want = -(np.asarray([1, 0, 1]) * np.log(1 / (1 + np.exp([3.0, 0, -2.0]))) +
             np.asarray([0, 1, 0]) * np.log(1 - 1 / (1 + np.exp([3.0, 0, -2.0]))))
    assert np.allclose(got, want)
    got = sess.run(tf.nn.softmax_cross_entropy_with_logits(labels=tf.constant(np.asarray([[0.0, 1.0, 0.0]])), logits=x))
    assert np.allclose(got, -np.log(np.exp(0.0) / np.exp([-3.0, 0.0, 2.0]).sum()))
    # top_k: equal values come out lower index first
    vals, idx = tf.nn.top_k(tf.constant(np.asarray([[1.0, 3.0, 3.0, 0.0, 3.0]])), k=5)
    assert sess.run(idx).tolist() == [[1, 2, 4, 0, 3]]
    # l2_normalize clamps the squared norm at epsilon; l2_loss halves
    tiny = tf.constant(np.asarray([[1e-9, 0.0]]))
    assert np.allclose(sess.run(tf.nn.l2_normalize(tiny, 1)), [[1e-9 / 1e-6, 0.0]])
    assert np.isclose(sess.run(tf.nn.l2_loss(tf.constant(np.asarray([3.0, 4.0])))), 12.5)
    # reduce_max shares the gradient among tied maxima
    v = tf.Variable(np.asarray([[2.0, 5.0, 5.0]]), name="v")
    g = tf.gradients(tf.reduce_sum(tf.reduce_max(v, axis=1)), v)
    assert np.allclose(sess.run(g), [[0.0, 0.5, 0.5]])
    # dropout with keep_prob 1 is the identity; random_shuffle permutes dimension 0 with the injected permutation
    assert np.array_equal(sess.run(tf.nn.dropout(x, 1 - tf.placeholder_with_default(0., shape=()))), sess.run(x))
    tf.shim.inject_shuffle.append([2, 0, 1])
    assert sess.run(tf.random_shuffle(tf.constant(np.asarray([10, 11, 12])))).tolist() == [12, 10, 11]
    # python scalars, shapes and slices as the reference uses them
    t = tf.constant(np.arange(12.0).reshape(3, 4))
    dims = tf.shape(t)
    assert sess.run(tf.reshape(t, (dims[0] * dims[1],))).shape == (12,)
    assert sess.run(tf.slice(t, [0, 0], [-1, 2])).shape == (3, 2)
    assert np.isclose(sess.run(0 + 2.0 * tf.reduce_sum(t) / tf.cast(dims[0], tf.float32)), 2 * 66 / 3)
