"""Weight initialisers with the reference's ranges (graphsage/inits.py:9-30), drawn on the host
with NumPy (seeded like supervised_train.py:20-22) and uploaded once."""
import numpy as np

_rng = np.random.RandomState(123)


def set_seed(seed):
    global _rng
    _rng = np.random.RandomState(seed)


def uniform(shape, scale=0.05):
    """inits.py:9-12"""
    return _rng.uniform(-scale, scale, size=shape).astype(np.float32)


def glorot(shape):
    """U(-r, r), r = sqrt(6/(fan_in+fan_out))  -- inits.py:15-20; also the range of
    tf.contrib.layers.xavier_initializer() used by Dense (layers.py:94-96)."""
    r = np.sqrt(6.0 / (shape[0] + shape[1]))
    return _rng.uniform(-r, r, size=shape).astype(np.float32)


def zeros(shape):
    """inits.py:22-25"""
    return np.zeros(shape, dtype=np.float32)


def ones(shape):
    """inits.py:27-30"""
    return np.ones(shape, dtype=np.float32)
