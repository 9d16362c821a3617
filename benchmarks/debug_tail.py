"""Debug: fused tail vs unfused chain on identical inputs (same init, same batches), several steps; reports where
d_h0 / gradients differ."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from graphsage_amd import engine as eng  # noqa: E402
import test_bench_parity_gpu as T  # noqa: E402


def run(fuse, steps, device_epoch):
    G, it, model, order = T.build("mean")
    model.fuse_tail = fuse
    out = []
    for t in range(steps):
        if device_epoch:
            model.train_step_device(T.B, fetch=True)
        else:
            b = order[t * T.B:(t + 1) * T.B]
            model.train_step({model.placeholders['batch']: b, model.placeholders['labels']: it.label_matrix[b],
                              model.placeholders['batch_size']: T.B})
        e = eng.get_engine()
        e.sync()
        dh0 = e._ws[("m", (model.name, "d_hidden", 0), 5632, 256)].numpy().copy()
        h0 = model._tape[0][4].numpy().copy()
        out.append((dh0, h0, e.grads.cpu().numpy().copy(), e.params.cpu().numpy().copy(), [(v.name, v.offset, v.size) for v in e.variables]))
    return out


for device_epoch in (False,):
    a = run(True, 3, device_epoch)
    b = run(False, 3, device_epoch)
    for t in range(3):
        dA, hA, gA, pA, vs = a[t]
        dB, hB, gB, pB, _ = b[t]
        if t > 0:
            pd = np.abs(a[t - 1][3] - b[t - 1][3])
            print("   params before this step: max diff %.3e at %d (n > 1e-6: %d)" % (pd.max(), int(pd.argmax()), int((pd > 1e-6).sum())))
        rr, cc = np.where(np.abs(dA - dB) > 1e-6 + 1e-4 * np.abs(dB))
        for r_, c_ in list(zip(rr, cc))[:6]:
            print("   entry", r_, c_, "h0 fused %.6e unfused %.6e  d_h0 fused %.6e unfused %.6e" % (hA[r_, c_], hB[r_, c_], dA[r_, c_], dB[r_, c_]))
        bad = np.abs(dA - dB) > 1e-6 + 1e-4 * np.abs(dB)
        print("device_epoch", device_epoch, "step", t, "d_h0 mismatches", int(bad.sum()), "of", bad.size,
              "rows", np.unique(np.where(bad)[0])[:20], "cols", np.unique(np.where(bad)[1])[:20])
        for name, off, size in vs:
            d = np.abs(gA[off:off + size] - gB[off:off + size]).max()
            print("   grad", name, "max abs diff %.3e" % d, "max |g| %.3e" % np.abs(gB[off:off + size]).max())
