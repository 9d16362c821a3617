"""Diagnostics: per-phase timeline of the fused tail kernel inside the real step schedule (needs a -DGS_TIMELINE
build, see timeline_wgrad.py).  GS_COGATHER_TAIL=0 shows the kernel without gather riders."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from graphsage_amd import _lib  # noqa: E402
import test_fullsize_gpu as T  # noqa: E402

NAMES = ["h0 rows -> relu mask bits", "issue of every later phase's operands", "wait for the z helpers + pick z up", "l2norm",
         "logits partials", "logits sum, softmax/loss, dlogits", "d_y = dlogits.Wh^T", "l2norm bwd -> dz", "d_in = dz.W^T", "d_h0 stores"]


def main():
    dev = torch.device("cuda:0")
    DG, model, order = T.build(dev)
    model.train_steps_device(T.B, 25, steps_per_launch=8)
    torch.cuda.synchronize()
    lib = _lib.load()
    NW = 64 * 16 + 256 * 8
    buf = (ctypes.c_ulonglong * NW)()
    assert lib.gs_debug_tail_timeline(buf, NW) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    t = raw[:64 * 16].reshape(64, 16)[:32, :11]
    hp = raw[64 * 16:].reshape(256, 8)[:128, :6]
    t00 = min(t[:, 0].min(), hp[:, 0].min())
    hp = (hp - t00) * 0.01
    print("z helpers (128 workgroups), us after the launch's first stamp:")
    for k, name in enumerate(["entry", "operands landed, A rows in LDS", "MFMAs + partial tiles in LDS", "z stores issued",
                              "z stores acknowledged (all waves)", "arrival counter incremented (returned)"]):
        print("  %-42s min %5.2f  median %5.2f  max %5.2f" % (name, hp[:, k].min(), np.median(hp[:, k]), hp[:, k].max()))
    for term, sl in (("self-term slabs", [i for i in range(128) if i % 4 < 2]), ("neighbor-term slabs", [i for i in range(128) if i % 4 >= 2])):
        print("  %-20s operands landed median %5.2f, counter done median %5.2f" % (term, np.median(hp[sl, 1]), np.median(hp[sl, 5])))
    t = (t - t00) * 0.01
    print("riders: split3 %.2f tail %.2f; block start %.1f..%.1f us, end %.1f..%.1f us" % (
        model.cogather_split3, model.cogather_tail, t[:, 0].min(), t[:, 0].max(), t[:, 10].min(), t[:, 10].max()))
    d = np.diff(t, axis=1)
    for k, name in enumerate(NAMES):
        print("  phase %d %-38s mean %5.2f us  (min %5.2f max %5.2f)" % (k, name, d[:, k].mean(), d[:, k].min(), d[:, k].max()))


if __name__ == "__main__":
    main()
