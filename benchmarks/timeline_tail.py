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
    buf = (ctypes.c_ulonglong * (64 * 16))()
    assert lib.gs_debug_tail_timeline(buf, 64 * 16) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)[:32, :11]
    t = (t - t[:, 0].min()) * 0.01
    print("riders: split3 %.2f tail %.2f; block start %.1f..%.1f us, end %.1f..%.1f us" % (
        model.cogather_split3, model.cogather_tail, t[:, 0].min(), t[:, 0].max(), t[:, 10].min(), t[:, 10].max()))
    d = np.diff(t, axis=1)
    for k, name in enumerate(NAMES):
        print("  phase %d %-38s mean %5.2f us  (min %5.2f max %5.2f)" % (k, name, d[:, k].mean(), d[:, k].min(), d[:, k].max()))


if __name__ == "__main__":
    main()
