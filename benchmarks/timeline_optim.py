"""Diagnostics: where the optimizer launch (flat_reduce_adam_kernel) of the real step schedule spends its time.  Needs a
-DGS_TIMELINE build of gs_optim.hip:
    bash benchmarks/probes/build_variant.sh opt_tl gs_optim.hip -DGS_TIMELINE
    GS_LIB=benchmarks/probes/_lib/libgs_opt_tl.so python benchmarks/timeline_optim.py
Stamps are those of the LAST optimizer launch of the replayed graph (wall clock, 100 MHz)."""
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


def main():
    dev = torch.device("cuda:0")
    DG, model, order = T.build(dev)
    model.train_steps_device(T.B, 24, steps_per_launch=8)
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = (ctypes.c_ulonglong * (1024 * 4))()
    lib.gs_debug_opt_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.gs_debug_opt_timeline(buf, 1024 * 4) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4).astype(np.int64)
    opt = t[:512]
    opt = opt[opt[:, 0] > 0]
    rid = t[512:]
    rid = rid[(rid[:, 0] > 0) & (rid[:, 1] > 0)]
    t0 = min(opt[:, 0].min(), rid[:, 0].min() if len(rid) else opt[:, 0].min())
    o = (opt - t0) * 0.01
    print("optimizer workgroups: %d" % len(o))
    for k, name in enumerate(["entry", "descriptors known (slab requests next)", "slabs landed", "stores issued (end)"]):
        print("  %-42s min %5.2f  median %5.2f  max %5.2f us" % (name, o[:, k].min(), np.median(o[:, k]), o[:, k].max()))
    d = np.diff(o, axis=1)
    for k, name in enumerate(["entry -> descriptors", "slab round trip", "Adam + stores"]):
        print("  %-42s min %5.2f  median %5.2f  max %5.2f us" % (name, d[:, k].min(), np.median(d[:, k]), d[:, k].max()))
    if len(rid):
        r = (rid[:, :2] - t0) * 0.01
        print("rider workgroups seen: %d  entry %5.2f..%5.2f  end %5.2f..%5.2f  (duration median %.2f max %.2f us)" % (
            len(r), r[:, 0].min(), r[:, 0].max(), r[:, 1].min(), r[:, 1].max(), np.median(r[:, 1] - r[:, 0]), (r[:, 1] - r[:, 0]).max()))


if __name__ == "__main__":
    main()
