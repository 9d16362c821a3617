"""Duration of the peer-store gradient exchange (gs_peer_*) with all ranks on ONE device (no xGMI: this is the protocol's
own latency -- two dependent counter hand-overs + three passes over the slices), next to RCCL's 1-rank all-reduce of the
same buffer.  Run with GPU_MAX_HW_QUEUES=16 (every in-process rank needs a hardware queue of its own).
    GPU_MAX_HW_QUEUES=16 python benchmarks/peer_exchange.py [n_floats]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from graphsage_amd import ops  # noqa: E402
from test_peer_gpu import _ranks  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 230121
    dev = torch.device("cuda:0")
    out = {"n_floats": n, "bytes": 4 * n, "ranks_on_one_device": {}}
    import time
    reps = 50
    for world in (1, 2, 4):
        for chunks in (0, 8, 32):
            rs = _ranks(n, world, dev, chunks=chunks, spin_limit=1 << 17)     # a stalled wait costs ~0.2 s, not seconds
            for warm in range(2):
                for r in rs:
                    r.launch()
                for r in rs:
                    r.stream.sync()
            t0 = time.perf_counter()
            for it in range(reps):                                            # `reps` exchanges queued back to back per rank
                for r in rs:
                    r.launch()
            for r in rs:
                r.stream.sync()
            dt = (time.perf_counter() - t0) / reps * 1e6
            ok = all(r.status() == (reps + 2, 0) for r in rs)
            for r in rs:
                r.close()
            out["ranks_on_one_device"]["world%d_chunks%d" % (world, chunks)] = {"us_per_exchange": dt, "ok": ok}
            sys.stderr.write("world %d chunks %d: %.1f us ok=%s\n" % (world, chunks, dt, ok))
    try:
        h = ctypes.c_void_p()
        buf = (ctypes.c_uint8 * 128)()
        ops.call("gs_comm_unique_id", ctypes.addressof(buf), 128)
        ops.call("gs_comm_init_rank", ctypes.byref(h), 1, 0, ctypes.addressof(buf), 128)
        s = ops.Stream()
        x = torch.zeros(n, device=dev)
        ops.call("gs_comm_allreduce_sum_f32", h.value, ops.ptr(x), n, s.handle)
        s.sync()
        t0 = time.perf_counter()
        for it in range(reps):
            ops.call("gs_comm_allreduce_sum_f32", h.value, ops.ptr(x), n, s.handle)
        s.sync()
        out["rccl_1_rank_us"] = (time.perf_counter() - t0) / reps * 1e6
    except Exception as ex:
        out["rccl_1_rank_us"] = repr(ex)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
