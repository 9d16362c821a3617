"""The peer-store gradient exchange between TWO PROCESSES on one device (windows mapped with hipIpcOpenMemHandle; no xGMI):
microseconds per exchange for several chunk counts, HIP events around 200 back-to-back exchanges per process.
    python benchmarks/peer_exchange_2proc.py [n_floats]"""
import json
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Shell(object):
    """The three things the hook needs from an engine."""

    def __init__(self, n, dev):
        from graphsage_amd import ops
        self.device = dev
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self._stream = ops.Stream()
        self.stream = self._stream.handle

    def sync(self):
        self._stream.sync()


def worker(rank, world, port, n, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": "0", "GS_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    sys.path.insert(0, ROOT)
    import faulthandler
    faulthandler.dump_traceback_later(120, exit=True)
    import torch.distributed as dist
    from graphsage_amd import distributed as gsd, ops
    gsd.init_from_env()
    dev = torch.device("cuda:0")
    e = Shell(n, dev)
    res = {}
    keep = []
    for chunks in (2, 4, 2, 8, 2, 32, 0, 2, 4, 16, 2, 32, 0, 8, 2, 4, 2, 2, 32, 0, 2, 2, 4, 4, 8, 8, 2, 0, 0, 2):
        hook = gsd.PeerPushAllReduce(e, chunks=chunks, spin_limit=1 << 19)
        try:
            hook.self_test()
            reps = 200
            dist.barrier()
            a, b = ops.Event(), ops.Event()
            a.record(e.stream)
            for _ in range(reps):
                hook.all_reduce(e.grads)
            b.record(e.stream)
            e.sync()
            hook.check()
            res.setdefault("chunks%d" % chunks, []).append(round(a.elapsed_ms(b) * 1e3 / reps, 1))
            res["self_test_attempts"] = res.get("self_test_attempts", []) + [hook.attempts]
            if hook.failures:
                res["self_test_failures"] = res.get("self_test_failures", []) + [(chunks, f[:160]) for f in hook.failures]
        except Exception as ex:
            res.setdefault("chunks%d" % chunks, []).append(repr(ex)[45:110])
        dist.barrier()
        if os.environ.get("PEER_KEEP_WINDOWS") == "1":
            keep.append(hook)          # windows are never freed: no address (and IPC handle) is ever reused
        else:
            hook.close()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 230121
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    got = []
    while len(got) < 2:
        try:
            got.append(q.get(timeout=1.0))
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise SystemExit("a worker died: %r" % [p.exitcode for p in procs])
    res = dict(got)
    for p in procs:
        p.join(timeout=30)
    print("JSON " + json.dumps({"n_floats": n, "bytes": 4 * n, "two_processes_one_device_us_per_exchange": res[0], "rank1": res[1]}))


if __name__ == "__main__":
    main()
