"""Steady-state rate of the reference-style driver (graphsage_amd.supervised_train, --synthetic reddit): per-step time
between two printed lines derived from the driver's own running average `time=` (cumulative time = avg * steps), for the
device path (default) and the per-step feed_dict path.
    python benchmarks/driver_rate.py"""
import io
import json
import os
import re
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(path, print_every):
    from graphsage_amd import engine as eng
    from graphsage_amd import supervised_train as st
    eng.reset_engine()
    buf = io.StringIO()
    with redirect_stdout(buf):
        st.main(["--synthetic", "reddit", "--epochs", "4", "--max_total_steps", "900", "--print_every", str(print_every),
                 "--feed_path", path, "--base_log_dir", "gpurun_out/drv_" + path])
    pts = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"Iter: (\d+) .* time= (\d+\.\d+)", buf.getvalue())]
    # cumulative time after total_steps+1 steps = avg * (total_steps + 1); iterations restart per epoch, so use the index
    cum = [(i * print_every + 1, avg * (i * print_every + 1)) for i, (_, avg) in enumerate(pts)]
    (n0, t0), (n1, t1) = cum[len(cum) // 2], cum[-1]
    return (t1 - t0) / (n1 - n0) * 1e6


res = {}
for path in ("device", "host"):
    for pe in (5, 50):
        res["%s_print_every_%d_us_per_step" % (path, pe)] = run(path, pe)
        print(path, pe, "%.1f us/step" % res["%s_print_every_%d_us_per_step" % (path, pe)], flush=True)
edges = 512 * 260
res["edges_per_s"] = {k: edges / (v * 1e-6) for k, v in res.items()}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/driver_rate.json", "w"), indent=1)
print(json.dumps(res))
