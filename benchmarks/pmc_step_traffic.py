"""HBM-side traffic of every launch of the training step from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
bench.py (separate passes; FETCH_SIZE calibrated on a known-byte gather as in pmc_traffic.py).

    python benchmarks/pmc_step_traffic.py <fetch.db> <write.db> <calibration.db> <out.json>
"""
import json
import sqlite3
import sys

import numpy as np

KNOWN_CAL_BYTES = 232966 * 608 * 4
STEP_KERNELS = ["sage_tiled3_fwd_kernel", "sage_stream_fwd_kernel", "sage_tail_kernel", "wgrad_tiled3_kernel", "stream_wgrad_kernel", "flat_reduce_adam_kernel", "sample_fanout_kernel"]


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, grid_size, value, duration from counters_collection where counter_name=?", (counter,)).fetchall()
    out = {}
    for k, g, v, d in rows:
        for name in STEP_KERNELS:
            if name in k:
                out.setdefault((name, int(g)), []).append((float(v), float(d)))
    return out


def main():
    fetch_db, write_db, cal_db, out = sys.argv[1:5]
    cur = sqlite3.connect(cal_db).cursor()
    cal = [v for k, v in cur.execute("select kernel_name, value from counters_collection where counter_name='FETCH_SIZE'").fetchall()
           if "gather_mean_kernel<1" in k]
    factor = KNOWN_CAL_BYTES / (float(np.median(cal)) * 1024.0)
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    res = {"fetch_calibration_factor": factor, "launches": []}
    for key in sorted(set(f) | set(w)):
        fv = f.get(key, [])
        wv = w.get(key, [])
        rd = float(np.mean([v for v, _ in fv])) * 1024.0 * factor if fv else 0.0
        wr = float(np.mean([v for v, _ in wv])) * 1024.0 if wv else 0.0
        dur = float(np.mean([d for _, d in fv])) / 1e3 if fv else float("nan")
        res["launches"].append({"kernel": key[0], "grid": key[1], "launches_seen": len(fv), "hbm_read_MB": rd / 1e6, "hbm_write_MB": wr / 1e6,
                                "avg_us_under_pmc": dur, "TBps": (rd + wr) / (dur * 1e-6) / 1e12 if dur == dur and dur > 0 else None})
    # the kernels these counters belong to: digest of the library's sources (graphsage_amd/_C/build.stamp); bench.py marks the
    # profile stale when it differs from the library it runs
    import os
    stamp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "graphsage_amd", "_C", "build.stamp")
    res["lib_digest"] = open(stamp).read().strip() if os.path.exists(stamp) else None
    json.dump(res, open(out, "w"), indent=1)
    for r in res["launches"]:
        print("%-28s grid %8d  x%3d  read %7.1f MB  write %6.1f MB  %6.1f us  %s TB/s" % (
            r["kernel"], r["grid"], r["launches_seen"], r["hbm_read_MB"], r["hbm_write_MB"], r["avg_us_under_pmc"],
            "%.2f" % r["TBps"] if r["TBps"] else "-"))


if __name__ == "__main__":
    main()
