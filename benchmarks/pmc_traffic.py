"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (+ the calibration pass) into per-launch HBM traffic of
the K2 gather+mean kernel, as MI355X_MICROARCH.md §HBM prescribes (separate passes, FETCH_SIZE calibrated on a
known byte count in the same access pattern; on gfx950 it reads ~1/2 of a wide coalesced stream).

    python benchmarks/pmc_traffic.py <fetch.db> <write.db> <calibration.db> <out.json>
"""
import json
import sqlite3
import sys

import numpy as np

KNOWN_CAL_BYTES = 232966 * 608 * 4   # calibrate_fetch.py: every row (19 whole 128-B lines) read exactly once


def values(db, counter, pred):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, grid_size, value, duration from counters_collection where counter_name=?",
                       (counter,)).fetchall()
    return [(v, d) for k, g, v, d in rows if pred(k, g)]


def main():
    fetch_db, write_db, cal_db, out = sys.argv[1:5]
    hop2 = lambda k, g: "gather_mean_kernel<8" in k and g == 983040      # [5120 x 25] rows, 3 chunks, 4 waves/block
    f = values(fetch_db, "FETCH_SIZE", hop2)
    w = values(write_db, "WRITE_SIZE", hop2)
    cal = values(cal_db, "FETCH_SIZE", lambda k, g: "gather_mean_kernel<1" in k)
    cal_kb = float(np.median([v for v, _ in cal]))
    factor = KNOWN_CAL_BYTES / (cal_kb * 1024.0)
    fetch_b = float(np.mean([v for v, _ in f])) * 1024.0
    write_b = float(np.mean([v for v, _ in w])) * 1024.0
    res = {
        "kernel": "gather_mean_kernel<8> hop-2 [5120 x 25 rows of 602 fp32]",
        "launches_fetch_pass": len(f), "launches_write_pass": len(w),
        "FETCH_SIZE_bytes_raw": fetch_b, "WRITE_SIZE_bytes": write_b,
        "calibration": {"known_bytes": KNOWN_CAL_BYTES, "FETCH_SIZE_bytes_reported": cal_kb * 1024.0, "factor": factor},
        "hbm_read_bytes_corrected": fetch_b * factor,
        "traffic_bytes_per_launch": fetch_b * factor + write_b,
        "algorithmic_bytes_per_launch": 5120 * 25 * 602 * 4 + 5120 * 25 * 4 + 5120 * 602 * 4,
        "avg_duration_us_under_pmc": float(np.mean([d for _, d in f])) / 1e3,
    }
    res["traffic_over_algorithmic"] = res["traffic_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
    # the kernels these counters belong to: digest of the library's sources (graphsage_amd/_C/build.stamp); bench.py marks the
    # profile stale when it differs from the library it runs
    import os
    stamp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "graphsage_amd", "_C", "build.stamp")
    res["lib_digest"] = open(stamp).read().strip() if os.path.exists(stamp) else None
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
