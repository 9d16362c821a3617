"""MFMA utilisation per kernel from ONE rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE):

    util = MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 32 CUs * 4 SIMDs)

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over every SIMD of the chip; GRBM_GUI_ACTIVE comes back summed over the 8
XCDs (checked against the kernel durations: 1.56e7 "cycles" for an 835 us launch = 8 x 2.0e6 cycles at 2.4 GHz), so
one XCD's 128 SIMDs are the denominator.  A v_mfma_f32_32x32x2_f32 occupies its SIMD for 64 cycles = 4096 flops, so
busy/64*4096 must reproduce the kernel's algorithmic flops -- printed as a cross-check (1.75e9 for the layer-0
forward, 8.3e10 for the MaxPool MLP forward).

    python benchmarks/pmc_mfma.py <results.db> <out.md>
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db, out = sys.argv[1:3]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, grid_size, counter_name, value from counters_collection").fetchall()
    acc = defaultdict(lambda: defaultdict(list))
    for k, g, c, v in rows:
        acc[(k.split("(")[0][:70], g)][c].append(v)
    lines = ["| kernel [grid] | launches | MFMA busy cycles / launch | GUI active cycles / launch | MFMA util | flops implied (busy/64*4096) |",
             "|---|---|---|---|---|---|"]
    for (k, g), c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))):
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("GRBM_GUI_ACTIVE", [])
        if not busy or not act or sum(busy) == 0:
            continue
        b, a = sum(busy) / len(busy), sum(act) / len(act)
        lines.append("| %s [%d] | %d | %.3g | %.3g | %.1f %% | %.3g |" % (k, g, len(busy), b, a, 100.0 * b / (a * 128.0),
                                                                       b / 64.0 * 4096.0))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
