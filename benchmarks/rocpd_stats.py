"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table.

    python benchmarks/rocpd_stats.py gpurun_out/prof/r_results.db [--md out.md]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    md = sys.argv[sys.argv.index("--md") + 1] if "--md" in sys.argv else None
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end, grid_x from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e, gx in rows:
        name = name.split("(")[0] + " [grid %d]" % gx
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (name[:100], a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / total))
    out = "\n".join(lines)
    print(out)
    if md:
        with open(md, "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
