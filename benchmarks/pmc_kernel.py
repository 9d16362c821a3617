"""Per-kernel averages of every counter in a rocprofv3 --pmc results database.
    python benchmarks/pmc_kernel.py <results.db> [kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select kernel_name, grid_size, counter_name, value from counters_collection").fetchall()
acc = defaultdict(lambda: defaultdict(list))
for k, g, c, v in rows:
    if sub in k:
        acc[(k.split("(")[0][:60], g)][c].append(v)
for (k, g), cs in acc.items():
    print(k, g, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, "launches", len(next(iter(cs.values()))))
