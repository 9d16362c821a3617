"""Diagnostics: the split-K slab count of every variable in the step bench.py runs (what the optimizer launch sums):
    python benchmarks/debug_slabs.py [bench.py arguments]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd import engine as eng
orig = eng.Engine._var_descs
seen = [0]
def patched(self):
    if seen[0] < 3:
        seen[0] += 1
        sys.stderr.write("SLABS " + " | ".join("%s %dx%d ns=%d" % (v.name, v.rows, v.cols, v.n_slabs) for v in self.variables) + "\n")
    return orig(self)
eng.Engine._var_descs = patched
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
