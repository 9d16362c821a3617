#!/bin/bash
# Diagnostics: determinism.sh on the tree of an older commit (benchmarks/probes/old_tree.bin = `git archive <rev>`), built on the box.
#   bash benchmarks/determinism_old.sh <N> "<ENV>" <bench args>
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/old_tree && mkdir -p /tmp/old_tree && tar -xf $R/benchmarks/probes/old_tree.bin -C /tmp/old_tree
cp $R/benchmarks/determinism.sh /tmp/old_tree/benchmarks/determinism.sh
cd /tmp/old_tree && python -m graphsage_amd.build > /dev/null 2>&1
GRAFT_REPO_ROOT=/tmp/old_tree bash /tmp/old_tree/benchmarks/determinism.sh "$@"
