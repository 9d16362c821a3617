#!/bin/bash
# Diagnostics: hunt a rare run-to-run difference.  Runs a configuration N times; if the losses differ, repeats it under each
# variant environment to see which schedule piece the difference follows.   bash benchmarks/determinism_bisect.sh <N> "<base env>" "<variant env>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; BASE=$2; shift 2
cd $R
out=$(bash benchmarks/determinism.sh $N "$BASE" --steps 200)
echo "$out"
if [ $(echo "$out" | wc -l) -gt 1 ]; then
  for v in "$@"; do bash benchmarks/determinism.sh $((2 * N)) "$BASE $v" --steps 200; done
else
  echo "(no difference in $N runs on this box)"
fi
