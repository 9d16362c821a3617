#!/bin/bash
# Rider-share sweep (host-side knobs only): us/step of the headline step and of the unsupervised step.
#   bash benchmarks/sweep_shares.sh > gpurun_out/sweep_shares.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
one() {   # label, extra bench args; env comes from the caller
  timeout 120 python $R/bench.py --no-cpu-baseline --no-aux --steps 192 $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'] * 1e3, 1))"
}
for t in 0.4 0.5 0.6; do for f in 0.1 0.15 0.2; do
  GS_COGATHER_TAIL=$t GS_COGATHER_SPLIT3=$f one "sup tail=$t fwd=$f"
done; done
for s in 0.4 0.5 0.6; do for z in 0.0 0.04; do
  GS_COGATHER_SPLIT=$s GS_COGATHER_Z=$z one "unsup split=$s z=$z" --unsupervised
done; done
