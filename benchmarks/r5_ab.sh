#!/bin/bash
# Round-5 A/B of the headline step inside ONE gpurun call (box-to-box variance is +-3 %):  bash benchmarks/r5_ab.sh <outdir> "<VAR=val ...>" ...
# every quoted argument is one configuration's environment; prints us/step (wall, K = 20 and K = 200) and the event median.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift
mkdir -p $O
cd $R
i=0
for cfg in "$@"; do
  i=$((i+1))
  for K in 20 200; do
    env $cfg timeout 300 python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-aux > $O/ab_${i}_K$K.json 2> $O/ab_${i}_K$K.err
    python - "$O/ab_${i}_K$K.json" "$cfg" $K <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    e = d.get("ms_per_step_events", {})
    print("[%s] K=%s: wall %.2f us/step | events median %.2f (p10 %.2f p90 %.2f) | K2 %.1f us | loss %.4f" % (
        sys.argv[2], sys.argv[3], d["ms_per_step"] * 1e3, e.get("ms_per_step_median", 0) * 1e3, e.get("ms_per_step_p10", 0) * 1e3,
        e.get("ms_per_step_p90", 0) * 1e3, d["roofline"]["avg_launch_us"], d["config"]["loss_after"]))
except Exception as ex:
    print("[%s] K=%s FAILED: %r" % (sys.argv[2], sys.argv[3], ex))
PY
  done
done
