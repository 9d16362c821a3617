#!/bin/bash
# Probe of the data-parallel step SCHEDULE on one GPU (a stand-in wait in the collective's place), one gpurun call:
#   GS_PROBE_DP_SCHEDULE=<us>  the in-graph schedule: slab sum (+ sampler) | sleeping wave (the collective) | clip + Adam
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-$R/gpurun_out/dp_probe}; shift
mkdir -p $O
cd $R
i=0
for cfg in "" "GS_PROBE_DP_SCHEDULE=0" "GS_PROBE_DP_SCHEDULE=10" "GS_PROBE_DP_SCHEDULE=18" "GS_PROBE_DP_SCHEDULE=30" "$@"; do
  i=$((i+1))
  env $cfg timeout 300 python bench.py --steps 96 --warmup 5 --no-cpu-baseline --no-aux --steps-per-launch 8 > $O/dp_$i.json 2> $O/dp_$i.err
  python - "$O/dp_$i.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    e = d.get("ms_per_step_events", {})
    dp = d.get("dp_schedule") or {}
    print("[%-48s] wall %.2f us/step | events %.2f | hook %s | loss %.4f" % (sys.argv[2] or "single-GPU step (8 steps per launch)", d["ms_per_step"] * 1e3,
          e.get("ms_per_step_median", 0) * 1e3, dp.get("allreduce"), d["config"]["loss_after"]))
except Exception as ex:
    print("[%s] FAILED: %r" % (sys.argv[2], ex))
PY
done
