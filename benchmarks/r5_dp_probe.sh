#!/bin/bash
# Round-5 probe of the data-parallel step SCHEDULES on one GPU (a stand-in wait in the collective's place), one gpurun call:
#   GS_PROBE_DP_SCHEDULE=<us>  three launches: slab sum | sleeping wave (the collective) | Adam          (round 4's schedule)
#   GS_PROBE_DP_PEER=<us>      ONE launch: slab sum | exchange workgroups holding their hand-over <us> | Adam (gs_peer_step),
#                              GS_COGATHER_DP_OPT=<share> of the gather riding behind the exchange workgroups
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-$R/gpurun_out/dp_probe}
mkdir -p $O
cd $R
i=0
for cfg in "" "GS_PROBE_DP_SCHEDULE=0" "GS_PROBE_DP_SCHEDULE=18" "GS_PROBE_DP_SCHEDULE=30" "GS_PROBE_DP_PEER=0" "GS_PROBE_DP_PEER=18" "GS_PROBE_DP_PEER=30" \
           "GS_PROBE_DP_PEER=18 GS_COGATHER_DP_OPT=0.10" "GS_PROBE_DP_PEER=18 GS_COGATHER_DP_OPT=0.20" "GS_PROBE_DP_PEER=30 GS_COGATHER_DP_OPT=0.20" "GS_PROBE_DP_PEER=30 GS_COGATHER_DP_OPT=0.30"; do
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
