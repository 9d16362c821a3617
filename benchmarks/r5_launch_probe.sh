#!/bin/bash
# Where do the ~100 us between the wall clock and the HIP events of the K = 20 timed call go?  Same-call A/B of host-side knobs.
#   bash benchmarks/r5_launch_probe.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
i=0
for envs in "GS_BENCH_WARM_CALLS=0" "GS_BENCH_WARM_CALLS=1" "GS_BENCH_WARM_CALLS=2" "GS_BENCH_WARM_CALLS=4" "GS_X=default" "GS_BENCH_WARM_CALLS=8"; do
  i=$((i+1))
  env GS_BENCH_LAUNCH_PROBE=1 $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-aux > $O/lp_$i.json 2> $O/lp_$i.err
  python - "$O/lp_$i.json" "$envs" <<'PY' | tee -a $O/lp.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    e = d.get("ms_per_step_events", {})
    print("[%s]: wall %.2f us/step | events median %.2f | probe %s" % (sys.argv[2], d["ms_per_step"] * 1e3, e.get("ms_per_step_median", 0) * 1e3,
          {k: (round(v, 1) if not isinstance(v, list) else v) for k, v in d.get("launch_probe", {}).items()}))
except Exception as ex:
    print("[%s] FAILED: %r" % (sys.argv[2], ex))
PY
done
