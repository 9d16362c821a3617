#!/bin/bash
# Round-5 A/B of the RMAT configuration (configs[4]) inside ONE gpurun call:  bash benchmarks/r5_rmat_ab.sh <outdir> "<VAR=val ...>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift
mkdir -p $O
cd $R
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 300 python bench.py --workload rmat --steps 64 --warmup 5 --no-cpu-baseline --no-aux > $O/rab_$i.json 2> $O/rab_$i.err
  python - "$O/rab_$i.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    e = d.get("ms_per_step_events", {})
    print("[%s]: wall %.2f us/step | events median %.2f | loss %.4f" % (sys.argv[2], d["ms_per_step"] * 1e3, e.get("ms_per_step_median", 0) * 1e3, d["config"]["loss_after"]))
except Exception as ex:
    print("[%s] FAILED: %r" % (sys.argv[2], ex))
PY
done
