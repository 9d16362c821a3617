#!/bin/bash
# same-call A/B of environment settings over bench configurations:
#   bash benchmarks/ab_env.sh <outdir> "<cfg name>:<bench args>" ... -- "<VAR=val ...>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd $R
cfgs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do cfgs+=("$1"); shift; done
shift
for cfg in "${cfgs[@]}"; do
  name=${cfg%%:*}; args=${cfg#*:}
  i=0
  for envs in "$@"; do
    i=$((i+1))
    env $envs timeout 300 python bench.py $args --warmup 5 --no-cpu-baseline --no-aux > $O/ab_${name}_$i.json 2> $O/ab_${name}_$i.err
    python - "$O/ab_${name}_$i.json" "$name" "$envs" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    e = d.get("ms_per_step_events", {})
    print("%-9s [%s]: wall %.2f us/step | events median %.2f | loss %.4f" % (sys.argv[2], sys.argv[3], d["ms_per_step"] * 1e3, e.get("ms_per_step_median", 0) * 1e3, d["config"].get("loss_after", float("nan"))))
except Exception as ex:
    print("%s [%s] FAILED: %r" % (sys.argv[2], sys.argv[3], ex))
PY
  done
done
