#!/bin/bash
# Same-call A/B of two builds of the library over the timed configurations:  bash benchmarks/lib_ab.sh <outdir> <libA.so|""> <libB.so|"">
# ("" = the in-tree library).  Prints us/step (HIP-event median) per configuration and library.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift
mkdir -p $O
cd $R
for cfg in "headline:--steps 200" "gcn:--model gcn --steps 96" "rmat:--workload rmat --steps 96" "unsup:--unsupervised --steps 64" "maxpool:--model graphsage_maxpool --steps 32"; do
  name=${cfg%%:*}; args=${cfg#*:}
  i=0
  for lib in "$@"; do
    i=$((i+1))
    if [ -n "$lib" ]; then export GS_LIB=$R/$lib; else unset GS_LIB; fi
    timeout 300 python bench.py $args --warmup 5 --no-cpu-baseline --no-aux > $O/lab_${name}_$i.json 2> $O/lab_${name}_$i.err
    python - "$O/lab_${name}_$i.json" "$name" "${lib:-in-tree}" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    e = d.get("ms_per_step_events", {})
    print("%-9s %-40s wall %8.2f us/step | events median %8.2f (p10 %.2f p90 %.2f)" % (sys.argv[2], sys.argv[3], d["ms_per_step"] * 1e3,
          e.get("ms_per_step_median", 0) * 1e3, e.get("ms_per_step_p10", 0) * 1e3, e.get("ms_per_step_p90", 0) * 1e3))
except Exception as ex:
    print("%s %s FAILED: %r" % (sys.argv[2], sys.argv[3], ex))
PY
  done
done
unset GS_LIB
