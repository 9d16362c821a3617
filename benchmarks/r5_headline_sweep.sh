#!/bin/bash
# headline kernel trace + env sweeps in one call:  bash benchmarks/r5_headline_sweep.sh <outdir> "<VAR=val ...>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trh
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trh -o t -- python $R/bench.py --steps 200 --no-cpu-baseline --no-aux > $O/headline_bench.json 2> $O/headline_bench.err
python $R/benchmarks/rocpd_stats.py $(ls $O/trh/*_results.db $O/trh/*/*_results.db 2>/dev/null | head -1) --md $O/headline_kernel_stats.md > /dev/null 2>&1
rm -rf $O/trh
head -8 $O/headline_kernel_stats.md
cd $R
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-aux > $O/sw_$i.json 2> $O/sw_$i.err
  python - "$O/sw_$i.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    e = d.get("ms_per_step_events", {})
    print("[%s]: wall %.2f us/step | events median %.2f | loss %.4f" % (sys.argv[2], d["ms_per_step"] * 1e3, e.get("ms_per_step_median", 0) * 1e3, d["config"]["loss_after"]))
except Exception as ex:
    print("[%s] FAILED: %r" % (sys.argv[2], ex))
PY
done
