#!/bin/bash
# One gpurun call of a round's record: [GPU test suite] + the driver's own bench command + headline / aux kernel traces
# [+ the PMC passes of profile_round.sh].  Files are named <tag>_* (TAG, default r06).
#   [TAG=r06] bash benchmarks/round.sh <outdir> [tests] [pmc]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
TAG=${TAG:-r06}
mkdir -p $O
cd $R
for a in "$@"; do case $a in tests) TESTS=1;; pmc) PMC=1;; esac; done
if [ -n "$TESTS" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
  echo "pytest rc=$?"; tail -3 $O/pytest.log
fi
# the driver's command, twice (the first one also pays the page-in of a fresh box)
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_driver_bench.json 2> $O/${TAG}_driver_bench.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-aux > $O/${TAG}_driver_bench2.json 2>> $O/${TAG}_driver_bench.err
python - <<P
import json
for f in ("${TAG}_driver_bench.json", "${TAG}_driver_bench2.json"):
    try:
        d = json.load(open("$O/" + f)); e = d.get("ms_per_step_events", {})
        print(f, "wall %.2f us/step, events median %.2f" % (d["ms_per_step"] * 1e3, e.get("ms_per_step_median", 0) * 1e3),
              {k: round(v["ms_per_step"] * 1e3, 1) for k, v in d.get("aux", {}).items() if "ms_per_step" in v})
    except Exception as x:
        print(f, "unreadable", x)
P
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trh
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trh -o t -- python $R/bench.py --steps 96 --warmup 20 --no-cpu-baseline --no-aux > $O/${TAG}_headline_bench.json 2> $O/${TAG}_headline_bench.err
python $R/benchmarks/rocpd_stats.py $(ls $O/trh/*_results.db $O/trh/*/*_results.db 2>/dev/null | head -1) --md $O/${TAG}_bench_kernel_stats.md > /dev/null 2>&1
rm -rf $O/trh
head -30 $O/${TAG}_bench_kernel_stats.md
bash $R/benchmarks/trace_aux_all.sh $O $TAG
if [ -n "$PMC" ]; then
  bash $R/benchmarks/profile_round.sh $(basename $O)/pmc notests > $O/profile_round.log 2>&1
  tail -3 $O/profile_round.log
fi
