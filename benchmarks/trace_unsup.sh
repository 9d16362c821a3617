#!/bin/bash
# kernel trace of the unsupervised bench configuration under the given env:  bash benchmarks/trace_unsup.sh <out.md> VAR=val ...
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trcu
env "$@" rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trcu -o t -- python $R/bench.py --unsupervised --steps 64 --no-cpu-baseline --no-aux > $R/gpurun_out/trcu.json 2>/dev/null
echo "== $* : $(python -c "import json; d=json.load(open('$R/gpurun_out/trcu.json')); print('%.2f us/step' % (d['ms_per_step']*1e3))")"
python $R/benchmarks/rocpd_stats.py $(ls $R/gpurun_out/trcu/*_results.db $R/gpurun_out/trcu/*/*_results.db 2>/dev/null | head -1) --md $OUT 2>/dev/null | grep -v "at::\|rocprim\|rocclr" | sed -n 3,14p | cut -c1-120
rm -rf $R/gpurun_out/trcu
