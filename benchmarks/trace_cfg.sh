#!/bin/bash
# kernel trace of bench.py under the given env settings:  bash benchmarks/trace_cfg.sh VAR=val ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trc
env "$@" rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trc -o t -- python $R/bench.py --steps 96 --no-cpu-baseline --no-aux > $R/gpurun_out/trc.json 2>/dev/null
echo "== $* : $(python -c "import json; d=json.load(open('$R/gpurun_out/trc.json')); print('%.2f us/step' % (d['ms_per_step']*1e3))")"
python $R/benchmarks/rocpd_stats.py $(ls $R/gpurun_out/trc/*_results.db | head -1) 2>/dev/null | grep -v "at::\|rocprim\|rocclr\|gather_mean_kernel" | sed -n 3,9p
rm -rf $R/gpurun_out/trc
