#!/bin/bash
# kernel trace of the MaxPool bench configuration (BASELINE configs[2])
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trcm
env "$@" rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trcm -o t -- python $R/bench.py --model graphsage_maxpool --steps 32 --no-cpu-baseline --no-aux > $R/gpurun_out/trcm.json 2>/dev/null
echo "== maxpool $* : $(python -c "import json; d=json.load(open('$R/gpurun_out/trcm.json')); print('%.2f us/step' % (d['ms_per_step']*1e3))")"
python $R/benchmarks/rocpd_stats.py $(ls $R/gpurun_out/trcm/*_results.db | head -1) 2>/dev/null | grep -v "at::\|rocprim\|rocclr" | sed -n 3,20p
rm -rf $R/gpurun_out/trcm
