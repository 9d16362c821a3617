#!/bin/bash
# kernel-trace summaries (markdown) of the MaxPool and unsupervised bench configurations -> gpurun_out/<tag>_{maxpool,unsup}_kernel_stats.md
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cfg in maxpool unsup; do
  rm -rf $R/gpurun_out/trca
  if [ $cfg = maxpool ]; then ARGS="--model graphsage_maxpool --steps 32"; else ARGS="--unsupervised --steps 64"; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trca -o t -- python $R/bench.py $ARGS --no-cpu-baseline --no-aux > $R/gpurun_out/${TAG}_${cfg}_bench.json 2>/dev/null
  python $R/benchmarks/rocpd_stats.py $(ls $R/gpurun_out/trca/*_results.db | head -1) --md $R/gpurun_out/${TAG}_${cfg}_kernel_stats.md > /dev/null 2>&1
  echo "$cfg: $(python -c "import json; d=json.load(open('$R/gpurun_out/${TAG}_${cfg}_bench.json')); print('%.1f us/step' % (d['ms_per_step']*1e3))")"
done
rm -rf $R/gpurun_out/trca
