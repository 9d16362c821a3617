#!/bin/bash
# kernel-trace summaries of every aux configuration of bench.py + the dominant-kernel table bench.py reads:
#   bash benchmarks/trace_aux_all.sh <outdir> <tag>   ->  <outdir>/<tag>_{maxpool,unsup,gcn,rmat}_kernel_stats.md, <tag>_aux_dominant.json
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-$R/gpurun_out/aux}; TAG=${2:-r05}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in maxpool unsup gcn rmat; do
  rm -rf $O/trca
  case $cfg in
    maxpool) ARGS="--model graphsage_maxpool --steps 32";;
    unsup) ARGS="--unsupervised --steps 64";;
    gcn) ARGS="--model gcn --steps 64";;
    rmat) ARGS="--workload rmat --steps 64";;
  esac
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/trca -o t -- python $R/bench.py $ARGS --no-cpu-baseline --no-aux > $O/${TAG}_${cfg}_bench.json 2> $O/${TAG}_${cfg}_bench.err
  python $R/benchmarks/rocpd_stats.py $(ls $O/trca/*_results.db $O/trca/*/*_results.db 2>/dev/null | head -1) --md $O/${TAG}_${cfg}_kernel_stats.md > /dev/null 2>&1
  echo "$cfg: $(python -c "import json; d=json.load(open('$O/${TAG}_${cfg}_bench.json')); print('%.1f us/step (under the kernel trace)' % (d['ms_per_step']*1e3))" 2>&1 | tail -1)"
done
rm -rf $O/trca
cd $R && python benchmarks/aux_dominant.py $O/${TAG}_aux_dominant.json graphsage_maxpool=$O/${TAG}_maxpool_kernel_stats.md unsupervised=$O/${TAG}_unsup_kernel_stats.md gcn=$O/${TAG}_gcn_kernel_stats.md rmat=$O/${TAG}_rmat_kernel_stats.md
