#!/bin/bash
# MFMA utilisation of the max-pool step's launches (one --pmc pass), both arithmetics of the pooling MLP.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  GS_POOL_F16=$f timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma_$f -o m -- python $R/bench.py --model graphsage_maxpool --steps 8 --warmup 5 --no-cpu-baseline --no-aux > $O/pmc_mfma_$f.log 2>&1
  python $R/benchmarks/pmc_mfma.py $(ls $O/pmc_mfma_$f/*/*_results.db $O/pmc_mfma_$f/*_results.db 2>/dev/null | head -1) $O/${TAG:-r06}_mfma_util_maxpool_f16_$f.md > /dev/null 2>> $O/pmc_mfma_$f.log
  echo "== GS_POOL_F16=$f"; head -8 $O/${TAG:-r06}_mfma_util_maxpool_f16_$f.md
done
find $O -name "*.db" -size +20M -delete
