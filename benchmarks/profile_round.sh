#!/bin/bash
# Run on the GPU box (via gpurun): GPU tests + bench + rocprofv3 kernel trace + PMC passes.
#   bash benchmarks/profile_round.sh <tag> [tests|notests]
# Outputs under gpurun_out/<tag>/ ; summaries to copy into profiles/ are written as *.md / *.json there.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "${2:-tests}" = "tests" ]; then
  (cd $R && timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log)
fi
timeout 900 python $R/bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --steps 96 --warmup 20 --no-cpu-baseline --no-aux > $O/trace.log 2>&1
python $R/benchmarks/rocpd_stats.py $(ls $O/trace/*/*_results.db $O/trace/*_results.db 2>/dev/null | head -1) --md $O/bench_kernel_stats.md > /dev/null 2>> $O/trace.log
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/bench.py --steps 16 --warmup 20 --no-cpu-baseline --no-aux > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/bench.py --steps 16 --warmup 20 --no-cpu-baseline --no-aux > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/cal -o cal -- python $R/benchmarks/calibrate_fetch.py > $O/cal.log 2>&1
python $R/benchmarks/pmc_traffic.py $(ls $O/pmc_fetch/*/*_results.db $O/pmc_fetch/*_results.db 2>/dev/null | head -1) \
       $(ls $O/pmc_write/*/*_results.db $O/pmc_write/*_results.db 2>/dev/null | head -1) \
       $(ls $O/cal/*/*_results.db $O/cal/*_results.db 2>/dev/null | head -1) $O/k2_pmc.json > $O/pmc_traffic.log 2>&1
python $R/benchmarks/pmc_step_traffic.py $(ls $O/pmc_fetch/*/*_results.db $O/pmc_fetch/*_results.db 2>/dev/null | head -1) \
       $(ls $O/pmc_write/*/*_results.db $O/pmc_write/*_results.db 2>/dev/null | head -1) \
       $(ls $O/cal/*/*_results.db $O/cal/*_results.db 2>/dev/null | head -1) $O/step_traffic.json > $O/step_traffic.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o m -- python $R/bench.py --steps 16 --warmup 20 --no-cpu-baseline --no-aux > $O/pmc_mfma.log 2>&1
python $R/benchmarks/pmc_mfma.py $(ls $O/pmc_mfma/*/*_results.db $O/pmc_mfma/*_results.db 2>/dev/null | head -1) $O/mfma_util.md > /dev/null 2>> $O/pmc_mfma.log
# keep the merged-back payload small: the raw databases stay on the box
find $O -name "*.db" -size +20M -delete
cat $O/bench.json
