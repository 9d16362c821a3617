#!/bin/bash
# HBM-side traffic of every launch of the step (PMC passes only):  bash benchmarks/profile_step_traffic.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/bench.py --steps 16 --warmup 20 --no-cpu-baseline --no-aux > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/bench.py --steps 16 --warmup 20 --no-cpu-baseline --no-aux > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/cal -o cal -- python $R/benchmarks/calibrate_fetch.py > $O/cal.log 2>&1
python $R/benchmarks/pmc_step_traffic.py $(ls $O/pmc_fetch/*/*_results.db $O/pmc_fetch/*_results.db 2>/dev/null | head -1) \
       $(ls $O/pmc_write/*/*_results.db $O/pmc_write/*_results.db 2>/dev/null | head -1) \
       $(ls $O/cal/*/*_results.db $O/cal/*_results.db 2>/dev/null | head -1) $O/step_traffic.json
find $O -name "*.db" -delete
