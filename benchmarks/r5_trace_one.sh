#!/bin/bash
# kernel trace of one bench configuration:  bash benchmarks/r5_trace_one.sh <outdir> <tag> <bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; TAG=$2; shift; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trx
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trx -o t -- python $R/bench.py "$@" --no-cpu-baseline --no-aux > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python $R/benchmarks/rocpd_stats.py $(ls $O/trx/*_results.db $O/trx/*/*_results.db 2>/dev/null | head -1) --md $O/${TAG}_kernel_stats.md > /dev/null 2>&1
rm -rf $O/trx
head -${HEADN:-26} $O/${TAG}_kernel_stats.md
