#!/bin/bash
# L2 (TCC) request / hit / miss counters per launch of the headline step and of K2 ("verify with TCP->TCC
# request counters", "K2's TCC hit rate").  Counters in their own run (no trace domains).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/pmc_tcc -o t -- python $R/bench.py --steps 16 --warmup 20 --no-cpu-baseline --no-aux > $O/pmc_tcc.log 2>&1
DB=$(ls $O/pmc_tcc/*/*_results.db $O/pmc_tcc/*_results.db 2>/dev/null | head -1)
python $R/benchmarks/pmc_kernel.py $DB 2>&1 | grep -E "stream_fwd|stream_wgrad|sage_tail|flat_reduce|gather_mean|sample_fanout" | tee $O/r05_tcc_counters.txt
timeout 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $O/pmc_tcp -o t -- python $R/bench.py --steps 16 --warmup 20 --no-cpu-baseline --no-aux > $O/pmc_tcp.log 2>&1
DB=$(ls $O/pmc_tcp/*/*_results.db $O/pmc_tcp/*_results.db 2>/dev/null | head -1)
python $R/benchmarks/pmc_kernel.py $DB 2>&1 | grep -E "stream_fwd|stream_wgrad|sage_tail|flat_reduce|gather_mean|sample_fanout" | tee -a $O/r05_tcc_counters.txt
tail -3 $O/pmc_tcp.log
find $O -name "*.db" -size +20M -delete
