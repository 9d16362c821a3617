#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel trace + PMC passes.  Outputs under gpurun_out/final/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/bench.py --steps 20 --warmup 20 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/bench.py --steps 20 --warmup 20 --no-cpu-baseline > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/cal -o cal -- python $R/benchmarks/calibrate_fetch.py > $O/cal.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_mp -o trace -- python $R/bench.py --steps 24 --warmup 10 --no-cpu-baseline --model graphsage_maxpool > $O/trace_mp.log 2>&1
timeout 200 python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --model gcn > $O/bench_gcn.json 2>/dev/null
timeout 200 python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --unsupervised > $O/bench_unsup.json 2>/dev/null
cat $O/bench.json
