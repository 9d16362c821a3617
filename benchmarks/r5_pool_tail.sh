#!/bin/bash
# split-K tail round of the wide pooling-MLP kernel: tests, micro-benchmark over row counts, max-pool step A/B (GS_SPLIT_WIDE_TAIL).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_split_gemm_gpu.py tests/test_bench_parity_gpu.py tests/test_ref_pin_gpu.py -m gpu -x -q -k "split or maxpool or pool" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python -m benchmarks.micro_split pool > $O/micro_pool.json 2> $O/micro_pool.err; cat $O/micro_pool.json
bash benchmarks/r5_ab_env.sh $1 "maxpool:--model graphsage_maxpool --steps 40" -- "GS_SPLIT_WIDE_TAIL=0" "GS_SPLIT_WIDE_TAIL=1" "GS_SPLIT_WIDE_TAIL=0" "GS_SPLIT_WIDE_TAIL=1"
