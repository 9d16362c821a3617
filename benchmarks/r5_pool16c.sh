#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_split_gemm_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "split or stream" > $O/pytest_split.log 2>&1
echo "pytest split rc=$?"; tail -2 $O/pytest_split.log
echo "pool16: $(timeout 100 python -m benchmarks.micro_split pool16 2>/dev/null | tail -1)"
timeout 600 python -m pytest tests/test_bench_parity_gpu.py tests/test_ref_pin_gpu.py tests/test_model_gpu.py tests/test_config_parity_gpu.py -m gpu -x -q -k "pool" > $O/pytest_pool.log 2>&1
echo "pytest pool rc=$?"; tail -3 $O/pytest_pool.log
bash benchmarks/r5_ab_env.sh $1 "maxpool:--model graphsage_maxpool --steps 40" -- "GS_POOL_F16=0" "GS_STREAM_FWD_POOL=0" "GS_SPLIT_WIDE_TAIL=0" "GS_X=default" "GS_POOL_F16=0" "GS_STREAM_FWD_POOL=0" "GS_SPLIT_WIDE_TAIL=0" "GS_X=default"
