#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_split_gemm_gpu.py -m gpu -x -q -k "unique or dedup or pool or split" > $O/pytest_dd.log 2>&1
echo "pytest dd rc=$?"; tail -2 $O/pytest_dd.log
timeout 600 python -m pytest tests/test_bench_parity_gpu.py tests/test_ref_pin_gpu.py tests/test_model_gpu.py tests/test_config_parity_gpu.py tests/test_driver_gpu.py -m gpu -x -q -k "pool" > $O/pytest_pool.log 2>&1
echo "pytest pool rc=$?"; tail -2 $O/pytest_pool.log
bash benchmarks/r5_ab_env.sh $1 "maxpool:--model graphsage_maxpool --steps 40" -- "GS_POOL_F16=0" "GS_X=default" "GS_POOL_F16=0" "GS_X=default"
