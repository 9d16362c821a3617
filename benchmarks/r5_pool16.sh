#!/bin/bash
# two-piece fp16 pooling MLP: tests, micro-benchmark, max-pool step A/B (GS_POOL_F16).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_split_gemm_gpu.py -m gpu -x -q -s > $O/pytest_split.log 2>&1
echo "pytest split rc=$?"; grep -E "max .error|passed|failed|Error|error" $O/pytest_split.log | tail -24
timeout 300 python -m benchmarks.micro_split pool > $O/micro_pool.json 2> $O/micro_pool.err; grep -E "split16|split_tiled_ws_8|split_tiled_8|GF" $O/micro_pool.json; tail -3 $O/micro_pool.err
timeout 600 python -m pytest tests/test_bench_parity_gpu.py tests/test_ref_pin_gpu.py tests/test_model_gpu.py tests/test_config_parity_gpu.py -m gpu -x -q -k "pool" > $O/pytest_pool.log 2>&1
echo "pytest pool rc=$?"; tail -3 $O/pytest_pool.log
bash benchmarks/r5_ab_env.sh $1 "maxpool:--model graphsage_maxpool --steps 40" -- "GS_POOL_F16=0" "GS_POOL_F16=1" "GS_POOL_F16=0" "GS_POOL_F16=1"
