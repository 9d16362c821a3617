#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_split_gemm_gpu.py -m gpu -x -q -s > $O/pytest_split.log 2>&1
echo "pytest split rc=$?"; grep -E "two fp16|passed|failed|Error" $O/pytest_split.log | tail -12
for dma in 1 0 1 0; do echo "GS_SPLIT16_DMA=$dma: $(GS_SPLIT16_DMA=$dma timeout 100 python -m benchmarks.micro_split pool16 2>/dev/null | tail -1)"; done
timeout 600 python -m pytest tests/test_bench_parity_gpu.py tests/test_ref_pin_gpu.py tests/test_model_gpu.py tests/test_config_parity_gpu.py -m gpu -x -q -k "pool" > $O/pytest_pool.log 2>&1
echo "pytest pool rc=$?"; tail -3 $O/pytest_pool.log
bash benchmarks/r5_ab_env.sh $1 "maxpool:--model graphsage_maxpool --steps 40" -- "GS_POOL_F16=0" "GS_SPLIT16_DMA=0" "GS_SPLIT16_DMA=1" "GS_POOL_F16=0" "GS_SPLIT16_DMA=0" "GS_SPLIT16_DMA=1"
