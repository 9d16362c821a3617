#!/bin/bash
# Round 6, stage (a) of the persistent step: tests of the one-launch layer 0 + tail, then same-call A/B on headline / RMAT / GCN
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_parity_gpu.py -x -q -m gpu -k "one_launch or layer0_inside or fused_tail" > $O/pytest_new.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_new.log
bash benchmarks/ab_env.sh $1 "head:--steps 96" "rmat:--workload rmat --steps 64" "gcn:--model gcn --steps 64" -- "GS_FUSE_FWD_TAIL=0" "GS_FUSE_FWD_TAIL=1" "GS_FUSE_FWD_TAIL=0" "GS_FUSE_FWD_TAIL=1"
