#!/bin/bash
# Round 6: loads in flight per RIDER wave of the two tiled launches: 13 (variant build, still three waves per SIMD) against 8
R=${GRAFT_REPO_ROOT:-/root/repo}
U13="GS_LIB=$R/benchmarks/probes/_lib/libgs_ru13.so"
bash benchmarks/ab_env.sh $1 "head:--steps 96" "gcn:--model gcn --steps 64" "rmat:--workload rmat --steps 64" "unsup:--unsupervised --steps 64" "maxpool:--model graphsage_maxpool --steps 32" -- "GS_TILED3_FWD=1" "$U13" "GS_TILED3_FWD=1" "$U13"
