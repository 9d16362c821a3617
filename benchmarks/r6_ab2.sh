#!/bin/bash
# Round 6: the LDS-tiled bf16x3 layer-0 forward (GS_TILED3_FWD) against the fp32-MFMA stream kernel, same call, every configuration
bash benchmarks/ab_env.sh $1 "head:--steps 96" "rmat:--workload rmat --steps 64" "gcn:--model gcn --steps 64" "unsup:--unsupervised --steps 64" "maxpool:--model graphsage_maxpool --steps 32" -- "GS_TILED3_FWD=0" "GS_TILED3_FWD=1" "GS_TILED3_FWD=0" "GS_TILED3_FWD=1"
