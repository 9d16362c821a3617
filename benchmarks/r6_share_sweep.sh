#!/bin/bash
# Round 6: rider shares with the LDS-tiled layer-0 forward (its workgroups hold a CU each: the share it hosts best may have moved)
bash benchmarks/ab_env.sh $1 "head:--steps 96" -- \
  "GS_X=0" "GS_COGATHER_SPLIT3=0.05" "GS_COGATHER_SPLIT3=0.10" "GS_COGATHER_SPLIT3=0.20" "GS_COGATHER_SPLIT3=0.25" \
  "GS_COGATHER_SPLIT3=0.10 GS_COGATHER_TAIL=0.55" "GS_COGATHER_SPLIT3=0.05 GS_COGATHER_TAIL=0.55" "GS_COGATHER_SPLIT3=0.10 GS_COGATHER_TAIL=0.45" \
  "GS_COGATHER_SPLIT3=0.20 GS_COGATHER_TAIL=0.45" "GS_COGATHER_SPLIT3=0.0 GS_COGATHER_TAIL=0.6" "GS_X=0"
bash benchmarks/ab_env.sh $1 "unsup:--unsupervised --steps 64" -- \
  "GS_X=0" "GS_COGATHER_LP_FWD=0.20" "GS_COGATHER_LP_FWD=0.25" "GS_COGATHER_LP_FWD=0.35" "GS_COGATHER_LP_FWD=0.40" "GS_COGATHER_LP_FWD=0.15 GS_COGATHER_LP_TAIL=0.30"
bash benchmarks/ab_env.sh $1 "rmat:--workload rmat --steps 64" -- "GS_X=0" "GS_COGATHER_SPLIT3=0.0 GS_COGATHER_TAIL=0.7" "GS_COGATHER_SPLIT3=0.3 GS_COGATHER_TAIL=0.4"
