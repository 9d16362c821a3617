#!/bin/bash
# Round 6: the LDS-tiled bf16x3 grouped weight gradients (GS_TILED3_WGRAD) against the fp32-MFMA stream kernel, same call, every
# configuration; then the rider shares with the tiled kernel (its launch hosts fewer riders: 153 VGPRs = 3 waves per SIMD, two of
# them the host's)
bash benchmarks/ab_env.sh $1 "head:--steps 96" "rmat:--workload rmat --steps 64" "gcn:--model gcn --steps 64" "unsup:--unsupervised --steps 64" "maxpool:--model graphsage_maxpool --steps 32" -- "GS_TILED3_WGRAD=0" "GS_TILED3_WGRAD=1" "GS_TILED3_WGRAD=0" "GS_TILED3_WGRAD=1"
mv ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/$1/ab.txt ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/$1/ab_main.txt 2>/dev/null
bash benchmarks/ab_env.sh $1 "head:--steps 96" "rmat:--workload rmat --steps 64" "gcn:--model gcn --steps 64" -- "GS_COGATHER_SPLIT3=0.20 GS_COGATHER_TAIL=0.55" "GS_COGATHER_SPLIT3=0.20 GS_COGATHER_TAIL=0.65" "GS_COGATHER_SPLIT3=0.25 GS_COGATHER_TAIL=0.60" "GS_COGATHER_SPLIT3=0.25 GS_COGATHER_TAIL=0.75" "GS_COGATHER_SPLIT3=0.15 GS_COGATHER_TAIL=0.60"
mv ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/$1/ab.txt ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/$1/ab_shares.txt 2>/dev/null
bash benchmarks/ab_env.sh $1 "unsup:--unsupervised --steps 64" -- "GS_COGATHER_LP_FWD=0.25 GS_COGATHER_LP_TAIL=0.30 GS_COGATHER_LP_NEG=0.15" "GS_COGATHER_LP_FWD=0.30 GS_COGATHER_LP_TAIL=0.30 GS_COGATHER_LP_NEG=0.20" "GS_COGATHER_LP_FWD=0.25 GS_COGATHER_LP_TAIL=0.35 GS_COGATHER_LP_NEG=0.10"
mv ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/$1/ab.txt ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/$1/ab_unsup_shares.txt 2>/dev/null
