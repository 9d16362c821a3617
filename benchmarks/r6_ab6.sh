#!/bin/bash
# Round 6: rider shares with the four-wave tiled weight gradients (more of the gather in ITS launch), and its ring depth (variant builds)
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/$1
bash benchmarks/ab_env.sh $1 "head:--steps 96" "gcn:--model gcn --steps 64" "rmat:--workload rmat --steps 64" -- "GS_TILED3_WGRAD=1" "GS_COGATHER_SPLIT3=0.15 GS_COGATHER_TAIL=0.45" "GS_COGATHER_SPLIT3=0.15 GS_COGATHER_TAIL=0.40" "GS_COGATHER_SPLIT3=0.10 GS_COGATHER_TAIL=0.50" "GS_COGATHER_SPLIT3=0.10 GS_COGATHER_TAIL=0.45" "GS_COGATHER_SPLIT3=0.10 GS_COGATHER_TAIL=0.40" "GS_COGATHER_SPLIT3=0.05 GS_COGATHER_TAIL=0.45" "GS_LIB=$R/benchmarks/probes/_lib/libgs_w3ns5.so" "GS_LIB=$R/benchmarks/probes/_lib/libgs_w3ns6.so" "GS_TILED3_WGRAD=1"
mv $D/ab.txt $D/ab_shares.txt 2>/dev/null
bash benchmarks/ab_env.sh $1 "unsup:--unsupervised --steps 64" -- "GS_TILED3_WGRAD=1" "GS_COGATHER_LP_FWD=0.15 GS_COGATHER_LP_TAIL=0.25 GS_COGATHER_LP_NEG=0.10" "GS_COGATHER_LP_FWD=0.20 GS_COGATHER_LP_TAIL=0.20 GS_COGATHER_LP_NEG=0.10" "GS_COGATHER_LP_FWD=0.15 GS_COGATHER_LP_TAIL=0.20 GS_COGATHER_LP_NEG=0.05" "GS_COGATHER_LP_FWD=0.20 GS_COGATHER_LP_TAIL=0.25 GS_COGATHER_LP_NEG=0.05" "GS_LIB=$R/benchmarks/probes/_lib/libgs_w3ns5.so" "GS_LIB=$R/benchmarks/probes/_lib/libgs_w3ns6.so"
mv $D/ab.txt $D/ab_unsup_shares.txt 2>/dev/null
bash benchmarks/ab_env.sh $1 "maxpool:--model graphsage_maxpool --steps 32" -- "GS_TILED3_WGRAD=1" "GS_COGATHER_SPLIT=0.4" "GS_COGATHER_SPLIT=0.3" "GS_COGATHER_SPLIT=0.6"
mv $D/ab.txt $D/ab_maxpool_shares.txt 2>/dev/null
