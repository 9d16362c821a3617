#!/bin/bash
# Round 6: the layer-0 forward as four-wave workgroups (50 KB of LDS: riders share the hosts' CUs) against the eight-wave build
# (benchmarks/probes/_lib/libgs_fwd8.so = the previous commit's gs_split.hip), same call; then the forward's rider share
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/$1
F8="GS_LIB=$R/benchmarks/probes/_lib/libgs_fwd8.so"
bash benchmarks/ab_env.sh $1 "head:--steps 96" "gcn:--model gcn --steps 64" "rmat:--workload rmat --steps 64" "unsup:--unsupervised --steps 64" "maxpool:--model graphsage_maxpool --steps 32" -- "$F8" "GS_TILED3_FWD=1" "$F8" "GS_TILED3_FWD=1"
mv $D/ab.txt $D/ab_main.txt 2>/dev/null
bash benchmarks/ab_env.sh $1 "head:--steps 96" "gcn:--model gcn --steps 64" "rmat:--workload rmat --steps 64" -- "GS_COGATHER_SPLIT3=0.15 GS_COGATHER_TAIL=0.40" "GS_COGATHER_SPLIT3=0.20 GS_COGATHER_TAIL=0.40" "GS_COGATHER_SPLIT3=0.25 GS_COGATHER_TAIL=0.40" "GS_COGATHER_SPLIT3=0.30 GS_COGATHER_TAIL=0.40" "GS_COGATHER_SPLIT3=0.20 GS_COGATHER_TAIL=0.35" "GS_COGATHER_SPLIT3=0.25 GS_COGATHER_TAIL=0.35" "GS_COGATHER_SPLIT3=0.30 GS_COGATHER_TAIL=0.30" "GS_COGATHER_SPLIT3=0.20 GS_COGATHER_TAIL=0.45" "GS_COGATHER_SPLIT3=0.25 GS_COGATHER_TAIL=0.45"
mv $D/ab.txt $D/ab_shares.txt 2>/dev/null
bash benchmarks/ab_env.sh $1 "unsup:--unsupervised --steps 64" -- "GS_COGATHER_LP_FWD=0.25 GS_COGATHER_LP_TAIL=0.25 GS_COGATHER_LP_NEG=0.10" "GS_COGATHER_LP_FWD=0.30 GS_COGATHER_LP_TAIL=0.25 GS_COGATHER_LP_NEG=0.10" "GS_COGATHER_LP_FWD=0.30 GS_COGATHER_LP_TAIL=0.20 GS_COGATHER_LP_NEG=0.10" "GS_COGATHER_LP_FWD=0.35 GS_COGATHER_LP_TAIL=0.20 GS_COGATHER_LP_NEG=0.05"
mv $D/ab.txt $D/ab_unsup_shares.txt 2>/dev/null
