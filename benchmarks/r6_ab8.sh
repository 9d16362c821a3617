#!/bin/bash
# Round 6: rider shares with the four-wave forward -- max-pool (two hosting launches), RMAT, and the headline once more
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/$1
bash benchmarks/ab_env.sh $1 "maxpool:--model graphsage_maxpool --steps 32" -- "GS_COGATHER_SPLIT=0.5" "GS_COGATHER_SPLIT=0.6" "GS_COGATHER_SPLIT=0.7" "GS_COGATHER_SPLIT=0.8" "GS_COGATHER_SPLIT=0.4" "GS_LIB=$R/benchmarks/probes/_lib/libgs_fwd8.so"
mv $D/ab.txt $D/ab_maxpool.txt 2>/dev/null
bash benchmarks/ab_env.sh $1 "rmat:--workload rmat --steps 64" -- "GS_COGATHER_SPLIT3=0.15 GS_COGATHER_TAIL=0.50" "GS_COGATHER_SPLIT3=0.10 GS_COGATHER_TAIL=0.50" "GS_COGATHER_SPLIT3=0.10 GS_COGATHER_TAIL=0.60" "GS_COGATHER_SPLIT3=0.05 GS_COGATHER_TAIL=0.55" "GS_COGATHER_SPLIT3=0.15 GS_COGATHER_TAIL=0.60" "GS_LIB=$R/benchmarks/probes/_lib/libgs_fwd8.so"
mv $D/ab.txt $D/ab_rmat.txt 2>/dev/null
bash benchmarks/ab_env.sh $1 "head:--steps 96" "gcn:--model gcn --steps 64" -- "GS_COGATHER_SPLIT3=0.25 GS_COGATHER_TAIL=0.40" "GS_COGATHER_SPLIT3=0.25 GS_COGATHER_TAIL=0.35" "GS_COGATHER_SPLIT3=0.30 GS_COGATHER_TAIL=0.35" "GS_COGATHER_SPLIT3=0.35 GS_COGATHER_TAIL=0.35" "GS_COGATHER_SPLIT3=0.35 GS_COGATHER_TAIL=0.30" "GS_COGATHER_SPLIT3=0.25 GS_COGATHER_TAIL=0.40"
mv $D/ab.txt $D/ab_head.txt 2>/dev/null
