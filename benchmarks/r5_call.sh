#!/bin/bash
# One gpurun call of round 5: GPU test suite, then the same-call library A/B (benchmarks/r5_lib_ab.sh).
#   bash benchmarks/r5_call.sh <outdir> [pytest args]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q "$@" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
bash benchmarks/r5_lib_ab.sh $O benchmarks/probes/_lib/libgs_prev.so "" | tee $O/lab.txt
