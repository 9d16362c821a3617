#!/bin/bash
# Diagnostics: the same bench configuration N times -- loss_after must repeat bit for bit (a race between launches / riders shows
# as a run that differs).   bash benchmarks/determinism.sh <N> "<ENV=val ...>" <bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; ENVS=$2; shift 2
cd $R
for i in $(seq 1 $N); do
  env $ENVS python bench.py "$@" --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.10f' % d['config'].get('loss_after'))"
done | sort | uniq -c | sed "s|^|[$ENVS] |"
