#!/bin/bash
# GPU box: sweep the share of the prefetch gather that rides with the layer-0 forward (rest: grouped wgrad launch)
for f in 1.0 0.7 0.55 0.45 0.35 0.2; do
  echo "split=$f: $(GS_COGATHER_SPLIT=$f timeout 200 python bench.py --steps 160 --warmup 24 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done
