#!/bin/bash
# A/B inside one gpurun call: the max-pool training step with the fp32-MFMA pooling GEMM (GS_SPLIT_POOL=0) vs the split-MFMA one
for v in 0 1; do
  GS_SPLIT_POOL=$v timeout 300 python bench.py --steps 64 --warmup 10 --no-cpu-baseline --model graphsage_maxpool --no-aux 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GS_SPLIT_POOL=$v maxpool %.1f us/step' % (d['ms_per_step']*1e3))"
done
