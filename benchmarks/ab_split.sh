#!/bin/bash
# max-pool step timing (used for A/B runs inside one gpurun call)
timeout 300 python bench.py --steps 64 --warmup 10 --no-cpu-baseline --model graphsage_maxpool --no-aux 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('maxpool %.1f us/step' % (d['ms_per_step']*1e3))"
