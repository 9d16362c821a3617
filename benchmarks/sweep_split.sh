#!/bin/bash
# A/B of the per-step schedule knobs on the bench workload (env overrides; one JSON line each, steps short).
run() { echo -n "$* : "; env "$@" python bench.py --steps 96 --no-cpu-baseline --no-aux ${EXTRA} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f us/step  loss %.4f' % (d['ms_per_step']*1e3, d['config']['loss_after']))"; }
run GS_X=0
run GS_STREAM_SLICE_ROWS=320
run GS_STREAM_SLICE_ROWS=384
run GS_STREAM_SLICE_ROWS=224
run GS_X=0
