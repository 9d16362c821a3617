#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 96 --no-cpu-baseline --no-aux 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f us/step  loss %.4f' % (d['ms_per_step']*1e3, d['config']['loss_after']))"; }
run GS_STREAM_GEMM=0 GS_COGATHER_SPLIT=0.7
run GS_STREAM_GEMM=1 GS_COGATHER_SPLIT=0.7
run GS_STREAM_GEMM=1 GS_COGATHER_SPLIT=0.6
run GS_STREAM_GEMM=1 GS_COGATHER_SPLIT=0.5
run GS_STREAM_GEMM=1 GS_COGATHER_SPLIT=0.5 GS_STREAM_SLICE_ROWS=224
