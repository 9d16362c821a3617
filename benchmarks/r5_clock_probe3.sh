#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
for v in base nt base nt; do
  lib=benchmarks/probes/_lib/libgs_s16_$v.so
  echo "== $v: $(GS_LIB=$lib timeout 100 python -m benchmarks.micro_split pool16 2>/dev/null | tail -1)"
done
GS_LIB=benchmarks/probes/_lib/libgs_s16_nt.so timeout 200 python bench.py --model graphsage_maxpool --steps 40 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('maxpool step with nt stores: %.1f us' % (d['ms_per_step']*1e3))"
timeout 200 python bench.py --model graphsage_maxpool --steps 40 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('maxpool step default: %.1f us' % (d['ms_per_step']*1e3))"
python -m benchmarks.micro_split poolloop16 > $O/poolloop16.txt 2>&1 &
PID=$!
for i in $(seq 1 60); do
  echo "--- sample $i lines=$(wc -l < $O/poolloop16.txt)" >> $O/clk_pool16.txt
  rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|Power \(W\)" >> $O/clk_pool16.txt
  sleep 0.5
  kill -0 $PID 2>/dev/null || break
done
wait $PID
tail -2 $O/poolloop16.txt
grep -A2 "lines=[1-9]" $O/clk_pool16.txt | grep -iE "sclk|power" | sort | uniq -c | sort -rn | head -8
