#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
python -m benchmarks.micro_split poolloop > $O/poolloop.txt 2>&1 &
PID=$!
for i in $(seq 1 40); do
  echo "--- sample $i lines=$(wc -l < $O/poolloop.txt)" >> $O/clk_pool.txt
  rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|Power \(W\)" >> $O/clk_pool.txt
  sleep 0.5
  kill -0 $PID 2>/dev/null || break
done
wait $PID
tail -2 $O/poolloop.txt
grep -A2 "lines=[1-9]" $O/clk_pool.txt | grep -iE "sclk|power" | sort | uniq -c | sort -rn | head -12
