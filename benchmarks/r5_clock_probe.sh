#!/bin/bash
# Engine clock / power while (a) the pooling MLP and (b) the headline training step run in a loop: rocm-smi polled beside them.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
poll() {  # poll <outfile> <n>
  for i in $(seq 1 $2); do
    echo "--- sample $i $(date +%s.%N)" >> $1
    rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -iE "sclk|mclk|fclk|power|busy|GPU use" >> $1
    sleep 0.4
  done
}
echo "idle:" > $O/clk_idle.txt; poll $O/clk_idle.txt 2
python -m benchmarks.micro_split poolloop > $O/poolloop.txt 2>&1 &
PID=$!
sleep 14    # import + table generation
poll $O/clk_pool.txt 8
wait $PID
python bench.py --steps 30000 --warmup 5 --no-cpu-baseline --no-aux > $O/long_bench.json 2> $O/long_bench.err &
PID=$!
sleep 16
poll $O/clk_step.txt 12
wait $PID
tail -4 $O/poolloop.txt
grep -iE "sclk|power" $O/clk_idle.txt | head -4
echo "== pool"; grep -iE "sclk|power" $O/clk_pool.txt | head -24
echo "== step"; grep -iE "sclk|power" $O/clk_step.txt | head -36
python -c "
import json; d=json.load(open('$O/long_bench.json')); print('long bench: wall %.2f events %.2f' % (d['ms_per_step']*1e3, d['ms_per_step_events']['ms_per_step_median']*1e3))"
