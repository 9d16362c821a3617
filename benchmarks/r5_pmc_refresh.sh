#!/bin/bash
# Light refresh of the digest-stamped profiles after a csrc change: relevant GPU tests, K2 / whole-step PMC traffic, MFMA utilisation,
# headline + max-pool kernel traces, then the driver's command (which must report traffic_stale = false afterwards -- the profiles
# are read from the repo copy, so THIS run still sees the old ones: check with a second call after copying).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 16 --warmup 20 --no-cpu-baseline --no-aux"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- $B > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/cal -o cal -- python $R/benchmarks/calibrate_fetch.py > $O/cal.log 2>&1
db() { ls $O/$1/*/*_results.db $O/$1/*_results.db 2>/dev/null | head -1; }
python $R/benchmarks/pmc_traffic.py $(db pmc_fetch) $(db pmc_write) $(db cal) $O/k2_pmc.json > $O/pmc_traffic.log 2>&1
python $R/benchmarks/pmc_step_traffic.py $(db pmc_fetch) $(db pmc_write) $(db cal) $O/step_traffic.json > $O/step_traffic.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o m -- $B > $O/pmc_mfma.log 2>&1
python $R/benchmarks/pmc_mfma.py $(db pmc_mfma) $O/mfma_util.md > /dev/null 2>> $O/pmc_mfma.log
rm -rf $O/trh; timeout 300 rocprofv3 --kernel-trace --stats -d $O/trh -o t -- python $R/bench.py --steps 96 --warmup 20 --no-cpu-baseline --no-aux > $O/headline_bench.json 2> $O/headline_bench.err
python $R/benchmarks/rocpd_stats.py $(db trh) --md $O/r05_bench_kernel_stats.md > /dev/null 2>&1; rm -rf $O/trh
bash $R/benchmarks/trace_aux_all.sh $O r05 2>&1 | tail -6
find $O -name "*.db" -size +20M -delete
python -c "
import json
d=json.load(open('$O/k2_pmc.json')); print('k2 traffic %.1f MB, %.1f us, digest %s' % (d['traffic_bytes_per_launch']/1e6, d['avg_duration_us_under_pmc'], d['lib_digest'][:8]))"
