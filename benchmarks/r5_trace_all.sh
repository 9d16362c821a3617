O=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trh
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trh -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --no-cpu-baseline --no-aux > $O/r05_headline_bench.json 2> $O/r05_headline_bench.err
python $GRAFT_REPO_ROOT/benchmarks/rocpd_stats.py $(ls $O/trh/*_results.db $O/trh/*/*_results.db 2>/dev/null | head -1) --md $O/r05_headline_kernel_stats.md > /dev/null 2>&1
rm -rf $O/trh
bash $GRAFT_REPO_ROOT/benchmarks/trace_aux_all.sh $O r05
