# rocprofv3 passes of the headline bench command (run on the GPU box: gpurun -- bash tools/profile_bench.sh):
# kernel trace + stats, then WRITE_SIZE and FETCH_SIZE in separate PMC passes (never combined with a trace domain).
set -e
# (in the environment before HIP starts -- under rocprofv3 the preloaded library initialises HIP before python runs: bench.py's own setdefault would come too late)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_stats.json 2> $O/stats.err
python tools/rocpd_summary.py stats $(find $O/stats -name "*_results.db" | head -1) $O/bench_cfg4_kernel_stats.csv
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_w.json 2> $O/w.err
python tools/rocpd_summary.py pmc $(find $O/w -name "*_results.db" | head -1) WRITE_SIZE > $O/pmc_write.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_f.json 2> $O/f.err
python tools/rocpd_summary.py pmc $(find $O/f -name "*_results.db" | head -1) FETCH_SIZE > $O/pmc_fetch.txt
rm -rf $O/stats $O/w $O/f
head -14 $O/bench_cfg4_kernel_stats.csv | cut -c1-150
grep "pg_fill_pipe\|pg_backptr" $O/pmc_write.txt | cut -c1-300
grep "pg_fill_pipe\|pg_backptr" $O/pmc_fetch.txt | cut -c1-300
