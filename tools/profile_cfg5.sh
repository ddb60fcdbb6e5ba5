# rocprofv3 kernel statistics of the cfg5 pass (512 x 10 kb on one GPU): gpurun_out/prof_r05_cfg5/bench_cfg5_kernel_stats.csv
set -e
# (in the environment before HIP starts -- under rocprofv3 the preloaded library initialises HIP before python runs: bench.py's own setdefault would come too late)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05_cfg5
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 bench.py --workload cfg5_512x10kb_dna_anchored --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_stats.json 2> $O/stats.err
python tools/rocpd_summary.py stats $(find $O/stats -name "*_results.db" | head -1) $O/bench_cfg5_kernel_stats.csv
rm -rf $O/stats
head -30 $O/bench_cfg5_kernel_stats.csv
# + the two counter passes (separate, never with a trace domain): HBM bytes written / fetched by the tiled fill
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w -- python3 bench.py --workload cfg5_512x10kb_dna_anchored --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_w.json 2> $O/w.err
python tools/rocpd_summary.py pmc $(find $O/w -name "*_results.db" | head -1) WRITE_SIZE > $O/pmc_write.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f -- python3 bench.py --workload cfg5_512x10kb_dna_anchored --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_f.json 2> $O/f.err
python tools/rocpd_summary.py pmc $(find $O/f -name "*_results.db" | head -1) FETCH_SIZE > $O/pmc_fetch.txt
rm -rf $O/w $O/f
grep "pg_fill_tiles\|pg_fill_pipe\|pg_backptr" $O/pmc_write.txt | cut -c1-200 | head
