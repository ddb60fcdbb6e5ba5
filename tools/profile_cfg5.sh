# rocprofv3 kernel statistics of the cfg5 pass (512 x 10 kb on one GPU): gpurun_out/prof_r02_cfg5/bench_cfg5_kernel_stats.csv
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02_cfg5
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 bench.py --workload cfg5_512x10kb_dna_anchored --steps 2 --warmup 1 --no-cpu-baseline --allow-stale-traffic > $O/bench_under_stats.json 2> $O/stats.err
python tools/rocpd_summary.py stats $(find $O/stats -name "*_results.db" | head -1) $O/bench_cfg5_kernel_stats.csv
rm -rf $O/stats
head -30 $O/bench_cfg5_kernel_stats.csv
