# rocprofv3 kernel statistics of the cfg2 pass (16 x 2 kb, full matrices, tiled kernel): gpurun_out/prof_r02_cfg2/bench_cfg2_tiles_kernel_stats.csv
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02_cfg2
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 bench.py --workload cfg2_16x2kb_dna_full --steps 3 --warmup 1 --no-cpu-baseline --allow-stale-traffic > $O/bench_under_stats.json 2> $O/stats.err
python tools/rocpd_summary.py stats $(find $O/stats -name "*_results.db" | head -1) $O/bench_cfg2_tiles_kernel_stats.csv
rm -rf $O/stats
head -8 $O/bench_cfg2_tiles_kernel_stats.csv
