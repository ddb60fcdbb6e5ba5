# rocprofv3 kernel trace + stats of the tunnels' forward/backward workload, and WRITE_SIZE / FETCH_SIZE of its sweeps in separate
# PMC passes (run on the GPU box: gpurun -- bash tools/profile_fb_band.sh)
set -e
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05_fb
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 bench.py --workload fb_cfg4_leafpairs_banded --steps 3 --warmup 1 --no-cpu-baseline > $O/fb_band_under_stats.json 2> $O/stats.err
python tools/rocpd_summary.py stats $(find $O/stats -name "*_results.db" | head -1) $O/fb_band_kernel_stats.csv
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w -- python3 bench.py --workload fb_cfg4_leafpairs_banded --steps 2 --warmup 1 --no-cpu-baseline > $O/fb_band_under_w.json 2> $O/w.err
python tools/rocpd_summary.py pmc $(find $O/w -name "*_results.db" | head -1) WRITE_SIZE > $O/fb_band_pmc_write.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f -- python3 bench.py --workload fb_cfg4_leafpairs_banded --steps 2 --warmup 1 --no-cpu-baseline > $O/fb_band_under_f.json 2> $O/f.err
python tools/rocpd_summary.py pmc $(find $O/f -name "*_results.db" | head -1) FETCH_SIZE > $O/fb_band_pmc_fetch.txt
rm -rf $O/stats $O/w $O/f
head -8 $O/fb_band_kernel_stats.csv | cut -c1-160
grep "pg_fb" $O/fb_band_pmc_write.txt | cut -c1-300
grep "pg_fb" $O/fb_band_pmc_fetch.txt | cut -c1-300
