# FETCH_SIZE of the headline bench with the follower workgroups on and off (PAGAN_DP_FOLLOW=0: pg_backptr writes every back-pointer
# afterwards): are the fill kernel's fetched bytes the followers' reads of the scores, or something the fill does by itself?
# (GPU box: gpurun -- bash tools/pmc_follow_ab.sh; separate PMC passes, no trace domain but the kernel trace)
set -e
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05_follow
mkdir -p $O
cd $R
for mode in 1 0; do
  export PAGAN_DP_FOLLOW=$mode
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/p -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_follow${mode}_$c.json 2> $O/err_follow${mode}_$c.txt
    python tools/rocpd_summary.py pmc $(find $O/p -name "*_results.db" | head -1) $c > $O/pmc_follow${mode}_$c.txt
    rm -rf $O/p
    echo "follow=$mode $c"; grep "pg_fill_pipe\|pg_backptr" $O/pmc_follow${mode}_$c.txt | cut -c1-260
  done
done
