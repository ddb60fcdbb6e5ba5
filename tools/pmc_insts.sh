# Instruction counts of the banded fill on one root alignment (run on the GPU box: gpurun -- bash tools/pmc_insts.sh [lib suffix ...]):
# rocprofv3 PMC pass (SQ_INSTS_VALU / SALU / LDS / SMEM, SQ_WAVES) over tests/diagnostics/probe_pair.py for the product library and
# for every timing variant named (tools/build_exp.sh) -- e.g. "_exp__idle" (assist waves return at once): the difference is
# what the assist waves execute.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_insts
mkdir -p $O
cd $R
export PG_ALIGN_RING=1
for x in "" "$@"; do
    export PAGAN_DP_LIB=$R/pagan2-msa_amd/libpagan_dp$x.so
    rm -rf $O/p
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d $O/p -o p -- python3 tests/diagnostics/probe_pair.py ${LEAVES:-32} > $O/run$x.log 2> $O/run$x.err
    for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM; do
        echo "lib$x $c $(python tools/rocpd_summary.py pmc $(find $O/p -name '*_results.db' | head -1) $c | grep 'pg_fill_pipe' | cut -c1-200)"
    done
    grep "^node" $O/run$x.log | tail -1
done
rm -rf $O/p
