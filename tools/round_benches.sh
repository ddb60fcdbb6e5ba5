# the round's bench lines, one per workload, into gpurun_out/r05/ (run on the GPU box: gpurun -- bash tools/round_benches.sh)
set -e
# (in the environment before HIP starts -- under rocprofv3 the preloaded library initialises HIP before python runs: bench.py's own setdefault would come too late)
export GPU_MAX_HW_QUEUES=8
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python bench.py > $O/bench_final.json 2> $O/bench_final.err
tail -c 1500 $O/bench_final.json
for w in cfg2_16x2kb_dna_full cfg3_64x500aa_protein_full cfg5_512x10kb_dna_anchored codon_16x1500_full codon_16x1500_anchored fb_cfg2_16x2kb_dna_full fb_cfg4_leafpairs_banded; do
    python bench.py --workload $w --steps 5 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err
    python - $O/bench_$w.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"], d["value"], d["unit"], d["ms_per_step"], "ms")
PY
done
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --dist-backend gloo --share-device > $O/bench_n2_rehearsal.json 2> $O/bench_n2.err
tail -c 600 $O/bench_n2_rehearsal.json
