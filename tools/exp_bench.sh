# timing experiments on the headline bench: one line of level times per library named on the command line (GPU box)
# usage: bash tools/exp_bench.sh TAG lib1.so lib2.so ...   (libraries relative to pagan2-msa_amd/; results under gpurun_out/r05/)
TAG=$1; shift
O=gpurun_out/r05
mkdir -p $O
for L in "$@"; do
  PAGAN_DP_LIB=$PWD/pagan2-msa_amd/$L PAGAN_DP_SCORE_CHECK=${SCORE_CHECK:-0} PAGAN_DP_RERUN=0 python bench.py --no-cpu-baseline ${BENCH_ARGS} > $O/exp_${TAG}_$L.json 2> $O/exp_${TAG}_$L.err
  python - $O/exp_${TAG}_$L.json $L <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    lv = d["roofline"].get("launch_ms_by_level") or d["roofline"].get("fill_ms_by_level")
    print(sys.argv[2], "ms_per_step %.1f" % d["ms_per_step"], "levels", [round(x, 1) for x in lv], "parity", d.get("parity_self_check"))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
done
