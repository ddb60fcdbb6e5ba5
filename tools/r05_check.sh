# round 5: parity of the banded kernel after a change, then the headline bench (GPU box: gpurun -- bash tools/r05_check.sh TAG [sweep cases])
TAG=${1:-x}
N=${2:-200}
O=gpurun_out/r05
mkdir -p $O
python -m pytest tests/test_pipe_gpu.py tests/test_dp_parity_gpu.py tests/test_strips_gpu.py -x -q > $O/pipe_tests_$TAG.log 2>&1
rc=$?
tail -3 $O/pipe_tests_$TAG.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tests/diagnostics/sweep_parity.py $N > $O/sweep_$TAG.log 2>&1
rc=$?
tail -2 $O/sweep_$TAG.log
[ $rc -ne 0 ] && exit $rc
grep -q "mismatches: 0" $O/sweep_$TAG.log || exit 9
python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
rc=$?
python - $O/bench_$TAG.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "levels", [round(x, 1) for x in d["roofline"]["launch_ms_by_level"]], "parity", d["parity_self_check"], d["cpu_baseline"]["matches_gpu"], "e2e", d["e2e_wall_s"])
PY
exit $rc
