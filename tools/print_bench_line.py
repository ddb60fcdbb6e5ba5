"""One line per bench.py JSON file: workload, value, ms per step, self-checks."""
import json, sys
for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    cb = d.get("cpu_baseline") or {}
    print(d["config"]["workload"], "%.4g" % d["value"], d["unit"], "%.1f ms" % d["ms_per_step"], "self-check", d.get("parity_self_check"), "oracle", cb.get("matches_gpu"),
          "kernels", [round(k["ms_per_step"], 1) for k in d.get("roofline", {}).get("kernels", []) if "ms_per_step" in k][:4])
