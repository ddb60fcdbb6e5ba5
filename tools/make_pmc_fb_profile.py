"""profiles/r05_pmc_fb_band.json from the two PMC passes of tools/profile_fb_band.sh (per-dispatch sums of WRITE_SIZE / FETCH_SIZE
in KB, one line per kernel and grid): HBM bytes of the forward + backward sweeps per pass of bench.py --workload
fb_cfg4_leafpairs_banded.  Usage: make_pmc_fb_profile.py <dir with fb_band_pmc_write.txt / fb_band_pmc_fetch.txt> <cells per step> <out.json>"""
import json, re, sys
src, cells, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def per_pass(path):
    total, kernels = 0.0, {}
    for ln in open(path):
        m = re.match(r"(\S*pg_fb_\w+?_(?:ring|tiled)\S*|\S*pg_fb_(?:forward|backward)E\S*) \[grid ([^\]]+)\] \[([^\]]+)\]", ln)
        if not m:
            continue
        vals = [float(v) for v in m.group(3).split(",")]
        kb = sum(vals) / len(vals)                       # one dispatch per pass and grid
        key = ("forward" if "forward" in m.group(1) else "backward") + " [grid " + m.group(2) + "]"
        kernels[key] = 1024.0 * kb
        total += 1024.0 * kb
    return total, kernels


w, wk = per_pass(src + "/fb_band_pmc_write.txt")
f, fk = per_pass(src + "/fb_band_pmc_fetch.txt")
json.dump({
    "command": "rocprofv3 --kernel-trace --pmc WRITE_SIZE (and, in a separate pass, FETCH_SIZE) -- python3 bench.py --workload fb_cfg4_leafpairs_banded --steps 2 --warmup 1 --no-cpu-baseline (tools/profile_fb_band.sh); per-dispatch sums by tools/rocpd_summary.py pmc, this file by tools/make_pmc_fb_profile.py",
    "workload": "fb_cfg4_leafpairs_banded", "kernel": "pg_fb_forward_ring + pg_fb_backward_ring", "cells_per_step": cells,
    "algorithmic_bytes_per_cell": 48.0, "hbm_bytes_per_step": w + f,
    "write_bytes_per_cell": w / cells, "fetch_bytes_per_cell": f / cells,
    "write_bytes_per_step_by_launch": wk, "fetch_bytes_per_step_by_launch_raw": fk,
    "note": "FETCH_SIZE left uncorrected (raw counter), as in r05_pmc_fill.json",
}, open(out, "w"), indent=1)
print(open(out).read()[:900])
