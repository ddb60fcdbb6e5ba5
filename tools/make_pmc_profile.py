#!/usr/bin/env python3
"""Turns the PMC passes of tools/profile_bench.sh / profile_cfg2.sh (gpurun_out/prof_r03*/pmc_write.txt, pmc_fetch.txt:
per-dispatch sums of WRITE_SIZE / FETCH_SIZE in KB, one line per kernel and grid) into the JSON bench.py reads
(profiles/r03_pmc_fill.json) and the per-kernel traffic summaries kept beside it.

    python tools/make_pmc_profile.py gpurun_out/prof_r03 cfg4_32x100kb_dna_anchored profiles/r03_pmc_fill.json
    python tools/make_pmc_profile.py gpurun_out/prof_r03_cfg2 cfg2_16x2kb_dna_full profiles/r03_pmc_fill_tiles_cfg2.json

The timed launches are told from the untimed tree walk's by their count: bench.py runs (warmup + steps) passes over the
resident level batches, so dispatches of one (kernel, grid) and equal size that occur more often than that belong to a level
batch; the per-launch figure is the median of such a group."""
import json
import re
import statistics
import sys


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"^(\S+?)(?:\.kd)? \[grid (\d+) x wg (\d+)\] \[(.*)\]\s*$", line)
        if not m:
            continue
        name, grid, wg, vals = m.group(1), int(m.group(2)), int(m.group(3)), [float(x) for x in m.group(4).split(",") if x.strip()]
        out.setdefault((name, grid, wg), []).extend(vals)
    return out


def short(name):
    if "pg_fill_pipeILb1ELb1E" in name:          # pg_fill_pipe<true, true>: row strips of wide jobs
        return "pg_fill_pipe (row strips)"
    for k in ("pg_fill_pipe", "pg_backptr", "pg_fill_tiles_flow", "pg_fill_tiles", "pg_fill_wavefront", "pg_trace_spec", "pg_trace_compose",
              "pg_trace_emit", "pg_end_corner"):
        if k in name:
            return k
    return name


def main():
    d, workload, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    bench = json.loads(open(d + "/bench_under_w.json").read().strip().splitlines()[-1])
    cells = bench["config"]["cells_per_step"]
    passes = bench["steps"] + bench["warmup"]
    w, f = parse(d + "/pmc_write.txt"), parse(d + "/pmc_fetch.txt")
    kernels = {}
    for (name, grid, wg), vals in w.items():
        k = short(name)
        if not k.startswith("pg_fill") and k != "pg_backptr":        # (the strips' name starts with pg_fill too)
            continue
        fv = f.get((name, grid, wg), [])
        # Two levels can share a grid (the tiled kernel caps it): group a grid's dispatches by size (1 %); a group with more
        # than `passes` members is one level batch (its passes plus the tree walk's own launches of it) -- the side-by-side
        # batch of all nodes is launched fewer times and drops out.
        groups = []
        for i, v in enumerate(vals):
            g = next((g for g in groups if abs(g["w"][0] - v) <= 0.01 * g["w"][0]), None)
            if g is None:
                g = {"w": [], "f": []}
                groups.append(g)
            g["w"].append(v)
            if i < len(fv):
                g["f"].append(fv[i])
        for g in groups:
            if len(g["w"]) < passes + 1:
                continue
            e = kernels.setdefault(k, {"grids": [], "WRITE_SIZE_kb_per_launch": [], "FETCH_SIZE_kb_per_launch": []})
            e["grids"].append(grid)
            e["WRITE_SIZE_kb_per_launch"].append(statistics.median(g["w"]))
            e["FETCH_SIZE_kb_per_launch"].append(statistics.median(g["f"]) if g["f"] else None)
    per_kernel = []
    for k, e in kernels.items():
        wr = 1024.0 * sum(e["WRITE_SIZE_kb_per_launch"])
        fe = 1024.0 * sum(x for x in e["FETCH_SIZE_kb_per_launch"] if x is not None)
        kc = next((q["cells"] for q in bench["roofline"].get("kernels", []) if q["kernel"] == k), None)
        if kc is None:
            # the wide jobs' slot of a batch: row strips where a job qualifies, tiles otherwise (bench.py names the slot by both)
            slot = next((q for q in bench["roofline"].get("kernels", []) if k in q["kernel"].split(" + ")), None)
            if slot is not None and "row_strip_cells" in slot:
                kc = slot["row_strip_cells"] if k == "pg_fill_pipe (row strips)" else slot["cells"] - slot["row_strip_cells"]
            else:
                kc = cells
        kc = max(int(kc), 1)
        per_kernel.append(dict(kernel=k, launches_per_step=len(e["grids"]), cells=kc, write_bytes_per_step=wr, fetch_bytes_per_step_raw=fe,
                               write_bytes_per_cell=wr / kc, fetch_bytes_per_cell=fe / kc, **e))
    dom = bench["roofline"]["kernel"]
    parts = [p for p in per_kernel if p["kernel"] in dom.split(" + ")]
    main_k = {"launches_per_step": sum(p["launches_per_step"] for p in parts),
              "write_bytes_per_step": sum(p["write_bytes_per_step"] for p in parts),
              "fetch_bytes_per_step_raw": sum(p["fetch_bytes_per_step_raw"] for p in parts)}
    kcd = max(sum(p["cells"] for p in parts), 1)
    main_k["write_bytes_per_cell"] = main_k["write_bytes_per_step"] / kcd
    main_k["fetch_bytes_per_cell"] = main_k["fetch_bytes_per_step_raw"] / kcd
    out = {
        "command": "rocprofv3 --kernel-trace --pmc WRITE_SIZE (and, in a separate pass, FETCH_SIZE) -- python3 bench.py --workload %s --steps 2 "
                   "--warmup 1 --no-cpu-baseline (tools/profile_bench.sh / profile_cfg2.sh); per-dispatch sums by tools/rocpd_summary.py pmc, "
                   "this file by tools/make_pmc_profile.py" % workload,
        "kernel": dom, "workload": workload, "cells_per_step": cells, "launches_per_step": main_k["launches_per_step"],
        "algorithmic_bytes_per_cell": next(q["algorithmic_bytes_per_cell"] for q in bench["roofline"]["kernels"] if q["kernel"] == dom),
        "hbm_bytes_per_step": main_k["write_bytes_per_step"] + main_k["fetch_bytes_per_step_raw"],
        "write_bytes_per_cell": main_k["write_bytes_per_cell"], "fetch_bytes_per_cell": main_k["fetch_bytes_per_cell"],
        "kernels": per_kernel,
        "note": "FETCH_SIZE left uncorrected (the x2 gfx950 correction of MI355X_MICROARCH.md is calibrated on 16 B/lane streaming "
                "reads; these kernels read scalars, 4-24 B records and cells): fetch_bytes_* are raw counter sums.  Round 3: the "
                "fill kernels store 24 B of scores per cell; the 12 B of back-pointers are written by the banded fill's follower "
                "workgroups while it runs (they read the scores back out of L2), by pg_backptr afterwards for the tiled fill "
                "and for what the followers left.",
    }
    json.dump(out, open(dst, "w"), indent=1)
    for p in per_kernel:
        print("%-20s launches %d  written %.2f B/cell  fetched %.2f B/cell" % (p["kernel"], p["launches_per_step"], p["write_bytes_per_cell"], p["fetch_bytes_per_cell"]))


if __name__ == "__main__":
    main()
