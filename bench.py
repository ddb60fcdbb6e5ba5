#!/usr/bin/env python3
"""bench.py -- DP cells/s of the pairwise graph-vs-graph Viterbi aligner on MI355X.

Workload (BASELINE.json configs[3], the headline): 32 x 100 kb synthetic DNA on a balanced
guide tree (branch 0.01), prefix anchors with --anchors-offset 15.  Untimed setup runs the
whole progressive alignment on the GPU (twice; the wall-clock of the second walk is reported as
e2e_wall_s, of the process's first as e2e_wall_first_in_process_s) and
keeps every internal node's aligner inputs -- child graphs, model table, band -- resident in
HBM.  A timed "step" is one pass of the hot path (matrix fill + end corner + traceback) over
that batch of 31 node alignments.  value = in-band DP cells per second over all ranks.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

With N > 1 every rank runs the same-sized workload (its own seed) on its own GPU: the path
shards by independent node alignments, there is no collective in the data path ("weak").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (leaves, length, branch, sub, indel_start, mean_len, anchors)
    "cfg4_32x100kb_dna_anchored": (32, 100000, 0.01, 0.008, 0.0008, 4.0, 1),
    "cfg2_16x2kb_dna_full": (16, 2000, 0.05, 0.04, 0.004, 4.0, 0),
    "smoke_8x3kb_dna_anchored": (8, 3000, 0.01, 0.008, 0.0008, 4.0, 1),
    # BASELINE.json configs[4] on one GPU: 511 node alignments, up to 256 of them side by side
    "cfg5_512x10kb_dna_anchored": (512, 10000, 0.01, 0.008, 0.0008, 4.0, 1),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_CELL = 36            # 3 states x (f64 score + u32 back-pointer), SURVEY.md s.8(d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg4_32x100kb_dna_anchored", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the aligner has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import numpy as np
    import pagan2_msa_amd as pg
    from pagan2_msa_amd import abi, host, synth
    import ctypes as C

    pg.lib().pagan_dp_select_device(local_rank)
    leaves, length, branch, sub, indel, mean_len, anchors = WORKLOADS[args.workload]
    seed = 20240807 + 4 + 1000 * rank
    names, seqs, newick = synth.evolve_balanced(leaves, length, branch=branch, sub=sub, indel_start=indel,
                                                mean_len=mean_len, seed=seed)

    # ---- untimed setup: whole progressive alignment on the GPU, inputs stay resident ----
    # (twice: the first walk also pays the process's one-off costs -- code object load, first hipMalloc,
    # staging buffers; the second is what a tree costs in a running process)
    t0 = time.time()
    msa = host.Msa(names, seqs, newick, use_anchors=anchors, first_device=local_rank, n_devices=1)
    msa.align()
    e2e_wall_cold = time.time() - t0
    del msa
    t0 = time.time()
    msa = host.Msa(names, seqs, newick, use_anchors=anchors, first_device=local_rank, n_devices=1)
    msa.align()
    e2e_wall = time.time() - t0
    tm = msa.timing()
    n_nodes = msa.n_internal
    cjobs = (abi.CJob * n_nodes)()
    for k in range(n_nodes):
        cjobs[k] = msa.node_cjob(k)
    opts = abi.COpts(0, local_rank)
    L = pg.lib()
    hb = C.c_void_p()
    rc = L.pagan_batch_create(n_nodes, cjobs, C.byref(opts), C.byref(hb))
    if rc != 0:
        raise SystemExit("pagan_batch_create failed: %d" % rc)
    cells = L.pagan_batch_cells(hb)

    def step():
        rc = L.pagan_batch_run(hb)
        if rc != 0:
            raise SystemExit("pagan_batch_run failed: %d" % rc)

    def fence():
        L.pagan_batch_sync(hb)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    fill_ms, trace_ms = [], []
    ms = (C.c_double * 2)()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP events on the library's own stream bracket each kernel of the step
        L.pagan_batch_last_ms(hb, ms)
        fill_ms.append(ms[0])
        trace_ms.append(ms[1])
    fence()
    elapsed = time.perf_counter() - t0
    from pagan2_msa_amd import dist as pdist
    elapsed_max, total_cells = pdist.reduce_step(elapsed, cells, device="cuda")

    # parity spot check of the resident batch against what the tree walk produced
    res = (abi.CResult * n_nodes)()
    rc = L.pagan_batch_fetch(hb, res)
    ok = rc == 0 and all(abi.Result(res[k]).same_alignment(msa.node_result(k)) for k in range(n_nodes))
    for k in range(n_nodes):
        L.pagan_result_free(C.byref(res[k]))

    if rank == 0:
        fill_avg_ms = float(np.mean(fill_ms))
        achieved = BYTES_PER_CELL * cells / (fill_avg_ms * 1e-3) / 1e9
        out = {
            "metric": ("DP cells/sec, 32x100 kb DNA progressive align (hot path: fill + traceback)"
                       if args.workload.startswith("cfg4") else
                       "DP cells/sec, %s (hot path: fill + traceback; not the headline workload)" % args.workload),
            "value": total_cells * args.steps / elapsed_max,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "leaves": leaves, "length": length, "branch": branch,
                       "anchors": "prefix, offset 15" if anchors else "none", "node_alignments": n_nodes,
                       "cells_per_step_per_gpu": int(cells), "parallelism": "independent node alignments per GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.workload, cells),
                         "kernel": fill_kernel(anchors), "avg_launch_ms": fill_avg_ms,
                         "algorithmic_bytes_per_cell": BYTES_PER_CELL,
                         "algorithmic_bytes_per_launch": BYTES_PER_CELL * int(cells)},
            "kernels_ms": {"fill": fill_avg_ms, "end_and_trace": float(np.mean(trace_ms))},
            "e2e_wall_s": e2e_wall,
            "e2e_wall_first_in_process_s": e2e_wall_cold,
            "e2e_breakdown_s": tm,
            "parity_self_check": bool(ok),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(msa, n_nodes, args.cpu_seconds)
        print(json.dumps(out))
    L.pagan_batch_destroy(hb)
    if world > 1:
        dist.destroy_process_group()


def fill_kernel(anchors):
    """Name of the kernel the timed fill launches: the banded workloads run the register-wavefront kernel
    (or the older LDS ring kernel behind PAGAN_DP_FILL=ring), full matrices the tiled kernel (one launch per
    tile anti-diagonal; or the one-workgroup HBM wavefront behind PAGAN_DP_WIDE=wavefront)."""
    if not anchors:
        return "pg_fill_wavefront" if os.environ.get("PAGAN_DP_WIDE") == "wavefront" else "pg_fill_tiles"
    return "pg_fill_ring" if os.environ.get("PAGAN_DP_FILL") == "ring" else "pg_fill_pipe"


def pmc_traffic(workload, cells):
    """HBM bytes per launch of the fill kernel from the committed rocprofv3 --pmc passes of this same
    command (WRITE_SIZE and FETCH_SIZE need separate passes and cannot be collected from inside
    the bench); None when the profile is for another workload, kernel or cell count."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_fill.json")
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return None
    if (prof.get("workload") != workload or prof.get("cells_per_launch") != int(cells) or
            prof.get("kernel") != fill_kernel(WORKLOADS[workload][6])):
        return None
    return prof["hbm_bytes_per_launch"]


def cpu_baseline(msa, n_nodes, budget_s):
    """The oracle (single-thread CPU restatement of the reference's fill + traceback) timed on
    a bounded sample of the same node alignments; checker role only, never the product path."""
    import oracle
    oracle.build()
    cells, secs, used, agree = 0, 0.0, 0, True
    order = sorted(range(n_nodes), key=lambda k: msa.node_info(k).level)
    for k in order:
        if secs >= budget_s:
            break
        left, right, model, band = msa.node_job(k)
        t0 = time.perf_counter()
        r = oracle.dp_align(left, right, model, band)
        secs += time.perf_counter() - t0
        cells += r.cells
        used += 1
        agree = agree and r.same_alignment(msa.node_result(k))
    return {"value": cells / secs if secs > 0 else 0.0, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "%d of %d node alignments of the same workload, %d cells, %.1f s, single thread" %
                      (used, n_nodes, cells, secs), "matches_gpu": bool(agree)}


if __name__ == "__main__":
    main()
