#!/usr/bin/env python3
"""bench.py -- DP cells/s of the pairwise graph-vs-graph Viterbi aligner on MI355X.

    python bench.py [--gpus 1] --steps K --warmup W [--workload NAME]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One GPU (the default; workload = BASELINE.json configs[3], the headline: 32 x 100 kb synthetic DNA on
a balanced guide tree, branch 0.01, prefix anchors, --anchors-offset 15).  Untimed setup walks the
tree on the GPU (twice; the second walk's wall-clock is e2e_wall_s) and keeps every internal node's
aligner inputs -- child graphs, model table, band -- resident in HBM, one resident batch per guide-tree
level.  A timed "step" is one pass of the hot path (matrix fill + end corner + traceback) over the whole
tree IN DEPENDENCY ORDER: the level batches run one after the other, each waiting for the one before,
exactly as a progressive alignment has to (a parent's inputs come from its children).
value = in-band DP cells of the tree / time of that pass.  The same nodes launched side by side as one
batch (no dependency order; the round-1 headline) are reported as value_resident_batch.

N > 1 GPUs: ONE tree whose ready nodes the N ranks take from a dynamic queue in the job's key-value store
(pagan2_msa_amd.dist.align_sharded: a rank claims ready nodes with an atomic counter per node, runs model + anchors +
DP + parent graph for them on its own GPU, posts the finished paths under the nodes' keys and imports what the others
have posted when it next looks; no collective on the data path, no round barrier).  A step is one whole progressive
alignment; value = cells of the tree / wall-clock of the walk (max over ranks): "strong" scaling.  Without --workload
the line is about configs[3] (cfg4, what --gpus 1 times: one workload from N = 1 to N = 8) and carries configs[4]
(cfg5: 512 x 10 kb, branch 0.02 -- the workload whose units fill more than one GPU) in `also`; `roofline` is the
aggregate one (36 B x cells / wall against N x 8 TB/s, the fill kernels' device time per rank beside it), `cpu_baseline`
the oracle on rank 0's host.  Rank 0 also times the same walk alone on its GPU (value_one_gpu_same_workload) so that a
speed-up can be read off one line.  Without torchrun, `--gpus N` runs the in-process work queue over N devices of this
process instead (one feeder thread per device, pagan_msa_align with n_devices = N).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (config index, leaves, length, branch, sub, indel_start, mean_len, anchors, alphabet)
    "cfg4_32x100kb_dna_anchored": (4, 32, 100000, 0.01, 0.008, 0.0008, 4.0, 1, "ACGT"),
    "cfg2_16x2kb_dna_full": (2, 16, 2000, 0.05, 0.04, 0.004, 4.0, 0, "ACGT"),
    "cfg3_64x500aa_protein_full": (3, 64, 500, 0.05, 0.04, 0.004, 4.0, 0, "ARNDCQEGHILKMFPSTWYV"),
    "cfg5_512x10kb_dna_anchored": (5, 512, 10000, 0.02, 0.016, 0.0016, 4.0, 1, "ACGT"),
    "smoke_8x3kb_dna_anchored": (0, 8, 3000, 0.01, 0.008, 0.0008, 4.0, 1, "ACGT"),
    # DNA read as codons (--codons): the empirical codon model's 1892 states, a 14 MB score table per node that stays in
    # HBM / L2 (the banded kernel's assist waves gather the scores, the tiled kernel stages 64 x 64 of them per tile);
    # length in codons, anchors found in the translation
    "codon_16x1500_anchored": (6, 16, 1500, 0.03, 0.04, 0.004, 3.0, 1, "CODON"),
    "codon_16x1500_full": (6, 16, 1500, 0.03, 0.04, 0.004, 3.0, 0, "CODON"),
    # forward/backward (--full-probability) over the 15 node pairs of cfg2's tree
    "fb_cfg2_16x2kb_dna_full": (2, 16, 2000, 0.05, 0.04, 0.004, 4.0, 0, "ACGT"),
    # forward/backward inside the tunnel: the 16 leaf pairs of cfg4's tree (2 x 100 kb each, the band define_tunnel gives them);
    # narrow diagonals, so the sweeps are pg_fb_forward / pg_fb_backward themselves (one workgroup per pair and direction)
    "fb_cfg4_leafpairs_banded": (4, 32, 100000, 0.01, 0.008, 0.0008, 4.0, 1, "ACGT"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_CELL = 36            # 3 states x (f64 score + u32 back-pointer), SURVEY.md s.8(d)
# Latency floors of one anti-diagonal step of the banded kernel.  The recurrence's true chain per diagonal
# is two dependent fp64 operations (add -> max): ~2 x 8.4 cycles.  THIS FORMULATION's floor is what its class 0 step costs
# a wave that waits for nobody: ~78 instructions (42 vector ones at 5 cycles, 9 LDS operations, two stores, ~25 scalar ones at
# about a cycle each, two branches that are not taken at 7 each), MEASURED on an alignment whose every diagonal lies in one
# wave's rows and holds simple sites only (tools/probe_one_wave.py): 489 cycles per diagonal at 2.4 GHz.  (Round 2 and the
# first half of round 3 quoted 389 cycles from a build with the flag checks removed: in such a build the waves do not wait
# for each other and overlap their intervals -- a number about something else.)
CHAIN_FLOOR_US = 2 * 8.4 / 2400.0
STEP_FLOOR_US = 489 / 2400.0
PMC_PROFILE = os.path.join("profiles", "r05_pmc_fill.json")
PMC_FB_PROFILE = os.path.join("profiles", "r05_pmc_fb_band.json")      # the tunnels' forward/backward workload
KERNELS = ("pg_fill_pipe", "pg_backptr", "pg_fill_tiles_flow", "pg_fill_wavefront")


def make_inputs(workload):
    from pagan2_msa_amd import synth
    cfg, leaves, length, branch, sub, indel, mean_len, anchors, alphabet = WORKLOADS[workload]
    if alphabet == "CODON":
        # evolved over 61 letters, one per sense codon (the model's order: lexical without TAA, TAG, TGA), written out as triplets
        codons = [a + b + c for a in "ACGT" for b in "ACGT" for c in "ACGT" if a + b + c not in ("TAA", "TAG", "TGA")]
        letters = "".join(chr(64 + k) for k in range(61))
        names, seqs, newick = synth.evolve_balanced(leaves, length, branch=branch, sub=sub, indel_start=indel, mean_len=mean_len,
                                                    seed=20240807 + cfg, alphabet=letters)
        return names, ["".join(codons[ord(ch) - 64] for ch in sq) for sq in seqs], newick
    return synth.evolve_balanced(leaves, length, branch=branch, sub=sub, indel_start=indel, mean_len=mean_len,
                                 seed=20240807 + cfg, alphabet=alphabet)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="N > 1 under torchrun: nccl (= RCCL, the default) or gloo (rehearsal on a box with fewer GPUs than ranks)")
    ap.add_argument("--fb-one-by-one", action="store_true",
                    help="forward/backward workloads: one pagan_fb_run per pair from a thread pool instead of one pagan_fb_run_batch")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: rank r uses device r %% device_count instead of device r (numbers are meaningless)")
    ap.add_argument("--allow-stale-traffic", action="store_true", help="(accepted for older scripts: this is the default now)")
    ap.add_argument("--strict-traffic", action="store_true",
                    help="fail instead of reporting roofline.traffic = null when %s does not describe this run "
                         "(for the line that is committed after tools/make_pmc_profile.py; the profiling passes themselves "
                         "run without it: they are what the profile is made from)" % PMC_PROFILE)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    if (args.workload or "").startswith("fb_"):
        # The forward/backward workload keeps 15 pairs x 2 sweeps in flight, a kernel each; the HIP runtime deals a process's
        # streams onto 4 hardware queues unless told otherwise, and kernels of one queue run one after the other
        # (measured: 4 queues 1.39e8 cells/s, 8 queues 2.68e8, 16 the same).  Has to be in the environment before HIP starts.
        # (Under rocprofv3 the profiler's library has initialised HIP before this line runs: tools/*.sh export the variable instead,
        #  and the line below reports what was in the environment when the process started, not what this setdefault asked for.)
        hw_queues_at_start = os.environ.get("GPU_MAX_HW_QUEUES")
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        os.environ["PAGAN_BENCH_HW_QUEUES_AT_START"] = hw_queues_at_start or "unset (8 requested before import torch)"
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the aligner has no CPU path")
    if world == 1 and args.gpus > torch.cuda.device_count():
        raise SystemExit("bench.py: --gpus %d but %d device(s) visible; start one rank per GPU with torch.distributed.run"
                         % (args.gpus, torch.cuda.device_count()))
    if args.share_device:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if args.gpus > 1:
        return bench_work_queue(args, rank, local_rank, world)
    if (args.workload or "").startswith("fb_"):
        return bench_forward_backward(args, local_rank)
    return bench_one_gpu(args, local_rank)


# ---------------------------------------------------------------------------------------------------------------
def bench_forward_backward(args, device):
    """Forward + backward sum-product sweeps (pg_fb_forward / pg_fb_backward, log space) over every node pair of the
    workload's tree: cells/s of the two sweeps, per-kernel HIP-event times, the HBM roofline at 48 B per cell (two matrices
    of 3 fp64 states, each written once), and the oracle's log-space restatement as the CPU baseline."""
    import numpy as np
    import pagan2_msa_amd as pg
    from pagan2_msa_amd import host

    workload = args.workload
    cfg, leaves, length, branch, sub, indel, mean_len, anchors, alphabet = WORKLOADS[workload]
    pg.lib().pagan_dp_select_device(device)
    names, seqs, newick = make_inputs(workload)
    msa = host.Msa(names, seqs, newick, use_anchors=anchors, first_device=device, n_devices=1).align()
    bf = np.array([sum(sq.count(x) for sq in seqs) for x in alphabet], np.float32)
    bf /= bf.sum()
    n_nodes = msa.n_internal
    jobs = []
    leaf_pairs_only = "leafpairs" in workload
    for k in range(n_nodes):
        info = msa.node_info(k)
        if leaf_pairs_only and not (info.left < leaves and info.right < leaves):
            continue
        left, right, _model, band = msa.node_job(k)
        jobs.append((left, right, host.model_prob(1 if len(alphabet) == 4 else 2, msa.node_info(k).dist, base_freq=bf), band))

    def one_pair(job):
        left, right, mp, band = job
        fb = pg.FullProbability(left, right, mp, band, device=device)
        out = (fb.forward_ms, fb.backward_ms, fb.cells, (fb.log_fwd, fb.log_bwd))
        fb.close()
        return out

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(min(len(jobs), 16))      # the node pairs are independent: their sweeps (a workgroup each) run side by side

    def one_pass():
        t0 = time.perf_counter()
        if args.fb_one_by_one:
            rs = list(pool.map(one_pair, jobs))
        else:
            # round 5: every pair of the tree in ONE call -- the forward sweeps of all wide pairs in one launch, the backward sweeps in
            # another (pagan_fb_run_batch); --fb-one-by-one: a call per pair from a thread pool, as before (the HIP runtime's
            # hardware queues then decide how many sweeps run side by side)
            fbs = pg.full_probability_batch(jobs, device=device)
            rs = [(fb.forward_ms, fb.backward_ms, fb.cells, (fb.log_fwd, fb.log_bwd)) for fb in fbs]
            for fb in fbs:
                fb.close()
        wall = time.perf_counter() - t0
        return sum(r[0] for r in rs), sum(r[1] for r in rs), sum(r[2] for r in rs), [r[3] for r in rs], wall, [r[2] for r in rs]

    for _ in range(args.warmup):
        one_pass()
    t0 = time.perf_counter()
    acc = [one_pass() for _ in range(args.steps)]
    elapsed = time.perf_counter() - t0
    fwd_ms = float(np.mean([a[0] for a in acc])); bwd_ms = float(np.mean([a[1] for a in acc]))
    cells, totals = acc[-1][2], acc[-1][3]
    dev_s = float(np.mean([a[4] for a in acc]))       # wall-clock of a pass: every pair's two sweeps side by side (one workgroup each), uploads included
    ok = all(abs(f - b) <= 1e-7 * max(1.0, abs(f)) for f, b in totals)       # the reference's own check (VA:351-355): forward total = backward total
    out = {
        "metric": "DP cells/sec of the forward + backward sweeps (--full-probability), %s node pairs" % workload,
        "value": cells / dev_s, "unit": "cells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dev_s, "higher_is_better": True, "scaling": None, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload, "leaves": leaves, "length": length, "node_pairs": len(jobs), "cells_per_step": int(cells),
                   "hip_hw_queues_requested": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "hip_hw_queues_in_environment_at_start": os.environ.get("PAGAN_BENCH_HW_QUEUES_AT_START"),
                   "call": "one pagan_fb_run per pair from a thread pool" if args.fb_one_by_one else "one pagan_fb_run_batch for all pairs (the wide pairs' forward sweeps in one launch, their backward sweeps in another)",
                   "note": "value = cells / wall-clock of a pass with all node pairs in flight at once (wide pairs: block-scheduled sweeps of one-wave workgroups, each pair its share of the device's workgroup slots); the per-kernel ms are the launches' own durations (one by one: their sums); wall per pass incl. allocation and upload: %.1f ms" % (1e3 * elapsed / args.steps)},
        "roofline": {"bound": "hbm", "achieved": 48 * cells / dev_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": 48 * cells / dev_s / 1e9 / HBM_PEAK_GBS, "kernel": "pg_fb_forward + pg_fb_backward",
                     "algorithmic_bytes_per_cell": 48, "traffic": None,
                     "kernels": [{"kernel": "pg_fb_forward", "ms_per_step": fwd_ms, "achieved": 24 * cells / (fwd_ms * 1e-3) / 1e9},
                                 {"kernel": "pg_fb_backward", "ms_per_step": bwd_ms, "achieved": 24 * cells / (bwd_ms * 1e-3) / 1e9}]},
        "parity_self_check": bool(ok),
    }
    # HBM bytes per pass from the committed PMC passes of this same command (tools/profile_fb_band.sh), when they describe this run
    try:
        prof = json.load(open(os.path.join(ROOT, PMC_FB_PROFILE)))
        if prof.get("workload") == workload and prof.get("cells_per_step") == int(cells):
            out["roofline"].update({"traffic": prof["hbm_bytes_per_step"], "traffic_source": PMC_FB_PROFILE,
                                    "traffic_over_algorithmic": prof["hbm_bytes_per_step"] / (48.0 * cells)})
    except (OSError, ValueError):
        pass
    if not args.no_cpu_baseline:
        import oracle
        oracle.build()
        t0 = time.perf_counter()
        c_cells, used, agree = 0, 0, True
        per_pair = acc[-1][5]
        for (left, right, mp, band), (lf, lb), pair_cells in zip(jobs, totals, per_pair):
            if time.perf_counter() - t0 > args.cpu_seconds:
                break
            olf, olb, _p, _f = oracle.fb(left, right, mp, band=band, matrices=False)
            agree = agree and abs(olf - lf) <= 1e-9 * abs(olf) and abs(olb - lb) <= 1e-9 * abs(olb)
            c_cells += pair_cells            # (the cells of the pair's matrix, or of its tunnel: what the GPU counted for it)
            used += 1
        secs = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": c_cells / secs if secs > 0 else 0.0, "unit": "cells/s", "cores": 1, "kind": "port",
                               "sample": "%d of %d node pairs, log-space restatement of the reference's forward + backward, %.1f s, single thread"
                                         % (used, len(jobs), secs), "matches_gpu": bool(agree)}
        ok = ok and agree
    print(json.dumps(out))
    if not ok:
        raise SystemExit("bench.py: forward/backward self-check FAILED")


# ---------------------------------------------------------------------------------------------------------------
def bench_one_gpu(args, device):
    import ctypes as C
    import numpy as np
    import torch
    import pagan2_msa_amd as pg
    from pagan2_msa_amd import abi, host

    workload = args.workload or "cfg4_32x100kb_dna_anchored"
    cfg, leaves, length, branch, sub, indel, mean_len, anchors, alphabet = WORKLOADS[workload]
    L = pg.lib()
    L.pagan_dp_select_device(device)
    names, seqs, newick = make_inputs(workload)

    # ---- untimed setup: whole progressive alignment on the GPU (twice: the first walk also pays the process's
    # one-off costs -- code object load, first hipMalloc, staging buffers) ----
    data_type = 3 if alphabet == "CODON" else 0               # 0: guessed from the residues, as the reference does
    t0 = time.time()
    msa = host.Msa(names, seqs, newick, use_anchors=anchors, first_device=device, n_devices=1, data_type=data_type)
    msa.align()
    e2e_wall_cold = time.time() - t0
    del msa
    t0 = time.time()
    msa = host.Msa(names, seqs, newick, use_anchors=anchors, first_device=device, n_devices=1, data_type=data_type)
    msa.align()
    e2e_wall = time.time() - t0
    tm = msa.timing()
    n_nodes = msa.n_internal
    infos = [msa.node_info(k) for k in range(n_nodes)]
    levels = sorted({i.level for i in infos})
    opts = abi.COpts(0, device)

    def resident(ks):
        cjobs = (abi.CJob * len(ks))()
        for a, k in enumerate(ks):
            cjobs[a] = msa.node_cjob(k)
        hb = C.c_void_p()
        rc = L.pagan_batch_create(len(ks), cjobs, C.byref(opts), C.byref(hb))
        if rc != 0:
            raise SystemExit("pagan_batch_create failed: %d" % rc)
        return hb

    by_level = [[k for k in range(n_nodes) if infos[k].level == lv] for lv in levels]
    batches = [resident(ks) for ks in by_level]
    cells_level = [int(L.pagan_batch_cells(hb)) for hb in batches]
    cells = sum(cells_level)
    steps_level = [max(infos[k].left_sites + infos[k].right_sites - 3 for k in ks) for ks in by_level]   # diagonals

    def run(hb):
        rc = L.pagan_batch_run(hb)
        if rc == 0:
            rc = L.pagan_batch_sync(hb)          # the next level's inputs depend on this one's results
        if rc != 0:
            raise SystemExit("pagan_batch_run failed: %d" % rc)

    def step():
        for hb in batches:
            run(hb)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    fill_ms = np.zeros((args.steps, len(batches)))
    trace_ms = np.zeros((args.steps, len(batches)))
    kern_ms = np.zeros((args.steps, len(batches), 4))          # per kernel: KERNELS
    ms = (C.c_double * 6)()
    t0 = time.perf_counter()
    for s in range(args.steps):
        for b, hb in enumerate(batches):
            run(hb)
            L.pagan_batch_last_ms_detail(hb, ms)  # HIP events on the library's own streams bracket each kernel
            fill_ms[s, b], trace_ms[s, b] = ms[5], ms[4]
            kern_ms[s, b] = [max(ms[k], 0.0) for k in range(4)]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # parity of the resident batches against what the tree walk produced
    ok = True
    for ks, hb in zip(by_level, batches):
        res = (abi.CResult * len(ks))()
        rc = L.pagan_batch_fetch(hb, res)
        ok = ok and rc == 0 and all(abi.Result(res[a]).same_alignment(msa.node_result(k)) for a, k in enumerate(ks))
        for a in range(len(ks)):
            L.pagan_result_free(C.byref(res[a]))
    # how much of the back-pointer work the banded fill's follower workgroups did in the last timed pass (chunks of 16
    # diagonals written while the fill ran / all chunks of the banded alignments; the rest was pg_backptr's)
    followed = [0, 0]
    for ks, hb in zip(by_level, batches):
        for a in range(len(ks)):
            cnt = (C.c_int32 * 2)()
            if L.pagan_batch_debug_followed(hb, a, cnt) == 0:
                followed[0] += int(cnt[0]); followed[1] += int(cnt[1])
    for hb in batches:
        L.pagan_batch_destroy(hb)

    # the same nodes side by side, no dependency order (what round 1 reported as the headline)
    hb_all = resident(list(range(n_nodes)))
    run(hb_all)
    t1 = time.perf_counter()
    for _ in range(max(2, args.steps // 2)):
        run(hb_all)
    side_by_side = cells * max(2, args.steps // 2) / (time.perf_counter() - t1)
    L.pagan_batch_destroy(hb_all)

    fill_launch_ms = fill_ms.mean(axis=0)                      # per level launch
    fill_step_ms = float(fill_launch_ms.sum())
    crit_steps = int(sum(steps_level))
    us_per_step = 1e3 * fill_step_ms / crit_steps
    # per kernel: its cells (the planner's routing of every node, host only), its launches, its time.  The dominant kernel's
    # figures are the roofline object's top-level fields; every kernel of the workload is listed under roofline.kernels.
    routed, route_of = {}, {}
    strip_cells = 0
    for k in range(n_nodes):
        left, right, model, band = msa.node_job(k)
        full = pg.debug_route(left, right, model, band)[0]
        route_of[k] = full
        route = full.split(" ")[0]
        if full == "pg_fill_pipe (row strips)":
            # a wide job filled as row strips by the banded kernel: timed (and counted) in the wide jobs' slot of the batch --
            # pagan_batch_last_ms_detail brackets the strips' launch and the tiled kernel's together
            route = "pg_fill_tiles_flow"
            strip_cells += int(infos[k].cells)
        routed[route] = routed.get(route, 0) + int(infos[k].cells)
    kmean = kern_ms.mean(axis=0)                               # [level, kernel]
    # A banded-fill DISPATCH of at most 32 alignments carries follower workgroups that write the back-pointers while the fill
    # runs (dp_pipe.hip, pipe_follower; PAGAN_DP_FOLLOW=0 switches them off); dp_abi.hip: launch_fill decides per dispatch --
    # a level's small-table jobs and its large-table jobs are a dispatch each -- and so does the accounting here.
    followed_cells = 0
    if os.environ.get("PAGAN_DP_FOLLOW") != "0":
        for ks in by_level:
            for kind in ("pg_fill_pipe", "pg_fill_pipe (large table)"):
                js = [k for k in ks if route_of[k] == kind]
                if 0 < len(js) <= 32:
                    followed_cells += sum(int(infos[k].cells) for k in js)
    pipe_cells = routed.get("pg_fill_pipe", 0)
    per_kernel = []
    for q, name in enumerate(KERNELS):
        t_ms = float(kmean[:, q].sum())
        launches = int((kmean[:, q] > 0).sum())
        if launches == 0:
            continue
        kcells = cells if name == "pg_backptr" else routed.get(name, 0)
        if name == "pg_backptr":
            # (behind a banded dispatch with follower workgroups it only looks at the chunks' flags)
            kcells = (pipe_cells - followed_cells) + routed.get("pg_fill_tiles_flow", 0)
            if kcells == 0:
                continue
            kbytes = 36                                        # 24 B of scores read (neighbours out of L2) + 12 B written per cell
        elif name == "pg_fill_pipe":
            # scores by the fill's workgroups (24 B) + back-pointers by its follower workgroups where the dispatch had them (12 B)
            kbytes = 24 + 12 * followed_cells / max(pipe_cells, 1)
        else:
            kbytes = BYTES_PER_CELL if name == "pg_fill_wavefront" else 24    # the banded and the tiled fill store scores only since round 3
        ach = kbytes * kcells / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
        per_kernel.append({"kernel": name, "ms_per_step": t_ms, "launches_per_step": launches, "avg_launch_ms": t_ms / launches,
                           "cells": int(kcells), "algorithmic_bytes_per_cell": kbytes, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                           "launch_ms_by_level": [float(x) for x in kmean[:, q]]})
        if name == "pg_fill_tiles_flow" and strip_cells > 0:
            # the wide jobs' slot of a batch: row strips on pg_fill_pipe<true, true> where a job qualifies, tiles otherwise
            per_kernel[-1]["kernel"] = "pg_fill_pipe (row strips) + pg_fill_tiles_flow" if strip_cells < kcells else "pg_fill_pipe (row strips)"
            per_kernel[-1]["row_strip_cells"] = int(strip_cells)
    dom = max(per_kernel, key=lambda e: e["ms_per_step"])
    achieved = dom["achieved"]
    out = {
        "metric": ("DP cells/sec, 32x100 kb DNA progressive align (hot path in dependency order: fill + traceback per tree level)"
                   if workload.startswith("cfg4") else
                   "DP cells/sec, %s (hot path in dependency order; not the headline workload)" % workload),
        "value": cells * args.steps / elapsed,
        "unit": "cells/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": None,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload, "leaves": leaves, "length": length, "branch": branch,
                   "data_type": "codon (1892 states)" if alphabet == "CODON" else "protein" if len(alphabet) == 20 else "dna",
                   "anchors": "prefix, offset 15" if anchors else "none", "node_alignments": n_nodes,
                   "levels": [len(ks) for ks in by_level], "cells_per_step": int(cells),
                   "order": "guide-tree levels one after the other (a parent needs its children)"},
        "value_resident_batch": side_by_side,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "kernel": dom["kernel"], "launches_per_step": dom["launches_per_step"],
                     "avg_launch_ms": dom["avg_launch_ms"], "launch_ms_by_level": dom["launch_ms_by_level"],
                     "fill_ms_by_level": [float(x) for x in fill_launch_ms],
                     "cells_by_level": cells_level,
                     "algorithmic_bytes_per_cell": dom["algorithmic_bytes_per_cell"],
                     "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_cell"] * dom["cells"] / dom["launches_per_step"],
                     "algorithmic_bytes_per_cell_whole_fill": BYTES_PER_CELL,
                     "kernels": per_kernel,
                     "back_pointer_chunks_by_followers": {"written_while_the_fill_ran": followed[0], "of": followed[1]},
                     "latency": {"steps": crit_steps, "us_per_step": us_per_step,
                                 "chain_floor_us_per_step": CHAIN_FLOOR_US, "formulation_floor_us_per_step": STEP_FLOOR_US,
                                 "floor_us_per_step": STEP_FLOOR_US, "frac_of_floor": STEP_FLOOR_US / us_per_step,
                                 "note": "steps = anti-diagonals on the critical path (longest alignment of every level; banded "
                                         "workloads).  chain floor: the recurrence's two dependent fp64 operations per diagonal; "
                                         "formulation floor: this kernel's class 0 step with no wave waiting for another "
                                         "(~78 instructions, 9 of them LDS operations: 489 cycles measured on a one-wave alignment, "
                                         "tools/probe_one_wave.py)"}},
        "kernels_ms": {"fill": fill_step_ms, "end_and_trace": float(trace_ms.mean(axis=0).sum()),
                       "end_and_trace_by_level": [float(x) for x in trace_ms.mean(axis=0)]},
        "e2e_wall_s": e2e_wall,
        "e2e_wall_first_in_process_s": e2e_wall_cold,
        "e2e_cells_per_s": cells / e2e_wall,
        "e2e_breakdown_s": tm,
        "parity_self_check": bool(ok),
    }
    out["roofline"].update(pmc_traffic(workload, cells, dom["kernel"], dom["launches_per_step"], args.strict_traffic))
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(msa, n_nodes, args.cpu_seconds)
        ok = ok and out["cpu_baseline"]["matches_gpu"]
    print(json.dumps(out))
    if not ok:
        raise SystemExit("bench.py: parity self-check FAILED")


# ---------------------------------------------------------------------------------------------------------------
def bench_work_queue(args, rank, local_rank, world):
    """N > 1: one guide tree farmed over N GPUs as a work queue (pagan2-msa_amd/dist.py).  A step is one whole
    progressive alignment; `value` = cells of the tree / wall-clock of the walk (max over ranks).  Without --workload the
    line is about BASELINE.json's headline, cfg4 -- the workload `--gpus 1` times, so that a 1 -> N curve compares one
    workload with itself -- and carries cfg5 (the workload whose units fill more than one GPU) in `also`."""
    import torch
    import torch.distributed as dist
    import pagan2_msa_amd as pg
    from pagan2_msa_amd import dist as pdist, host

    in_process = world == 1                      # no torchrun: one process feeds args.gpus devices
    pg.lib().pagan_dp_select_device(local_rank)
    xdev = "cuda" if args.dist_backend == "nccl" else "cpu"     # where the exchanged bytes live
    if not in_process:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    def one_workload(workload, steps, warmup, with_cpu):
        cfg, leaves, length, branch, sub, indel, mean_len, anchors, alphabet = WORKLOADS[workload]
        names, seqs, newick = make_inputs(workload)  # the same tree on every rank

        def walk(sharded):
            """One progressive alignment.  Returns (wall seconds of the alignment, msa)."""
            if in_process:
                msa = host.Msa(names, seqs, newick, use_anchors=anchors, first_device=0, n_devices=args.gpus if sharded else 1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                msa.align()
                return time.perf_counter() - t0, msa
            msa = host.Msa(names, seqs, newick, use_anchors=anchors, first_device=local_rank, n_devices=1)
            if sharded:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if sharded:
                pdist.align_sharded(msa, host.assign_units, device=xdev)
                torch.cuda.synchronize()
                dist.barrier()
            else:
                msa.align()
            return time.perf_counter() - t0, msa

        # rank 0 alone first: the one-GPU time of the same workload (and the reference alignment for the check below)
        solo_s, solo = None, None
        if rank == 0:
            walk(False)
            solo_s, solo = walk(False)
        for _ in range(warmup):
            walk(True)
        elapsed, msa = 0.0, None
        for _ in range(steps):
            dt, msa = walk(True)
            elapsed += dt
        tm = msa.timing()
        if in_process:
            elapsed_max = elapsed
            fills = [tm["dp_fill_dev_s"]]
            aligned = [sum(1 for k in range(msa.n_internal) if msa.node_device(k) >= 0)]
        else:
            elapsed_max, _ = pdist.reduce_step(elapsed, 0, device=xdev)
            # every rank's device time in the fill kernels and the nodes it aligned itself (last walk)
            mine_n = sum(1 for k in range(msa.n_internal) if msa.node_device(k) >= 0)
            t = torch.zeros(2 * world, dtype=torch.float64, device=xdev)
            t[2 * rank] = tm["dp_fill_dev_s"]; t[2 * rank + 1] = mine_n
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            fills = [float(t[2 * r].item()) for r in range(world)]
            aligned = [int(t[2 * r + 1].item()) for r in range(world)]
        if rank != 0:
            return None, True
        n_nodes = msa.n_internal
        cells = sum(int(msa.node_info(k).cells) for k in range(n_nodes))
        same = msa.alignment() == solo.alignment() and all(
            msa.node_info(k).score == solo.node_info(k).score and msa.node_info(k).cells == solo.node_info(k).cells
            for k in range(n_nodes))
        wall = elapsed_max / steps
        # Aggregate HBM roofline of the walk: the fill's algorithmic bytes (24 B of scores + 12 B of back-pointers per cell,
        # DESIGN.md s.2.1) over the WALL-CLOCK of the walk against N x the chip's peak -- host work and idle devices
        # included, which is what an end-to-end farm is worth; the fill kernels' own device time per rank beside it.
        ach = BYTES_PER_CELL * cells / wall / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS * args.gpus, "unit": "GB/s", "frac": ach / (HBM_PEAK_GBS * args.gpus),
                "traffic": None, "basis": "36 B x cells of the tree / wall-clock of the walk, against %d x %.0f GB/s" % (args.gpus, HBM_PEAK_GBS),
                "fill_device_s_by_rank": fills, "nodes_aligned_by_rank": aligned,
                "fill_frac_of_one_gpu_peak_by_rank": [BYTES_PER_CELL * cells * (a / max(n_nodes, 1)) / f / 1e9 / HBM_PEAK_GBS if f > 0 else None
                                                       for f, a in zip(fills, aligned)]}
        out = {
            "metric": "DP cells/sec, %s, one guide tree farmed over %d GPUs as a work queue (end-to-end walk)" % (workload, args.gpus),
            "value": cells * steps / elapsed_max,
            "unit": "cells/s",
            "n_gpus": args.gpus,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": 1e3 * wall,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "leaves": leaves, "length": length, "branch": branch,
                       "anchors": "prefix, offset 15" if anchors else "none", "node_alignments": n_nodes,
                       "cells_per_step": int(cells),
                       "parallelism": ("in-process ready queue over %d devices (host_tree.cpp)" % args.gpus) if in_process else
                                      ("one rank per GPU; dynamic work queue in the job's key-value store: a rank claims ready nodes (largest "
                                       "first) with an atomic counter, posts finished paths to a log and imports the others' as they appear -- "
                                       "no round, no barrier, no collective on the data path (process group: %s); nodes aligned by rank: %s%s"
                                       % ("RCCL" if args.dist_backend == "nccl" else "gloo", aligned,
                                          "; REHEARSAL: ranks share a device" if args.share_device else "")),
                       "step": "one whole progressive alignment: model + anchors + DP + parent graphs, host work included"},
            "value_one_gpu_same_workload": cells / solo_s,
            "one_gpu_wall_s": solo_s,
            "speedup_vs_one_gpu": solo_s / wall,
            "e2e_breakdown_s_rank0": tm,
            "parity_self_check": bool(same),
            "roofline": roof,
            "cpu_baseline": cpu_baseline(solo, n_nodes, args.cpu_seconds) if with_cpu and not args.no_cpu_baseline else None,
        }
        return out, same

    if args.workload:
        out, same = one_workload(args.workload, args.steps, args.warmup, True)
    else:
        out, same = one_workload("cfg4_32x100kb_dna_anchored", args.steps, args.warmup, True)
        also, same5 = one_workload("cfg5_512x10kb_dna_anchored", max(1, min(args.steps, 2)), min(args.warmup, 1), False)
        same = same and same5
        if rank == 0:
            out["also"] = {k: also[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "config", "value_one_gpu_same_workload",
                                                "one_gpu_wall_s", "speedup_vs_one_gpu", "parity_self_check", "roofline")}
            out["note"] = ("cfg4 (BASELINE.json's headline, what --gpus 1 times) barely shards: its levels hold 16/8/4/2/1 banded alignments, one "
                           "workgroup each, which one MI355X already runs side by side, and a level lasts as long as its slowest alignment -- "
                           "expect a flat DP time at 2/4/8 GPUs (only host work shards).  `also` is cfg5 (512 x 10 kb: 256 ready nodes at the "
                           "first level, wide-band alignments above), the workload whose units fill more than one GPU.  At N = 1 `value` is the "
                           "device-resident pass (BASELINE's metric); here it is the whole walk's wall-clock, whose one-GPU figure is "
                           "`value_one_gpu_same_workload`.")
    if rank == 0:
        print(json.dumps(out))
        if not same:
            if not in_process:
                dist.destroy_process_group()
            raise SystemExit("bench.py: the sharded walk differs from the one-GPU walk")
    if not in_process:
        dist.destroy_process_group()


def fill_kernel(anchors):
    """Name of the kernel the timed fill launches: the banded workloads run the register-wavefront kernel
    (or the older LDS ring kernel behind PAGAN_DP_FILL=ring), full matrices the tiled kernel (one persistent launch
    per batch, tiles in dataflow order; one launch per tile anti-diagonal behind PAGAN_DP_TILES=launches; or the
    one-workgroup HBM wavefront behind PAGAN_DP_WIDE=wavefront)."""
    if not anchors:
        if os.environ.get("PAGAN_DP_WIDE") == "wavefront":
            return "pg_fill_wavefront"
        return "pg_fill_tiles" if os.environ.get("PAGAN_DP_TILES") == "launches" else "pg_fill_tiles_flow"
    return "pg_fill_ring" if os.environ.get("PAGAN_DP_FILL") == "ring" else "pg_fill_pipe"


def pmc_traffic(workload, cells, kernel, launches, strict):
    """HBM bytes per launch of the dominant fill kernel from the committed rocprofv3 --pmc passes of this same command
    (WRITE_SIZE and FETCH_SIZE need separate profiler passes, they cannot be read from inside the bench).  The
    profile names its workload, kernel and cells; a mismatch means it was taken on something else: traffic = null and
    a warning (--strict-traffic: an error; meant for the bench line committed next to a fresh profile -- the profiling
    passes of tools/profile_bench.sh run without it, they are what the profile is made FROM)."""
    path = os.path.join(ROOT, PMC_PROFILE)
    try:
        prof = json.load(open(path))
        why = None
        if prof.get("workload") != workload:
            why = "profile is for workload %r" % prof.get("workload")
        elif prof.get("kernel") != kernel:
            why = "profile is for kernel %r, this run launches %r" % (prof.get("kernel"), kernel)
        elif prof.get("cells_per_step") != int(cells):
            why = "profile covers %r cells per step, this run %d" % (prof.get("cells_per_step"), cells)
    except (OSError, ValueError) as e:
        prof, why = None, "cannot read it: %s" % e
    if why is None:
        return {"traffic": prof["hbm_bytes_per_step"] / launches, "traffic_source": PMC_PROFILE,
                "traffic_over_algorithmic": prof["hbm_bytes_per_step"] / (prof.get("algorithmic_bytes_per_cell", BYTES_PER_CELL) * cells)}
    if strict:
        raise SystemExit("bench.py: %s does not describe this run (%s); re-collect it (DESIGN.md s.4)" % (PMC_PROFILE, why))
    print("bench.py: roofline.traffic = null: %s does not describe this run (%s)" % (PMC_PROFILE, why), file=sys.stderr)
    return {"traffic": None, "traffic_source": "%s not applicable: %s" % (PMC_PROFILE, why)}


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
    out = {"value": cells / secs if secs > 0 else 0.0, "unit": "cells/s", "cores": 1, "kind": "port",
           "sample": "%d of %d node alignments of the same workload, %d cells, %.1f s, single thread" %
                     (used, n_nodes, cells, secs), "matches_gpu": bool(agree)}
    # the same restatement over every core of the host: node alignments of one level are independent,
    # one thread each (ctypes releases the GIL); bounded to about a third of the single-thread budget
    try:
        from concurrent.futures import ThreadPoolExecutor
        n_thr = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))     # the box's CPU share for one GPU
        jobs, est = [], 0.0
        for k in order:
            c = int(msa.node_info(k).cells)
            if est >= budget_s * n_thr / 3.0 and jobs:
                break
            jobs.append(k); est += c / max(out["value"], 1.0)
        inputs = [msa.node_job(k) for k in jobs]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(n_thr) as ex:
            rs = list(ex.map(lambda j: oracle.dp_align(*j).cells, inputs))
        dt = time.perf_counter() - t0
        out["all_cores"] = {"value": sum(rs) / dt, "unit": "cells/s", "cores": n_thr,
                            "sample": "%d node alignments, one thread each over %d threads, %d cells, %.1f s" % (len(jobs), n_thr, sum(rs), dt)}
    except Exception as e:          # the single-thread figure stands on its own
        out["all_cores"] = {"value": None, "error": str(e)}
    return out


if __name__ == "__main__":
    main()
