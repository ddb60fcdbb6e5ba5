"""CPU, world_size 2 over gloo: the N>1 path of the bench / tree walk.  Units (node alignments)
are dealt to ranks by the work-queue rule with no data-path collective; each rank aligns only its
own units (here with the oracle standing in for the GPU -- there is none in this container), and
the ranks meet only to reduce timing and unit counts."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pagan2_msa_amd import dist as pdist, host, synth
    import oracle
    # the same deterministic set of units on every rank
    jobs = []
    for seed in range(7):
        left = synth.random_graph(40 + 25 * seed, 15, 10 + seed)
        right = synth.random_graph(50 + 20 * seed, 15, 30 + seed)
        jobs.append((left, right, synth.random_model(15, seed), None))
    costs = [(l.n_sites - 1) * (r.n_sites - 1) for l, r, _, _ in jobs]
    mine = pdist.my_units(costs, host.assign_units)
    pdist.barrier()
    scores = {k: oracle.dp_align(*jobs[k]).score for k in mine}
    cells = sum(costs[k] for k in mine)
    tmax, total = pdist.reduce_step(0.25 * (rank + 1), cells)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank),
            np.array([tmax, total, len(mine)] + [float(k) for k in mine] + [scores[k] for k in mine]))
    dist.destroy_process_group()


def test_two_ranks_shard_units_without_a_collective(tmp_path, oracle, pg):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from pagan2_msa_amd import host, synth
    seen, all_scores = [], {}
    for r in range(world):
        a = np.load(tmp_path / ("rank%d.npy" % r))
        tmax, total, n = a[0], a[1], int(a[2])
        ks = [int(x) for x in a[3:3 + n]]
        sc = a[3 + n:3 + 2 * n]
        assert tmax == 0.5                              # max over ranks of 0.25, 0.5
        seen += ks
        all_scores.update(dict(zip(ks, sc)))
        assert n >= 3                                   # 7 units over 2 ranks: 3 or 4 each
    assert sorted(seen) == list(range(7))               # every unit exactly once
    costs = []
    for seed in range(7):
        left = synth.random_graph(40 + 25 * seed, 15, 10 + seed)
        right = synth.random_graph(50 + 20 * seed, 15, 30 + seed)
        costs.append((left.n_sites - 1) * (right.n_sites - 1))
        assert all_scores[seed] == oracle.dp_align(left, right, synth.random_model(15, seed)).score
    assert total == sum(costs)                          # aggregate = all units, whole job
    owner = host.assign_units(costs, 2)
    loads = [sum(c for c, o in zip(costs, owner) if o == w) for w in range(2)]
    assert max(loads) <= 0.62 * sum(costs)              # the rule balances the two queues


def test_assign_units_rule(pg):
    from pagan2_msa_amd import host
    owner = host.assign_units([10, 1, 1, 1, 9, 8], 3)
    assert sorted(owner.tolist()) == [0, 0, 0, 1, 2, 2] or len(set(owner.tolist())) == 3
    assert owner[0] != owner[4] != owner[5] and owner[0] != owner[5]     # the three big units land apart
    assert host.assign_units([5, 4, 3], 1).tolist() == [0, 0, 0]
