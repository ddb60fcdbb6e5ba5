"""CPU tests of the multi-device / multi-rank work queue of the tree walk (host_tree.cpp,
pagan2-msa_amd/dist.py).  There is no GPU here, so the queue's batches go to a stand-in for
pagan_dp_align_batch installed through the test seam pagan_msa_set_batch_backend: the oracle's DP,
called per job.  Everything else is the product's code: dealing by cost, feeder threads per device,
the dynamic ready queue, the tunnel retry, parent building, result export/import, the all-gather."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from pagan2_msa_amd import host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_backend(oracle, log):
    """fn(n, jobs, opts, out, user) with pagan_dp_align_batch's contract, backed by oracle_dp_align."""
    L = oracle.lib()

    def fn(n, jobs, opts, out, user):
        log.append((int(opts.contents.device), int(n)))
        for k in range(n):
            j = jobs[k]
            rc = L.oracle_dp_align(j.left, j.right, j.model, j.band if j.band else None, opts, C.byref(out[k]))
            if rc != 0:
                return rc
        return 0
    return fn


def walk(oracle, names, seqs, nwk, log=None, **opts):
    msa = host.Msa(names, seqs, nwk, **opts)
    msa.set_batch_backend(oracle_backend(oracle, log if log is not None else []))
    return msa


def snapshot(msa):
    out = {"rows": msa.alignment(), "scores": [], "cols": []}
    for k in range(msa.n_internal):
        r = msa.node_result(k)
        out["scores"].append(r.score)
        out["cols"].append(r.cols.copy())
    return out


def same(a, b):
    assert a["rows"] == b["rows"]
    assert a["scores"] == b["scores"]
    for x, y in zip(a["cols"], b["cols"]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("shape", ["balanced", "caterpillar"])
def test_in_process_queue_over_devices_matches_one_device(oracle, pg, shape):
    if shape == "balanced":
        names, seqs, nwk = synth.evolve_balanced(16, 220, branch=0.03, sub=0.03, indel_start=0.01, mean_len=4, seed=11)
    else:
        names, seqs, nwk = synth.evolve_caterpillar(9, 180, seed=4)
    base = walk(oracle, names, seqs, nwk, use_anchors=0, n_devices=1, first_device=0).align()
    want = snapshot(base)
    for ndev in (2, 3):
        log = []
        msa = walk(oracle, names, seqs, nwk, log=log, use_anchors=0, n_devices=ndev, first_device=0).align()
        same(snapshot(msa), want)
        devs = {msa.node_device(k) for k in range(msa.n_internal)}
        assert devs <= set(range(ndev))
        if shape == "balanced":
            assert devs == set(range(ndev))                 # a 8-wide first level reaches every device
            assert sum(n for _, n in log) == msa.n_internal
        for r, s in zip(msa.alignment(), seqs):
            assert r.replace("-", "") == s


def test_queue_deals_largest_first_and_retries_failed_tunnels(oracle, pg):
    names, seqs, nwk = synth.evolve_balanced(8, 600, branch=0.02, sub=0.02, indel_start=0.004, mean_len=4, seed=12)
    log = []
    msa = walk(oracle, names, seqs, nwk, log=log, use_anchors=1, n_devices=2, first_device=3).align()
    assert {d for d, _ in log} == {3, 4}                    # first_device is honoured
    one = walk(oracle, names, seqs, nwk, use_anchors=1, n_devices=1, first_device=0).align()
    same(snapshot(msa), snapshot(one))
    # every node's recorded job reproduces its result (bands included)
    for k in range(msa.n_internal):
        left, right, model, band = msa.node_job(k)
        assert oracle.dp_align(left, right, model, band).same_alignment(msa.node_result(k))


def test_export_import_rebuilds_the_same_parents(oracle, pg):
    names, seqs, nwk = synth.evolve_balanced(8, 300, branch=0.04, sub=0.04, indel_start=0.012, mean_len=5, seed=13)
    a = walk(oracle, names, seqs, nwk, use_anchors=0).align()
    b = host.Msa(names, seqs, nwk, use_anchors=0)            # never aligns anything itself
    n = len(names)
    total = 0
    while b.remaining > 0:
        ready = b.ready()
        assert ready
        for node in ready:
            buf = a.export_result(node)
            total += buf.shape[0]
            b.import_result(buf)
    b.finish()
    assert b.alignment() == a.alignment()
    cols = sum(a.node_result(k).cols.shape[0] for k in range(a.n_internal))
    assert total < 64 * a.n_internal + cols + 4 * sum(
        a.node_result(k).left_used.shape[0] + a.node_result(k).right_used.shape[0] for k in range(a.n_internal)) + 8
    for node in range(n, 2 * n - 1):
        ga, gb = a.node_graph(node), b.node_graph(node)
        fa, fb = ga.flatten(), gb.flatten()
        for f in ("state", "bwd_off", "bwd_src", "bwd_eid"):
            assert np.array_equal(getattr(fa, f), getattr(fb, f))
        assert fa.bwd_logw.tobytes() == fb.bwd_logw.tobytes()
        for x, y in zip(ga.attrs(), gb.attrs()):
            assert x.tobytes() == y.tobytes()
    with pytest.raises(Exception):
        b.import_result(a.export_result(n))                  # already done: refused
    with pytest.raises(Exception):
        host.Msa(names, seqs, nwk, use_anchors=0).import_result(a.export_result(2 * n - 2))   # root before its children


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, out_dir, exchange, slow_rank):
    sys.path.insert(0, ROOT)
    import time
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pagan2_msa_amd import dist as pdist, host as h, synth as sy
    import oracle
    oracle.build()
    names, seqs, nwk = sy.evolve_balanced(16, 260, branch=0.02, sub=0.02, indel_start=0.006, mean_len=4, seed=14)
    log = []
    msa = h.Msa(names, seqs, nwk, use_anchors=1, prefix_hit_length=20)
    inner = oracle_backend(oracle, log)

    def backend(n, jobs, opts, out, user):
        if rank == slow_rank:
            time.sleep(3.0)                     # a rank whose device is busy with something else
        return inner(n, jobs, opts, out, user)
    msa.set_batch_backend(backend)
    stamps = []                                 # (wall-clock when the nodes were done and posted, nodes) per claim
    rounds = pdist.align_sharded(msa, h.assign_units, exchange=exchange, walk=7, quota=1 if slow_rank >= 0 else None,
                                 on_round=lambda k, ready, mine: stamps.append((time.time(), list(mine))))
    mine = [k for k in range(msa.n_internal) if msa.node_device(k) >= 0]
    # parent graphs this rank has built when the walk is over -- the nodes it aligned and the imported ones below the nodes it
    # claimed -- and after it has been asked for the rows (which need every node's graph)
    built_walk = msa.parents_built
    np.save(os.path.join(out_dir, "rows%d.npy" % rank), np.array(msa.alignment()))
    np.save(os.path.join(out_dir, "built%d.npy" % rank), np.array([built_walk, msa.parents_built]))
    np.save(os.path.join(out_dir, "scores%d.npy" % rank), np.array([msa.node_info(k).score for k in range(msa.n_internal)]))
    np.save(os.path.join(out_dir, "mine%d.npy" % rank), np.array(mine))
    np.save(os.path.join(out_dir, "rounds%d.npy" % rank), np.array(rounds))
    np.save(os.path.join(out_dir, "stamps%d.npy" % rank), np.array([(t, n) for t, ns in stamps for n in ns], np.float64).reshape(-1, 2))
    dist.barrier()
    dist.destroy_process_group()


def _one_rank_walk(oracle):
    names, seqs, nwk = synth.evolve_balanced(16, 260, branch=0.02, sub=0.02, indel_start=0.006, mean_len=4, seed=14)
    one = walk(oracle, names, seqs, nwk, use_anchors=1, prefix_hit_length=20).align()
    return np.array(one.alignment()), np.array([one.node_info(k).score for k in range(one.n_internal)])


@pytest.mark.parametrize("world,exchange", [(2, "queue"), (3, "queue"), (2, "store"), (3, "store")])
def test_ranks_shard_one_tree_and_exchange_paths(tmp_path, oracle, pg, world, exchange):
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path), exchange, -1), nprocs=world, join=True)
    want_rows, want_scores = _one_rank_walk(oracle)
    owned = []
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("rows%d.npy" % r)), want_rows)          # every rank ends with the whole MSA
        assert np.array_equal(np.load(tmp_path / ("scores%d.npy" % r)), want_scores)
        owned += np.load(tmp_path / ("mine%d.npy" % r)).tolist()
        rounds = np.load(tmp_path / ("rounds%d.npy" % r))
        if exchange == "store":
            assert rounds[:, 0].tolist() == [8, 4, 2, 1]                                   # the tree's levels
            assert rounds[0, 1] in (8 // world, 8 // world + 1)                            # level 1 is split
    assert sorted(owned) == list(range(15))                                                # every node aligned exactly once


def test_a_slow_rank_holds_up_only_the_parents_of_its_own_nodes(tmp_path, oracle, pg):
    """The queue in the store is dynamic (node.cpp:196-223, 273-345: a thread takes the next ready node; a parent is ready
    when its two children are done, whoever aligned them).  Rank 1 takes three seconds over every batch and claims one
    node at a time; rank 0 must meanwhile align everything that does not descend from rank 1's node -- the other seven
    nodes of level 1, three of level 2, one of level 3 -- instead of waiting at a round's end."""
    mp.spawn(_rank, args=(2, _free_port(), str(tmp_path), "queue", 1), nprocs=2, join=True)
    want_rows, want_scores = _one_rank_walk(oracle)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / ("rows%d.npy" % r)), want_rows)
        assert np.array_equal(np.load(tmp_path / ("scores%d.npy" % r)), want_scores)
    fast, slow = np.load(tmp_path / "stamps0.npy"), np.load(tmp_path / "stamps1.npy")
    assert sorted(fast[:, 1].tolist() + slow[:, 1].tolist()) == [float(n) for n in range(16, 31)]
    t_slow_first = slow[:, 0].min()
    before = fast[fast[:, 0] < t_slow_first]
    assert before.shape[0] >= 11, "rank 0 aligned only %d nodes while rank 1 was busy with its first" % before.shape[0]
    # (internal nodes are numbered in post-order, node.h:479-495: 16 17 | 18, 19 20 | 21, 22, ...)
    levels = {n: 1 for n in (16, 17, 19, 20, 23, 24, 26, 27)}
    levels.update({18: 2, 21: 2, 25: 2, 28: 2, 22: 3, 29: 3, 30: 4})
    assert {levels[int(n)] for n in before[:, 1]} >= {1, 2, 3}, "rank 0 never left level 1 before rank 1 posted"


def test_ranks_shard_the_parent_graphs(tmp_path, oracle, pg):
    """Import stores the path only; a parent graph is built by the rank that aligned the node, or by a rank that claims a
    node above it -- not by every rank for every node on the thread that drains the posting log (node.cpp:196-223, 273-345:
    a thread builds the ancestor of the node it aligned).  With two ranks the walk leaves each of them with fewer parents
    built than there are nodes; asking for the rows afterwards builds the rest (every node's graph is needed for them)."""
    mp.spawn(_rank, args=(2, _free_port(), str(tmp_path), "queue", -1), nprocs=2, join=True)
    want_rows, _ = _one_rank_walk(oracle)
    n_nodes = 15
    built = [np.load(tmp_path / ("built%d.npy" % r)) for r in range(2)]
    mine = [np.load(tmp_path / ("mine%d.npy" % r)).shape[0] for r in range(2)]
    for r in range(2):
        assert built[r][0] >= mine[r]                      # its own nodes' parents, at least
        assert built[r][1] == n_nodes                      # the rows needed all of them
        assert np.array_equal(np.load(tmp_path / ("rows%d.npy" % r)), want_rows)
    assert min(b[0] for b in built) < n_nodes, "every rank built every parent during the walk: %s" % [b.tolist() for b in built]
    # the root's owner needs its two children, and theirs, ...: it ends with the whole tree; the other rank does not
    assert sum(b[0] for b in built) < 2 * n_nodes


def _rank_twice(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pagan2_msa_amd import dist as pdist, host as h, synth as sy
    import oracle
    oracle.build()
    names, seqs, nwk = sy.evolve_balanced(8, 200, branch=0.02, sub=0.02, indel_start=0.006, mean_len=4, seed=21)
    rows = []
    for rep in range(2):                        # the SAME walk name twice in one store: the second call must not see the first one's keys
        msa = h.Msa(names, seqs, nwk, use_anchors=0)
        msa.set_batch_backend(oracle_backend(oracle, []))
        pdist.align_sharded(msa, h.assign_units, exchange="queue", walk=7)
        rows.append(msa.alignment())
    np.save(os.path.join(out_dir, "twice%d.npy" % rank), np.array(rows))
    dist.barrier()
    dist.destroy_process_group()


def test_a_walk_name_reused_in_the_same_store(tmp_path, oracle, pg):
    """advisor, round 4: count / done / claim keys of a walk stayed in the store, so a second call under the same explicit
    name started from a stale count and lost every claim.  The key space now carries an epoch per name."""
    mp.spawn(_rank_twice, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    names, seqs, nwk = synth.evolve_balanced(8, 200, branch=0.02, sub=0.02, indel_start=0.006, mean_len=4, seed=21)
    want = np.array(walk(oracle, names, seqs, nwk, use_anchors=0).align().alignment())
    for r in range(2):
        got = np.load(tmp_path / ("twice%d.npy" % r))
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want)
