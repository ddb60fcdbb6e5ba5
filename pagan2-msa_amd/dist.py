"""One-process-per-GPU plumbing (torch.distributed; backend "nccl" = RCCL on the GPU box, "gloo" in
the CPU tests).

The path shards by independent units -- guide-tree nodes whose two children are finished
(Node::start_threaded_alignment / build_queues, src/main/node.cpp:196-223, 273-345) -- and the reference runs
them from a shared-memory queue: a thread takes the next ready node, and a parent becomes ready the moment its
two children are done, whoever aligned them.  With one process per GPU there is no shared memory, so every rank
holds the whole (small) tree state and the queue lives in the job's key-value store (the rendezvous store
torch.distributed was initialised with):

    exchange="queue" (default)   a DYNAMIC queue.  A rank's ready set comes from what it has aligned itself plus what
                                 it has imported; it CLAIMS ready nodes, largest first, with an atomic counter per node
                                 (store.add: the first rank to ask owns the node), aligns them on its GPU (model, anchors,
                                 DP, parent graphs), posts the finished paths (node key, then an entry in a posting log)
                                 and imports what the others have posted since it last looked -- without waiting.  It only
                                 blocks when it has nothing ready and nothing to import, and then on the NEXT posting of
                                 whoever finishes first.  A slow rank holds up the parents of its own nodes and nothing
                                 else; there is no round and no barrier.
    exchange="store"             level-synchronous rounds with the store as a mailbox (round 3): the ready list is the same
                                 on every rank, pagan_assign_units deals it statically, every rank reads every other
                                 rank's nodes of the round before the next round starts.
    exchange="collective"        the same rounds with two all-gathers per round (round 2), for comparison.

NO collective on the data path in the first two (north_star: "an embarrassingly-parallel work queue (no RCCL
collectives)"): what travels is a finished path, ~1 byte per alignment column plus the used edge ids.  The ranks still
meet in torch.distributed for what the bench contract asks: the barrier around the timed region and the maximum of the
elapsed time.
"""
import datetime
import os
import time

import numpy as np
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def barrier():
    if world() > 1:
        dist.barrier()


def reduce_step(elapsed_s, units, device="cpu"):
    """(max elapsed over ranks, sum of units over ranks)."""
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    if world() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def my_units(costs, assign):
    """Indices of the units this rank owns under the work-queue rule `assign(costs, n_workers)`."""
    owner = assign(costs, world())
    return [k for k, o in enumerate(owner) if int(o) == rank()]


def all_gather_bytes(chunks, device="cpu"):
    """chunks: list of uint8 numpy arrays of this rank.  Returns, per rank, its list of chunks.
    Two collectives: the chunk sizes (fixed shape), then the padded payloads."""
    w = world()
    if w == 1:
        return [list(chunks)]
    sizes = torch.tensor([len(chunks)] + [int(c.shape[0]) for c in chunks], dtype=torch.int64, device=device)
    n_max = torch.tensor([sizes.shape[0]], dtype=torch.int64, device=device)
    dist.all_reduce(n_max, op=dist.ReduceOp.MAX)
    pad = torch.zeros(int(n_max.item()), dtype=torch.int64, device=device)
    pad[:sizes.shape[0]] = sizes
    all_sizes = [torch.zeros_like(pad) for _ in range(w)]
    dist.all_gather(all_sizes, pad)
    all_sizes = [s.cpu().numpy() for s in all_sizes]
    totals = [int(s[1:1 + int(s[0])].sum()) for s in all_sizes]
    cap = max(max(totals), 1)
    payload = torch.zeros(cap, dtype=torch.uint8, device=device)
    if chunks:
        flat = np.concatenate(chunks) if len(chunks) > 1 else chunks[0]
        payload[:flat.shape[0]] = torch.from_numpy(np.ascontiguousarray(flat)).to(device)
    gathered = [torch.zeros_like(payload) for _ in range(w)]
    dist.all_gather(gathered, payload)
    out = []
    for r in range(w):
        buf = gathered[r].cpu().numpy()
        cnt = int(all_sizes[r][0])
        lens = all_sizes[r][1:1 + cnt]
        offs = np.concatenate([[0], np.cumsum(lens)])
        out.append([buf[int(offs[k]):int(offs[k + 1])] for k in range(cnt)])
    return out


_walks = [0]
_posted = {}                      # key space -> keys this rank posted; deleted once every rank has acknowledged the walk


def _store():
    from torch.distributed import distributed_c10d as c10d
    return c10d._get_default_store()


def _get_blocking(store, key, what):
    """store.get(key) once the key exists.  Waits in slices (a store's own timeout would end the walk when one node's
    alignment outlasts it) up to PAGAN_STORE_TIMEOUT_S seconds (default two hours): a rank that died after claiming a node
    ends in an error that names the node, not in a silent hang."""
    deadline = time.time() + float(os.environ.get("PAGAN_STORE_TIMEOUT_S", "7200"))
    while True:
        try:
            store.wait([key], datetime.timedelta(seconds=20))
            return store.get(key)
        except RuntimeError as e:             # (torch.distributed.DistStoreError is one)
            if "imeout" not in str(e) and "imed out" not in str(e):
                raise
            if time.time() > deadline:
                raise RuntimeError("pagan2_msa_amd.dist: gave up waiting for %s (%s): did its owner die?" % (what, key)) from e


def _key_space(store, walk, r):
    """Key prefix of this call: the walk's name plus an EPOCH -- the number of earlier calls this rank made under the same
    name (a counter per rank in the store).  A name reused in the same store (tests pass walk=7; a service might number
    its jobs modulo something) starts from fresh keys instead of a stale count, stale claims and a finished log."""
    epoch = int(store.add("pagan/%d/epoch/%d" % (walk, r), 1)) - 1
    return "pagan/%d.%d/" % (walk, epoch)


def _reap(store, w):
    """Deletes this rank's keys of earlier walks that EVERY rank has finished reading (each rank adds 1 to the walk's
    `done` counter when it has imported everything): no assumption about who posted what in which walk."""
    for old in list(_posted):
        try:
            if store.add(old + "done", 0) < w:
                continue
            for key in _posted.pop(old):
                store.delete_key(key)
        except Exception:              # (a store without delete: the keys just stay)
            _posted.pop(old, None)


def _align_queue(msa, store, P, w, r, on_round, quota):
    mine_all, lost = set(), set()
    seen = 0                      # entries of the posting log this rank has gone through
    rounds = []
    if r == 0:
        _posted[P] += [P + "count", P + "done"]       # (the counters themselves: rank 0 deletes them with its keys)

    def drain(block):
        """Imports what has been posted since the last look, in posting order (a parent is always posted after its
        children: its poster had both before it could align it).  block: wait for the next posting first."""
        nonlocal seen
        moved = 0
        cnt = int(store.add(P + "count", 0))
        if block and cnt == seen:
            cnt = seen + 1
        while seen < cnt:
            seen += 1
            n = int(_get_blocking(store, P + "log/%d" % seen, "posting %d of the walk" % seen))     # (until the poster has written the entry)
            if n in mine_all:
                continue
            buf = np.frombuffer(_get_blocking(store, P + "node/%d" % n, "node %d" % n), np.uint8)
            msa.import_result(buf)
            lost.discard(n)
            moved += int(buf.shape[0])
        return moved

    while msa.remaining > 0:
        moved = drain(False)
        ready = [n for n in msa.ready() if n not in lost]
        if not ready:
            moved += drain(True)
            rounds.append((0, 0, moved))
            continue
        ready.sort(key=lambda n: -msa.node_cost(n))
        q = quota if quota else max(1, -(-len(ready) // w))
        start = (r * q) % len(ready)          # ranks start at different places of the same list: fewer lost claims
        mine = []
        for k in range(len(ready)):
            if len(mine) >= q:
                break
            n = ready[(start + k) % len(ready)]
            if int(store.add(P + "claim/%d" % n, 1)) == 1:
                mine.append(n)
            else:
                lost.add(n)                   # its owner will post it
        if mine:
            msa.align_nodes(mine)
            for n in mine:
                mine_all.add(n)
                store.set(P + "node/%d" % n, msa.export_result(n).tobytes())
                idx = int(store.add(P + "count", 1))
                store.set(P + "log/%d" % idx, str(n))
                _posted[P] += [P + "node/%d" % n, P + "log/%d" % idx, P + "claim/%d" % n]
        rounds.append((len(ready), len(mine), moved))
        if on_round:
            on_round(len(rounds) - 1, ready, mine)
    return rounds


def align_sharded(msa, assign, device="cpu", on_round=None, exchange="queue", walk=None, quota=None, lazy_rows=True):
    """The whole progressive alignment of `msa` (a host.Msa, created identically on every rank) sharded over the ranks.
    exchange: "queue" (dynamic claims through the store), "store" / "collective" (level-synchronous rounds dealt by
    `assign(costs, n_workers)`, the work-queue rule host.assign_units) -- see the module text.  `walk` names the key space
    of this call in the store; None: a per-process call counter, which is only right when every rank calls this the same
    number of times.  `quota`: most nodes a rank claims at a time in queue mode (None: its share of what is ready).
    lazy_rows: the parent graphs of imported nodes this rank never needed, and the rows, are left to the first call that asks
    for them (Msa.finish(lazy=True)): the ranks shard the parent graphs as they shard the alignments.
    Returns per-round records [(n_ready, n_mine, bytes imported)]."""
    w, r = world(), rank()
    rounds = []
    if walk is None:
        walk = _walks[0]
        _walks[0] += 1
    store = _store() if (w > 1 and exchange in ("store", "queue")) else None
    P = None
    if store is not None:
        _reap(store, w)
        P = _key_space(store, walk, r)
        _posted[P] = []
    if store is not None and exchange == "queue":
        rounds = _align_queue(msa, store, P, w, r, on_round, quota)
        store.add(P + "done", 1)
        msa.finish(lazy=lazy_rows)
        return rounds
    if store is not None and r == 0:
        _posted[P].append(P + "done")
    while msa.remaining > 0:
        ready = msa.ready()
        if not ready:
            raise RuntimeError("no ready node although %d remain" % msa.remaining)
        costs = [msa.node_cost(n) for n in ready]
        owner = assign(costs, min(w, len(ready)))
        mine = [n for n, o in zip(ready, owner) if int(o) == r]
        if mine:
            msa.align_nodes(mine)
        chunks = [msa.export_result(n) for n in mine]
        moved = 0
        if store is not None:
            for n, c in zip(mine, chunks):
                store.set(P + "%d" % n, c.tobytes())
                _posted[P].append(P + "%d" % n)
            for n, o in zip(ready, owner):
                if int(o) == r:
                    continue
                buf = np.frombuffer(_get_blocking(store, P + "%d" % n, "node %d" % n), np.uint8)        # (until the owner has posted it)
                msa.import_result(buf)
                moved += int(buf.shape[0])
        elif w > 1:
            for src, theirs in enumerate(all_gather_bytes(chunks, device=device)):
                if src == r:
                    continue
                for c in theirs:
                    msa.import_result(c)
                    moved += int(c.shape[0])
        rounds.append((len(ready), len(mine), moved))
        if on_round:
            on_round(len(rounds) - 1, ready, mine)
    if store is not None:
        store.add(P + "done", 1)
    msa.finish(lazy=lazy_rows)
    return rounds
