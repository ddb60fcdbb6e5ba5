"""One-process-per-GPU plumbing (torch.distributed; backend "nccl" = RCCL on the GPU box, "gloo" in
the CPU tests).

The path shards by independent units -- guide-tree nodes whose two children are finished
(Node::start_openmp_alignment / build_queues, src/main/node.cpp:227-285) -- and the reference runs
them from a shared-memory queue.  With one process per GPU there is no shared memory, so every rank
holds the whole (small) tree state and the queue is replayed in rounds:

    ready = msa.ready()                      the same list on every rank
    mine  = the units the work-queue rule (pagan_assign_units, largest first) gives this rank
    msa.align_nodes(mine)                    model, anchors, DP on this rank's GPU, parent graphs
    post(exported results of `mine`)         into the job's key-value store (the rendezvous store torch.distributed
                                             was initialised with): path columns + used child edges of the nodes a
                                             rank aligned, ~1 byte per alignment column
    fetch + msa.import_result(...)           the other ranks' nodes of the round, as they appear; builds those
                                             parents locally

NO collective on the data path (north_star: "an embarrassingly-parallel work queue (no RCCL collectives)"): a finished
path is posted under its node's key and read by whoever needs it -- a mailbox, point to point through the store, and a
rank waits only for the nodes it has not aligned itself.  (Round 2 exchanged the same bytes with two all-gathers per
round; that is kept behind exchange="collective" for comparison.)  The ranks still meet in torch.distributed for what the
bench contract asks: the barrier around the timed region and the maximum of the elapsed time.
"""
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
_posted = {}                      # walk -> keys this rank posted (deleted two walks later: by then every rank has read them --
                                  # a rank starts walk k only after it has received every other rank's nodes of walk k-1, which
                                  # those ranks posted after finishing walk k-2)


def _store():
    from torch.distributed import distributed_c10d as c10d
    return c10d._get_default_store()


def align_sharded(msa, assign, device="cpu", on_round=None, exchange="store"):
    """The whole progressive alignment of `msa` (a host.Msa, created identically on every rank) with the
    ready nodes of each round dealt over the ranks.  `assign(costs, n_workers)` is the work-queue rule
    (host.assign_units).  Returns per-round records [(n_ready, n_mine, bytes exchanged)]."""
    w, r = world(), rank()
    rounds = []
    walk = _walks[0]              # (every rank calls this the same number of times: the keys of one walk never meet another's)
    _walks[0] += 1
    store = _store() if (w > 1 and exchange == "store") else None
    if store is not None:
        for key in _posted.pop(walk - 2, []):
            try:
                store.delete_key(key)
            except Exception:          # (a store without delete: the keys just stay)
                pass
        _posted[walk] = []
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
                store.set("pagan/%d/%d" % (walk, n), c.tobytes())
                _posted[walk].append("pagan/%d/%d" % (walk, n))
            for n, o in zip(ready, owner):
                if int(o) == r:
                    continue
                buf = np.frombuffer(store.get("pagan/%d/%d" % (walk, n)), np.uint8)        # (blocks until the owner has posted it)
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
    msa.finish()
    return rounds
