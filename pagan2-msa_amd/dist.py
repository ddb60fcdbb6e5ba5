"""One-process-per-GPU plumbing for bench.py (torch.distributed; backend "nccl" = RCCL on the GPU
box, "gloo" in the CPU tests).  The data path has no collective: ranks only meet at the barrier
and to reduce the timing."""
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
