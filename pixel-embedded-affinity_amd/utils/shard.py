"""Multi-GPU sharding of the hot path: one process per GPU, images (or 3D sub-volumes) are independent units.

The reference runs the loss on cuda:0 over the gathered batch of an nn.DataParallel model and requires
`batch_size % n_gpu == 0` (scripts_cvppp/main.py:117-125).  Here each rank keeps its own slice of the batch and
runs the fused op with the LOCAL normaliser 1/(b*W), b = B/world; averaging gradients over ranks (DDP semantics,
RCCL all-reduce sum then / world) reproduces the global 1/(B*W) exactly:
    (1/G) * sum_r (1/(b W)) * sum_local(...) == (1/(B W)) * sum_all(...)
The op itself exchanges nothing; the only collectives are the backbone-gradient all-reduce and (optionally) one
small all-reduce of the logged loss scalars.
"""
import torch
import torch.distributed as dist


def shard_range(batch, rank, world):
    """[start, stop) of the batch items owned by `rank` (contiguous, equal shares like the reference)."""
    if batch % world != 0:
        raise ValueError("Batch size (%d) cannot be equally divided by GPU number (%d)" % (batch, world))
    b = batch // world
    return rank * b, (rank + 1) * b


def shard(t, rank, world):
    lo, hi = shard_range(t.shape[0], rank, world)
    return t[lo:hi]


def allreduce_mean_(tensors, group=None):
    """In-place DDP-style averaging of gradients / logged scalars over ranks (no-op in a single process)."""
    if not (dist.is_available() and dist.is_initialized()):
        return tensors
    world = dist.get_world_size(group)
    for t in ([tensors] if isinstance(tensors, torch.Tensor) else tensors):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.div_(world)
    return tensors
