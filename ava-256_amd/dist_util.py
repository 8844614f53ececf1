"""Multi-GPU plumbing for the hot path: one process per GPU, cameras sharded, no data-path collective.

The march is independent per ray and every camera/frame n owns its template and pose tensors
(SURVEY.md section 8e), so ranks simply take contiguous camera shards.  The only collectives are the
barrier and the MAX-reduce of the elapsed time that bench.py's contract prescribes (RCCL on GPUs,
gloo in the CPU tests)."""
import os

import torch


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total, rank, world):
    """Contiguous [lo, hi) shard of `total` units for `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, device="cpu"):
    """MAX-reduce a python float over all ranks (identity when not initialised)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
