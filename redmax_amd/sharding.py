"""Multi-GPU layer: the batch axis shards, nothing else does.

Trajectories are independent, so rank r of W owns the contiguous block [r*B, (r+1)*B) of the global batch,
generates its inputs from the GLOBAL trajectory index (results are shard-invariant) and steps with no
communication.  The single collective of the path is the final gather of (q, qdot): one all-gather per rollout
(RCCL over xGMI on GPUs - backend "nccl"; gloo in the CPU tests)."""
from __future__ import annotations


def shard_first(rank, batch_per_rank):
    """Global index of the first trajectory owned by `rank` (weak scaling: batch_per_rank is fixed)."""
    return int(rank) * int(batch_per_rank)


def gather_states(q_loc, qd_loc, q_all=None, qd_all=None):
    """all_gather the per-rank [B][nr] state tensors into [W*B][nr] tensors (rank order = global order)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if q_all is None:
        q_all = torch.empty((world * q_loc.shape[0],) + tuple(q_loc.shape[1:]), dtype=q_loc.dtype, device=q_loc.device)
    if qd_all is None:
        qd_all = torch.empty_like(q_all)
    dist.all_gather_into_tensor(q_all, q_loc.contiguous())
    dist.all_gather_into_tensor(qd_all, qd_loc.contiguous())
    return q_all, qd_all
