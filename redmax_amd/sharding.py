"""Multi-GPU layer: the batch axis shards, nothing else does.

Trajectories are independent (no coupling term anywhere in evalBDF1), the model is read-only and replicated, so a job of
`global_batch` rollouts over W ranks is W independent launches plus ONE collective: the final gather of (q, qdot)
(RCCL over xGMI on GPUs - torch.distributed backend "nccl"; gloo in the CPU tests and when several ranks have to share
one device).  This module is the whole of that layer:

  plan(rank, world, batch, scaling)   which rollouts a rank owns.  "weak": every rank owns `batch` rollouts (the job grows
                                      with W); "strong": `batch` rollouts in total, split into W contiguous blocks whose
                                      sizes differ by at most one (BASELINE.json north_star: "a 1024-rollout batch at
                                      1/2/4/8 MI355X").  Inputs are generated from the GLOBAL rollout index
                                      (scenes.syntheticStates(first=plan.first)), so results do not depend on W.
  init_process_group(...)             rendezvous on 127.0.0.1, one process per GPU, picks the backend.
  gather_states(q, qdot, plan)        the collective: every rank ends up with the [global_batch][nr] state in rank order
                                      = global rollout order.  Equal shards go through one all_gather_into_tensor per
                                      array; unequal shards (strong scaling with W not dividing the batch) are padded to
                                      the largest shard and trimmed after the gather.
  max_over_ranks(x)                   the bench's timing reduction.

The xGMI fabric is point-to-point (7 links per GPU); the gather moves 2 x 8 x nr x global_batch bytes in total (512 KiB for
the 1024 x 32 headline job), so one direct all-gather at the end of the rollout is latency-, not bandwidth-bound, and
there is nothing to bucket or overlap: no per-step communication exists.
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass(frozen=True)
class ShardPlan:
    rank: int
    world: int
    scaling: str          # "weak" | "strong"
    global_batch: int     # rollouts in the whole job
    first: int            # global index of this rank's first rollout
    count: int            # rollouts this rank owns (may be 0 in strong scaling when world > batch)

    @property
    def counts(self):
        """Rollouts owned by every rank, in rank order."""
        return [shard_range(r, self.world, self.global_batch)[1] for r in range(self.world)]

    @property
    def even(self):
        c = self.counts
        return min(c) == max(c)


def shard_range(rank, world, global_batch):
    """(first, count) of the contiguous block of `global_batch` rollouts that `rank` owns: sizes differ by at most one,
    the first `global_batch % world` ranks get the extra rollout."""
    rank, world, global_batch = int(rank), int(world), int(global_batch)
    base, extra = divmod(global_batch, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def plan(rank, world, batch, scaling="weak"):
    """scaling="weak": `batch` rollouts PER RANK; "strong": `batch` rollouts in TOTAL."""
    if scaling not in ("weak", "strong"):
        raise ValueError("scaling must be 'weak' or 'strong'")
    rank, world, batch = int(rank), int(world), int(batch)
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    total = batch * world if scaling == "weak" else batch
    first, count = shard_range(rank, world, total)
    return ShardPlan(rank, world, scaling, total, first, count)


def shard_first(rank, batch_per_rank):
    """Global index of the first rollout owned by `rank` under weak scaling (kept for callers of the first version)."""
    return int(rank) * int(batch_per_rank)


def env_rank():
    """(rank, local_rank, world) as torch.distributed.run exports them (1 process when absent)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(rank, world, backend, device=None, port=None):
    """Rendezvous on 127.0.0.1 (the container hostname may not resolve).  backend "nccl" = RCCL on ROCm."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if port is not None:
        os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL needs it)
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = torch.device("cuda", device)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def gather_states(q_loc, qd_loc, shard=None, q_all=None, qd_all=None):
    """all-gather the per-rank [count][nr] state tensors into [global_batch][nr] tensors (rank order = global order).
    `shard`: this rank's ShardPlan; None means equal shards (weak scaling)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if shard is None or shard.even:
        if q_all is None:
            q_all = torch.empty((world * q_loc.shape[0],) + tuple(q_loc.shape[1:]), dtype=q_loc.dtype, device=q_loc.device)
        if qd_all is None:
            qd_all = torch.empty_like(q_all)
        dist.all_gather_into_tensor(q_all, q_loc.contiguous())
        dist.all_gather_into_tensor(qd_all, qd_loc.contiguous())
        return q_all, qd_all
    counts = shard.counts
    cmax = max(counts)
    outs = []
    for x in (q_loc, qd_loc):
        pad = torch.zeros((cmax,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        pad[: x.shape[0]] = x
        buf = torch.empty((world * cmax,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(buf, pad)
        outs.append(torch.cat([buf[r * cmax: r * cmax + counts[r]] for r in range(world)], dim=0))
    return outs[0], outs[1]


def max_over_ranks(value, device=None):
    """MAX of a python float over the ranks (the bench's elapsed-time reduction); identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
