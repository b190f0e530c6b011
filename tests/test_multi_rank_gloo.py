"""N>1 path on CPU: world_size-2 gloo run of the sharding layer (redmax_amd/sharding.py).  Each rank owns a
contiguous block of the global batch, builds its inputs from global indices, steps its block independently
(the oracle stands in for the kernel: there is no GPU here) and the single collective gathers (q, qdot).
Property: the gathered result equals the single-process run bit for bit (shard invariance)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from redmax_amd.scenes import scenesRedMax, syntheticStates
from redmax_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _step_block(q, qd, K):
    from oracle import oracle as orc
    sc = scenesRedMax(2)
    sc.init()
    q, qd = q.copy(), qd.copy()
    for b in range(q.shape[0]):
        o = orc.Oracle(sc.desc())
        o.set_state(q[b], qd[b])
        o.step_bdf1(sc.h, K)
        q[b], qd[b] = o.get_state()
    return q, qd


def _worker(rank, world, port, B, K, scaling, out):
    sharding.init_process_group(rank, world, "gloo", port=port)
    sh = sharding.plan(rank, world, B, scaling)
    q, qd = syntheticStates(4, sh.count, first=sh.first)
    q, qd = _step_block(q, qd, K)
    qa, qda = sharding.gather_states(torch.from_numpy(q), torch.from_numpy(qd), sh)
    assert qa.shape[0] == sh.global_batch
    assert sharding.max_over_ranks(float(rank)) == world - 1
    if rank == 0:
        np.save(out, np.stack([qa.numpy(), qda.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_is_shard_invariant(tmp_path):
    """weak scaling: 3 rollouts per rank, equal shards, one all_gather_into_tensor per array."""
    world, B, K = 2, 3, 4
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), B, K, "weak", out), nprocs=world, join=True)
    got = np.load(out)
    q, qd = syntheticStates(4, world * B, first=0)
    qs, qds = _step_block(q, qd, K)
    assert np.array_equal(got[0], qs) and np.array_equal(got[1], qds)


def test_strong_scaling_with_uneven_shards_is_shard_invariant(tmp_path):
    """strong scaling: 5 rollouts in total over 2 ranks (3 + 2): the padded gather must return them in global order."""
    world, B, K = 2, 5, 3
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), B, K, "strong", out), nprocs=world, join=True)
    got = np.load(out)
    q, qd = syntheticStates(4, B, first=0)
    qs, qds = _step_block(q, qd, K)
    assert np.array_equal(got[0], qs) and np.array_equal(got[1], qds)


def test_shard_plans():
    assert [sharding.shard_range(r, 3, 10) for r in range(3)] == [(0, 4), (4, 3), (7, 3)]
    p = sharding.plan(1, 8, 1024, "strong")
    assert (p.first, p.count, p.global_batch, p.even) == (128, 128, 1024, True)
    p = sharding.plan(3, 4, 1024, "weak")
    assert (p.first, p.count, p.global_batch) == (3072, 1024, 4096)
    assert sum(sharding.plan(0, 8, 1000, "strong").counts) == 1000
