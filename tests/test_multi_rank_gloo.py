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
from redmax_amd.sharding import gather_states, shard_first


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


def _worker(rank, world, port, B, K, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q, qd = syntheticStates(4, B, first=shard_first(rank, B))
    q, qd = _step_block(q, qd, K)
    qa, qda = gather_states(torch.from_numpy(q), torch.from_numpy(qd))
    if rank == 0:
        np.save(out, np.stack([qa.numpy(), qda.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_is_shard_invariant(tmp_path):
    world, B, K = 2, 3, 4
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), B, K, out), nprocs=world, join=True)
    got = np.load(out)
    q, qd = syntheticStates(4, world * B, first=0)
    qs, qds = _step_block(q, qd, K)
    assert np.array_equal(got[0], qs) and np.array_equal(got[1], qds)
