"""Adjoint BDF1 forward + backward at 16 / 32 / 40 / 64 links, 512 rollouts: kernel time, Newton iterations per step, P of rollout 1 (the
32-, 40- and 64-link cases run the 32- and 64-lane forward kernels).  Usage: adjoint_sizes.py [lib.so]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import _abi  # noqa: E402
if len(sys.argv) > 1:
    _abi.LIB_PATH = sys.argv[1]
from redmax_amd import BatchSim, sceneAdjointChain  # noqa: E402

for n, K in ((16, 20), (32, 10), (40, 6), (64, 4)):
    sc = sceneAdjointChain(n)
    sc.init()
    B = 512
    p = 0.1 * np.random.default_rng(0).standard_normal((B, sc.nr))
    sim = BatchSim(sc, batch=B)
    q0, qd0 = sc.getQ()
    ms = []
    for rep in range(5):
        sim.set_state(q0[None, :], qd0[None, :])
        P, dPdp, info = sim.adjoint_bdf1(K, sc.h, dict(sc.task, t=K * sc.h), p, stats=True)
        ms.append(info["ms"])
    print("adjoint chain %d: %d steps x %d rollouts, kernels min %.3f ms, %.2f Newton iterations per step, P[1] %.12e, pivoted fallbacks %d"
          % (n, K, B, min(ms), info["newton_iters"].mean() / K, P[1], int(((info["status"] & 16) != 0).sum())), flush=True)
    sim.close()
