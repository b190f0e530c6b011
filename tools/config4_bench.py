"""BASELINE.json configs[3]: adjoint BDF1 forward+backward, 16-DOF chain, batch=512 (one GPU)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneAdjointChain, driverRedMaxAdjointBDF1  # noqa: E402

sc = sceneAdjointChain(16)
sc.init()
B = 512
rng = np.random.default_rng(0)
p = 0.1 * rng.standard_normal((B, sc.nr))        # p ~ N(0, 1e-2)
sim = BatchSim(sc, batch=B)
q0, qd0 = sc.getQ()
for rep in range(3):
    sim.set_state(q0[None, :], qd0[None, :])
    t0 = time.perf_counter()
    P, dPdp, info = sim.adjoint_bdf1(sc.nsteps, sc.h, sc.task, p, stats=True)
    wall = time.perf_counter() - t0
    print("adjoint 16-DOF B=%d, %d steps: kernels %.2f ms (wall incl. alloc/copies %.1f ms); newton iters/step %.2f; status!=0: %d; P mean %.4g; |dPdp| mean %.4g"
          % (B, sc.nsteps, info["ms"], 1e3 * wall, info["newton_iters"].mean() / sc.nsteps, int((info["status"] != 0).sum()), P.mean(), np.linalg.norm(dPdp, axis=1).mean()))
print("-> %.2f M forward+backward rollout-steps/s" % (B * sc.nsteps / info["ms"] / 1e3))
scene, res = driverRedMaxAdjointBDF1(2, verbose=False, maxiter=30)
print("driverRedMaxAdjointBDF1 (scene 100, BFGS): P %.6g after %d iterations, p = %s" % (res.fun, res.nit, np.array2string(res.x, precision=4)))
