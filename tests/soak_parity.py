"""Full-size parity soak: every one of the 1024 benchmark rollouts x 100 BDF1 steps on the GPU against the CPU oracle
(OpenMP over rollouts on the box's host cores), with the benchmark's Newton tolerance.  Prints the worst relative errors."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from redmax_amd import BatchSim, sceneChain, syntheticStates  # noqa: E402

B, K, h, tol = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 100, 1e-2, 1e-8
sc = sceneChain(32)
sc.init()
q, qd = syntheticStates(32, B)
sim = BatchSim(sc, batch=B)
sim.opts.tol = tol
sim.set_state(q, qd)
out = sim.step_bdf1(K, h=h, stats=True)
qg, qdg = sim.get_state()
orc.set_newton(tol=tol)
qc, qdc = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
t0 = time.perf_counter()
orc.batch_step_bdf1(sc.desc(), qc, qdc, h, K, nthreads=os.cpu_count())
dt = time.perf_counter() - t0
eq = np.linalg.norm(qg - qc, axis=1) / np.linalg.norm(qc, axis=1)
ed = np.linalg.norm(qdg - qdc, axis=1) / np.maximum(np.linalg.norm(qdc, axis=1), 1e-30)
print("GPU kernel %.2f ms; oracle %.1f s on %d threads (%.0f rollout-steps/s)" % (out["ms"], dt, os.cpu_count(), B * K / dt))
print("q  : max rel err %.3e (rollout %d), median %.3e" % (eq.max(), int(eq.argmax()), np.median(eq)))
print("qd : max rel err %.3e (rollout %d), median %.3e" % (ed.max(), int(ed.argmax()), np.median(ed)))
print("status nonzero (GPU): %d; newton iters/step %.3f" % (int((out["status"] & 15 != 0).sum()), out["newton_iters"].sum() / (B * K)))
