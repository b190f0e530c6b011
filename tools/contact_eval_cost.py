"""Kernel time of one (g) and one (g,H) evaluation with and without ground contact (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneChain, sceneChainGround  # noqa: E402

B = 1024
rng = np.random.default_rng(5)
for name, sc in (("plain", sceneChain(32)), ("ground", sceneChainGround(32, ground_z=-1.0))):
    sc.init()
    h = 5e-4
    q0 = rng.uniform(-0.05, 0.05, (B, 32))
    qd0 = rng.normal(size=(B, 32)) * 0.5
    sim = BatchSim(sc, batch=B)
    for rep in range(3):
        sim.eval_bdf1(q0 + h * qd0, q0, qd0, h, want_H=False)
        sim.eval_bdf1(q0 + h * qd0, q0, qd0, h, want_H=True)
    sim.set_state(q0, qd0)
    print(name, "V", sim.energy()[1][:3])
    sim.close()
