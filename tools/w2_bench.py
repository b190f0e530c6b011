"""tree64 (BASELINE.json configs[2] per-GPU share) with the one-wave and the two-wave step kernel: kernel ms per 100 steps."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneTree  # noqa: E402

sc = sceneTree(64)
sc.init()
qs, _ = sc.getQ()
for B in (128, 256, 512, 1024):
    q = np.empty((B, sc.nr))
    qd = np.empty((B, sc.nr))
    for i in range(B):
        rng = np.random.default_rng(20240 + i)
        q[i] = qs + rng.uniform(-0.05, 0.05, sc.nr)
        qd[i] = rng.uniform(-0.1, 0.1, sc.nr)
    for mode in ("0", "1"):
        os.environ["RMX_W2"] = mode
        sim = BatchSim(sc, batch=B)
        ms = []
        for r in range(3):
            sim.set_state(q, qd)
            sim.step_bdf1(5, h=1e-2)
            out = sim.step_bdf1(100, h=1e-2, stats=True)
            ms.append(out["ms"])
        print("tree64 B=%4d waves/traj=%d: %.3f ms per 100 steps (min of 3) -> %.2f M rollout-steps/s; iters/step %.2f; bad %d pivoted %d" % (
            B, 1 + int(mode), min(ms), B * 100 / min(ms) / 1e3, out["newton_iters"].mean() / 100, int((out["status"] & 15 != 0).sum()),
            int((out["status"] & 16 != 0).sum())), flush=True)
        sim.close()
