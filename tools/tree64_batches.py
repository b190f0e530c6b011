"""64-joint tree (BASELINE.json configs[2]) at several batch sizes: kernel time per 100 BDF1 steps with the per-node constants in LDS
(two wavefronts per CU) and in global memory (four per CU).  RMX_GCONST_MIN is read at model creation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from redmax_amd import BatchSim, sceneTree, syntheticStates  # noqa: E402

sc = sceneTree(64)
sc.init()
for B in (256, 512, 768, 1024, 2048):
    q, qd = syntheticStates(sc.nr, B)
    q = q * 0.5 + sc.getQ()[0]
    line = "B=%4d:" % B
    for thr, name in (("100000", "LDS consts"), ("1", "global consts"), (None, "default")):
        if thr is None:
            os.environ.pop("RMX_GCONST_MIN", None)
        else:
            os.environ["RMX_GCONST_MIN"] = thr
        sim = BatchSim(sc, batch=B)
        ms = []
        for rep in range(3):
            sim.set_state(q, qd)
            sim.step_bdf1(5, h=1e-2)
            o = sim.step_bdf1(100, h=1e-2, stats=True)
            ms.append(o["ms"])
        line += "  %s %.3f ms (%.2f M rollout-steps/s)" % (name, np.median(ms), B * 100 / np.median(ms) / 1e3)
        sim.close()
    print(line, flush=True)
