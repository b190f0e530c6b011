import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from redmax_amd import BatchSim, sceneChain, syntheticStates
sc = sceneChain(32); sc.init()
for B in (512, 1024, 2048, 4096, 8192):
    q, qd = syntheticStates(32, B)
    sim = BatchSim(sc, batch=B); sim.opts.tol = 1e-8
    ms = []
    for r in range(3):
        sim.set_state(q, qd); sim.step_bdf1(10, h=1e-2)
        ms.append(sim.step_bdf1(100, h=1e-2)["ms"])
    print("B=%5d: %.2f ms per 100 steps -> %.2f M rollout-steps/s" % (B, min(ms), B * 100 / min(ms) / 1e3))
    sim.close()
