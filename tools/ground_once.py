"""One config-5 launch (32-link chain over frictional ground, BDF2, 1024 x 100) for kernel traces: RMX_PARK_HALVINGS / RMX_GROUND_FUSED from the environment."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneChainGround, syntheticStates
sc = sceneChainGround(32); sc.init(); B = 1024
q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1); q[0], qd[0] = sc.getQ()
sim = BatchSim(sc, batch=B)
for r in range(3):
    sim.set_state(q, qd)
    o = sim.step_bdf2(100, h=sc.h, stats=True)
    print("ms", o["ms"])
