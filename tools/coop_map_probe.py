import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from redmax_amd import _abi
_abi.LIB_PATH = os.path.join(os.getcwd(), "redmax_amd/variants/libredmax_hip_mapaid.so")
from redmax_amd import BatchSim, sceneChainGround, syntheticStates
sc = sceneChainGround(32); sc.init(); B = 1024
q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1); q[0], qd[0] = sc.getQ()
sim = BatchSim(sc, batch=B)
for cm in ("0", "1", "0", "1"):
    os.environ["RMX_COOP_MAP"] = cm
    sim.set_state(q, qd)
    o = sim.step_bdf2(100, h=sc.h, stats=True)
    print("coop_map", cm, "ms", o["ms"], o["newton_iters"].sum())
