"""Config 5 at batch sizes the one-launch form must survive: more rollouts than SIMDs (the cooperative workgroups are dispatched behind
ALL rollout workgroups), an uneven batch, and two shards of 1024 on one device (two k_ground32 launches competing for the SIMDs).
Every case against one wavefront per rollout (RMX_PARK_HALVINGS=0): states bit-identical, no group gave up."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from redmax_amd import BatchSim, GroupSim, sceneChainGround, syntheticStates

sc = sceneChainGround(32)
sc.init()


def states(B):
    q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1)
    q[0], qd[0] = sc.getQ()
    return q, qd


def single(B, park):
    os.environ["RMX_PARK_HALVINGS"] = park
    q, qd = states(B)
    sim = BatchSim(sc, batch=B)
    sim.set_state(q, qd)
    t = time.time()
    o = sim.step_bdf2(100, h=sc.h, stats=True)
    qf, _ = sim.get_state()
    sim.close()
    return qf, o, time.time() - t


for B in (2048, 1500, 37):
    ref, o0, _ = single(B, "0")
    got, o1, w = single(B, "24")
    print("B %4d: kernel %.2f ms (one wavefront per rollout %.2f), bit-identical %s, gave up %d, wall %.2f s" % (
        B, o1["ms"], o0["ms"], np.array_equal(ref, got), int((o1["status"] & 512 != 0).sum()), w))
os.environ["RMX_PARK_HALVINGS"] = "24"
B = 2048
q, qd = states(B)
ref, _, _ = single(B, "24")
g = GroupSim(sc, B, devices=(0, 0))
g.set_state(q, qd)
t = time.time()
out = g.step(100, integrator=2, h=sc.h)
qf, _ = g.get_state()
print("two shards of 1024 on device 0: wall %.2f ms, kernels %s, bit-identical to one batch %s, gave up %d, host wall %.2f s" % (
    out["wall_ms"], np.round(out["kernel_ms"], 2), np.array_equal(qf, ref), int((out["status"] & 512 != 0).sum()), time.time() - t))
g.close()
