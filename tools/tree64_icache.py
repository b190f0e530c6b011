"""64-joint tree, one wavefront per trajectory, at 256 and at 512 rollouts (one or two workgroups per CU): run under
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS to see what the
second workgroup of a CU contends for."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneTree, syntheticStates  # noqa: E402

sc = sceneTree(64)
sc.init()
for B in (256, 512):
    q, qd = syntheticStates(sc.nr, B)
    q = q * 0.5 + sc.getQ()[0]
    sim = BatchSim(sc, batch=B)
    sim.opts.tol = 1e-9
    for rep in range(2):
        sim.set_state(q, qd)
        o = sim.step_bdf1(100, h=1e-2)
    print("B=%d: %.3f ms per 100 steps" % (B, o["ms"]))
    sim.close()
