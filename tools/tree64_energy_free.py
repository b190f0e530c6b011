import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from redmax_amd import BatchSim, sceneTree, syntheticStates
sc = sceneTree(64); sc.init()
for B in (512, 256):
    q, qd = syntheticStates(sc.nr, B); q = q * 0.5 + sc.getQ()[0]
    res = {}
    for hist in (True, False):
        sim = BatchSim(sc, batch=B); sim.set_state(q, qd); sim.step_bdf1(10, h=1e-2)
        q0, qd0 = sim.get_state(); ms = []
        for r in range(5):
            sim.set_state(q0, qd0); o = sim.step_bdf1(100, h=1e-2, stats=True, history=hist); ms.append(o["ms"])
        res[hist] = (min(ms), sim.get_state(), o["newton_iters"].copy(), sim.last_step_kernel())
        sim.close()
    (a, (qa, qda), ia, ka), (b, (qb, qdb), ib, kb) = res[True], res[False]
    print("tree64 B=%d: with energy record %s %.3f ms, without %s %.3f ms per 100 steps (x%.3f); same bits %s" % (B, ka, a, kb, b, a / b, np.array_equal(qa, qb) and np.array_equal(qda, qdb) and np.array_equal(ia, ib)))
