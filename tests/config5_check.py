import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from redmax_amd import BatchSim, sceneChainGround, syntheticStates
from oracle.oracle import Oracle
B=256
sc = sceneChainGround(32); sc.init()
q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1); q[0], qd[0] = sc.getQ()
sim = BatchSim(sc, batch=B)
sim.set_state(q, qd)
out = sim.step_bdf2(100, h=sc.h, stats=True, history=True)
st = out["status"]; it = out["newton_iters"]
bad = np.nonzero(st & 2)[0]
print("flagged", len(bad), bad[:10], "iters", it[bad[:10]])
qg, qdg = sim.get_state()
for b in list(bad[:3]) + [1]:
    o = Oracle(sc.desc()); o.set_state(q[b], qd[b])
    t=time.time(); s, T, V = o.step_bdf2(sc.h, 100, history=True); 
    qo, qdo = o.get_state()
    print(b, "oracle not_conv", s.not_converged, "div", s.diverged, "iters", s.newton_iters, "gpu iters", it[b], "relq %.2e" % (np.linalg.norm(qg[b]-qo)/np.linalg.norm(qo)), "%.1fs" % (time.time()-t))
