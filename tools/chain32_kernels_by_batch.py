import os, subprocess, sys
ROOT="/root/repo"
CHILD=r'''
import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from redmax_amd import BatchSim, sceneChain, syntheticStates
B, K, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sc = sceneChain(32); sc.init(); q, qd = syntheticStates(32, B)
sim = BatchSim(sc, batch=B)
sim.set_state(q, qd); sim.step_bdf1(5, h=1e-2)
q0, qd0 = sim.get_state()
ms=[]
for r in range(R):
    sim.set_state(q0, qd0)
    o = sim.step_bdf1(K, h=1e-2, stats=True)
    ms.append(o["ms"])
print("%s %.4f %.4f" % (sim.last_step_kernel(), min(ms), float(np.median(ms))))
'''
for B in (64, 128, 256, 512, 1024):
    row=[]
    for name, env in (("one-point", {"RMX_PAIRC":"0","RMX_W2_MAX":"0"}), ("two-wave", {"RMX_PAIRC":"0"}), ("two-point", {})):
        p=subprocess.run([sys.executable,"-c",CHILD,str(B),"100","7"],capture_output=True,text=True,env=dict(os.environ,**env))
        row.append("%s: %s" % (name, p.stdout.strip() or p.stderr.strip()[-200:]))
    print("B=%d  " % B + " | ".join(row), flush=True)
