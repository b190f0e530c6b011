"""Development aid: the adjoint launch of BASELINE.json configs[3] (16-DOF chain, 512 rollouts, 20 steps) on the in-tree library and on
every redmax_amd/variants/libredmax_hip_*.so: kernel time of forward + backward and P of rollout 0."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from redmax_amd import _abi
if sys.argv[1] != "-": _abi.LIB_PATH = sys.argv[1]
from redmax_amd import BatchSim, sceneAdjointChain
sc = sceneAdjointChain(16); sc.init(); B = 512
p = 0.1 * np.random.default_rng(0).standard_normal((B, sc.nr))
sim = BatchSim(sc, batch=B)
q0, qd0 = sc.getQ()
ms = []
for rep in range(8):
    sim.set_state(q0[None, :], qd0[None, :])
    P, dPdp, info = sim.adjoint_bdf1(20, sc.h, dict(sc.task, t=20 * sc.h), p, stats=True)
    ms.append(info["ms"])
print("%%.4f %%.4f %%.12g %%.12g" %% (min(ms), float(np.median(ms)), P[0], np.abs(dPdp).sum()))
''' % ROOT
for lib in ["-"] + sorted(glob.glob(os.path.join(ROOT, "redmax_amd", "variants", "libredmax_hip_*.so"))):
    p = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
    name = "in-tree" if lib == "-" else os.path.basename(lib)[len("libredmax_hip_"):-3]
    print("%-20s %s" % (name, p.stdout.strip() if p.returncode == 0 else "FAILED " + p.stderr.strip()[-300:]), flush=True)
