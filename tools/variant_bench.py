"""Development aid: kernel time of the headline launch (32-link chain, 1024 rollouts, 100 steps, reference tol) for the in-tree
library and every redmax_amd/variants/libredmax_hip_*.so, one subprocess per library, plus the final-state difference against
the in-tree library (bit-identical or a stated drift).   python tools/variant_bench.py [workload: chain|tree64] [reps]"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
from redmax_amd import _abi
if sys.argv[1] != "-": _abi.LIB_PATH = sys.argv[1]
from redmax_amd import BatchSim, sceneChain, sceneTree, syntheticStates
wl, R = sys.argv[2], int(sys.argv[3])
if wl == "chain":
    sc = sceneChain(32); sc.init(); B = 1024; q, qd = syntheticStates(32, B)
else:
    sc = sceneTree(64); sc.init(); B = int(os.environ.get("RMX_VB_BATCH", "512")); q, qd = syntheticStates(sc.nr, B); q = q * 0.5 + sc.getQ()[0]
sim = BatchSim(sc, batch=B)
ms = []
for r in range(R):
    sim.set_state(q, qd); sim.step_bdf1(10, h=1e-2)
    o = sim.step_bdf1(100, h=1e-2, stats=True); ms.append(o["ms"])
qf, _ = sim.get_state()
np.save(sys.argv[4], qf)
print("%%.4f %%.4f %%.4f %%d %%d" %% (min(ms), float(np.median(ms)), max(ms), int(o["newton_iters"].sum()), int(((o["status"] & 15) != 0).sum())))
''' % ROOT


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "chain"
    R = sys.argv[2] if len(sys.argv) > 2 else "7"
    libs = ["-"] + sorted(glob.glob(os.path.join(ROOT, "redmax_amd", "variants", "libredmax_hip_*.so")))
    import numpy as np
    ref = None
    for lib in libs:
        out = "/tmp/vb_%d.npy" % os.getpid()
        p = subprocess.run([sys.executable, "-c", CHILD, lib, wl, R, out], capture_output=True, text=True)
        name = "in-tree" if lib == "-" else os.path.basename(lib)[len("libredmax_hip_"):-3]
        if p.returncode != 0:
            print("%-28s FAILED: %s" % (name, p.stderr.strip().splitlines()[-1] if p.stderr.strip() else "?"))
            continue
        mn, med, mx, it, bad = p.stdout.split()
        qf = np.load(out)
        if ref is None:
            ref = qf
        d = np.abs(qf - ref).max()
        print("%-28s kernel ms/100 steps min %s median %s max %s  iters %s bad %s  max|q - q_intree| %.2e" % (name, mn, med, mx, it, bad, d), flush=True)


main()
