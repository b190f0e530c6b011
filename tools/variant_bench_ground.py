"""Development aid: the config-5 launch (32-link chain over frictional ground, BDF2, 1024 rollouts, 10 + 100 steps as bench.py --workload
ground runs them) for the in-tree library and every redmax_amd/variants/libredmax_hip_*.so, one subprocess per library: kernel ms,
Newton iterations, line-search halvings, rollouts with a failed step, and the final-state difference against the in-tree library.
    python tools/variant_bench_ground.py [reps]"""
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
from redmax_amd import BatchSim, sceneChainGround, syntheticStates
R = int(sys.argv[2])
sc = sceneChainGround(32); sc.init(); B = 1024
q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1); q[0], qd[0] = sc.getQ()
sim = BatchSim(sc, batch=B)
ms = []
for r in range(R):
    sim.set_state(q, qd); sim.step_bdf2(10, h=sc.h)
    o = sim.step_bdf2(100, h=sc.h, stats=True); ms.append(o["ms"])
qf, _ = sim.get_state()
np.save(sys.argv[3], qf)
print("%%.3f %%.3f %%d %%d %%d" %% (min(ms), float(np.median(ms)), int(o["newton_iters"].sum()), int(o["ls_halvings"].sum()), int(((o["status"] & 15) != 0).sum())))
''' % ROOT


def main():
    R = sys.argv[1] if len(sys.argv) > 1 else "3"
    libs = ["-"] + sorted(glob.glob(os.path.join(ROOT, "redmax_amd", "variants", "libredmax_hip_*.so")))
    import numpy as np
    ref = None
    for lib in libs:
        out = "/tmp/vbg_%d.npy" % os.getpid()
        p = subprocess.run([sys.executable, "-c", CHILD, lib, R, out], capture_output=True, text=True)
        name = "in-tree" if lib == "-" else os.path.basename(lib)[len("libredmax_hip_"):-3]
        if p.returncode != 0:
            print("%-28s FAILED: %s" % (name, p.stderr.strip().splitlines()[-1] if p.stderr.strip() else "?"))
            continue
        mn, med, it, hv, bad = p.stdout.split()
        qf = np.load(out)
        if ref is None:
            ref = qf
        print("%-28s kernel ms/100 steps min %s median %s (%.2f M rollout-steps/s)  iters %s halvings %s bad %s  max|q - q_intree| %.2e" % (
            name, mn, med, 1024 * 100 / float(med) / 1e3, it, hv, bad, np.abs(qf - ref).max()), flush=True)


main()
