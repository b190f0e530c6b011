"""Config 5 (32-link chain over frictional ground, BDF2, 1024 x 100) on the in-tree library and every redmax_amd/variants/libredmax_hip_*.so:
kernel ms one wavefront per rollout (RMX_PARK_HALVINGS=0) and with park and relaunch, Newton iterations / halvings, and the
slowest rollout's rmx_step_ticks (variants built with -DRMX_TICK_PHASE=k report the ticks of phase k there).  One subprocess per library."""
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
sc = sceneChainGround(32); sc.init(); B = 1024
q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1); q[0], qd[0] = sc.getQ()
sim = BatchSim(sc, batch=B)
res = []
for park, fused in (("0", "2"), ("24", "1"), ("24", "2")):
    os.environ["RMX_PARK_HALVINGS"] = park
    os.environ["RMX_GROUND_FUSED"] = fused
    ms = []
    for r in range(2):
        sim.set_state(q, qd)
        o = sim.step_bdf2(100, h=sc.h, stats=True); ms.append(o["ms"])
    tk = sim.step_ticks().astype(float)
    qf, _ = sim.get_state()
    i = int(np.argmax(o["newton_iters"]))
    res.append("park %%2s fused %%s: %%.2f ms, iters %%d halv %%d bad %%d | ticks/2.4e3: max %%.0f us, p50 %%.0f us, rollout %%d (most iterations: %%d) %%.0f us, sum over parked-like (top 48) %%.0f us" %% (
        park, fused, min(ms), o["newton_iters"].sum(), o["ls_halvings"].sum(), ((o["status"] & 15) != 0).sum(), tk.max() / 2.4e3, np.median(tk) / 2.4e3, i, o["newton_iters"][i], tk[i] / 2.4e3,
        np.sort(tk)[-48:].mean() / 2.4e3))
    np.save(sys.argv[2] + park + fused + ".npy", qf)
print("\n".join(res))
''' % ROOT


def main():
    import numpy as np
    libs = ["-"] + sorted(glob.glob(os.path.join(ROOT, "redmax_amd", "variants", "libredmax_hip_*.so")))
    ref = None
    for lib in libs:
        out = "/tmp/pb_%d_" % os.getpid()
        p = subprocess.run([sys.executable, "-c", CHILD, lib, out], capture_output=True, text=True)
        name = "in-tree" if lib == "-" else os.path.basename(lib)[len("libredmax_hip_"):-3]
        if p.returncode != 0:
            print("%-12s FAILED: %s" % (name, p.stderr.strip()[-300:]))
            continue
        q0, q24, q24f = np.load(out + "02.npy"), np.load(out + "241.npy"), np.load(out + "242.npy")
        if ref is None:
            ref = q0
        print("== %-12s park == no-park: %s, fused == no-park: %s, == in-tree: %s" % (name, np.array_equal(q0, q24), np.array_equal(q0, q24f), np.array_equal(q0, ref)))
        print(p.stdout.strip(), flush=True)


main()
