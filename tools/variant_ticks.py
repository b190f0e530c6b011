"""Development aid: the headline launch (32-link chain, B rollouts, K steps, reference tol) in SHADER-CLOCK TICKS per rollout
(rmx_step_ticks: immune to the clock the box happens to run at, which moves kernel milliseconds by 5-10 % between boxes and launches)
for the in-tree library - one-point kernel (RMX_PAIRC=0) and two-point kernel - and every redmax_amd/variants/libredmax_hip_*.so.
    python tools/variant_ticks.py [B] [K] [reps] [chain|tree64|tree48|tree64bdf2]      (tree64: BASELINE.json configs[2] on bench.py's states; B = 512)"""
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
B, K, R = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
wl = sys.argv[5]
if wl.startswith("tree"):       # tree64, tree48 (a partly filled 64-lane tree), tree64bdf2
    sc = sceneTree(int(wl[4:6])); sc.init(); qs, _ = sc.getQ()
    q, qd = np.empty((B, sc.nr)), np.empty((B, sc.nr))
    for i in range(B):
        rng = np.random.default_rng(20240 + i)
        q[i] = qs + rng.uniform(-0.05, 0.05, sc.nr); qd[i] = rng.uniform(-0.1, 0.1, sc.nr)
else:
    sc = sceneChain(32); sc.init(); q, qd = syntheticStates(32, B)
sim = BatchSim(sc, batch=B)
step = sim.step_bdf2 if wl.endswith("bdf2") else sim.step_bdf1
sim.set_state(q, qd); step(5, h=1e-2)
q0, qd0 = sim.get_state()
mx, md, ms = [], [], []
for r in range(R):
    sim.set_state(q0, qd0)
    o = step(K, h=1e-2, stats=True)
    t = sim.step_ticks().astype(np.float64)
    mx.append(t.max()); md.append(np.median(t)); ms.append(o["ms"])
qf, _ = sim.get_state()
print("%%.0f %%.0f %%.0f %%.4f %%d %%s" %% (min(mx), float(np.median(mx)), float(np.median(md)), min(ms), int(o["newton_iters"].sum()), hex(__import__("zlib").crc32(qf.tobytes()))))
''' % ROOT


def main():
    B = sys.argv[1] if len(sys.argv) > 1 else "1024"
    K = sys.argv[2] if len(sys.argv) > 2 else "100"
    R = sys.argv[3] if len(sys.argv) > 3 else "7"
    wl = sys.argv[4] if len(sys.argv) > 4 else "chain"
    libs = ([("one point (RMX_PAIRC=0)", "-", {"RMX_PAIRC": "0", "RMX_W2_MAX": "0"})] if wl == "chain" else []) + [("in-tree", "-", {})]
    libs += [(os.path.basename(p)[len("libredmax_hip_"):-3], p, {}) for p in sorted(glob.glob(os.path.join(ROOT, "redmax_amd", "variants", "libredmax_hip_*.so")))]
    ref = None
    for name, lib, env in libs:
        p = subprocess.run([sys.executable, "-c", CHILD, lib, B, K, R, wl], capture_output=True, text=True, env=dict(os.environ, **env))
        if p.returncode != 0:
            print("%-28s FAILED: %s" % (name, p.stderr.strip().splitlines()[-1] if p.stderr.strip() else "?"))
            continue
        mxmin, mxmed, med, ms, it, h = p.stdout.split()
        if ref is None:
            ref = (float(mxmin), float(med))
        print("%-28s slowest rollout: min %s median %s ticks (x%.4f); median rollout %s ticks (x%.4f); kernel min %s ms; iters %s; state hash %s" %
              (name, mxmin, mxmed, ref[0] / float(mxmin), med, ref[1] / float(med), ms, it, h), flush=True)


main()
