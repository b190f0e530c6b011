"""configs[2] (64-joint branching tree, 512 rollouts, BDF1) and relatives: the multifrontal solve along the tree (tree_solve64, the default
where a tree of 33..64 nodes is at most 7 levels deep) against the dense block-column solve (RMX_TREE_SOLVE=0, read at model creation):
final states (agree to roundoff, not bit for bit: another elimination order), Newton counts, status words, kernel milliseconds.
    python tools/tree_solve_check.py [B] [K]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
from redmax_amd import BatchSim, sceneTree
n, B, K, integ, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
sc = sceneTree(n); sc.init(); qs, _ = sc.getQ()
q, qd = np.empty((B, sc.nr)), np.empty((B, sc.nr))
for i in range(B):
    rng = np.random.default_rng(20240 + i)
    q[i] = qs + rng.uniform(-0.05, 0.05, sc.nr); qd[i] = rng.uniform(-0.1, 0.1, sc.nr)
sim = BatchSim(sc, batch=B)
step = sim.step_bdf2 if integ == "bdf2" else sim.step_bdf1
ms = []
for r in range(5):
    sim.set_state(q, qd); step(5, h=1e-2)
    o = step(K, h=1e-2, stats=True); ms.append(o["ms"])
qf, qdf = sim.get_state()
np.savez(out, q=qf, qd=qdf, it=o["newton_iters"], ls=o["ls_halvings"], st=o["status"], ms=np.array(ms), kern=np.array([sim.last_step_kernel()]))
''' % ROOT


def run(n, B, K, integ, tree):
    out = "/tmp/tsc_%d_%d.npz" % (os.getpid(), tree)
    env = dict(os.environ, RMX_TREE_SOLVE="1" if tree else "0")
    p = subprocess.run([sys.executable, "-c", CHILD, str(n), str(B), str(K), integ, out], capture_output=True, text=True, env=env, timeout=600)
    if p.returncode != 0:
        raise SystemExit("child failed: " + p.stderr[-800:])
    import numpy as np
    return dict(np.load(out))


def main():
    import numpy as np
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    for n, b, integ in ((64, B, "bdf1"), (64, B, "bdf2"), (64, 2 * B, "bdf1"), (48, B, "bdf1"), (64, 7, "bdf1")):
        d, t = run(n, b, K, integ, 0), run(n, b, K, integ, 1)
        dq = np.abs(t["q"] - d["q"]).max() / np.abs(d["q"]).max()
        print("tree%d B=%d %s [%s]: dense min %.3f ms, along the tree min %.3f ms (x%.3f); max|dq|/max|q| %.2e; Newton iterations %d vs %d (%d rollouts differ), "
              "halvings %d vs %d, status != 0 on %d vs %d rollouts" %
              (n, b, integ, str(t["kern"][0]), d["ms"].min(), t["ms"].min(), d["ms"].min() / t["ms"].min(), dq, d["it"].sum(), t["it"].sum(), int((d["it"] != t["it"]).sum()),
               d["ls"].sum(), t["ls"].sum(), int((d["st"] != 0).sum()), int((t["st"] != 0).sum())), flush=True)


if __name__ == "__main__":
    main()
