"""Where the time of BASELINE.json configs[4] goes (32-link chain over frictional ground, BDF2, 1024 rollouts).  The launch
ends with its slowest rollout; this replays single rollouts as a whole batch (1024 copies: launch time = that rollout's time)
and reads the cost of a Newton iteration and of a line-search trial point off them."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneChainGround, syntheticStates  # noqa: E402

B, K = 1024, 100
sc = sceneChainGround(32)
sc.init()
q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1)
q[0], qd[0] = sc.getQ()
sim = BatchSim(sc, batch=B)
sim.opts.tol = 1e-8
sim.set_state(q, qd)
out = sim.step_bdf2(K, h=sc.h, stats=True)
it, ls, st = out["newton_iters"], out["ls_halvings"], out["status"]
print("all: %.2f ms; iters mean %.1f max %d; halvings mean %.1f max %d; maxiter %d" % (out["ms"], it.mean(), it.max(), ls.mean(), ls.max(), int(((st & 2) != 0).sum())))
order = np.argsort(it + 0.3 * ls)
rows = []
for name, idx in (("slowest", order[-1]), ("2nd", order[-2]), ("p90", order[int(0.9 * B)]), ("median", order[B // 2]), ("fastest", order[0]), ("most halvings", int(np.argmax(ls)))):
    sim.stats_reset()
    sim.set_state(np.repeat(q[idx:idx + 1], B, 0), np.repeat(qd[idx:idx + 1], B, 0))
    o = sim.step_bdf2(K, h=sc.h, stats=True)
    rows.append((o["newton_iters"][0], o["ls_halvings"][0], o["ms"]))
    print("%-14s rollout %4d: %7.2f ms  iters %4d  halvings %5d  status %d" % (name, idx, o["ms"], o["newton_iters"][0], o["ls_halvings"][0], o["status"][0]))
clean = [r for r in rows if r[1] == 0 and r[2] > 3.0]      # no halvings, not a pure free-flight rollout
slow = max(rows, key=lambda r: r[1])
if clean:
    per_it = np.mean([r[2] / r[0] for r in clean]) * 1e3
    print("per Newton iteration ((g,H) + solve + first trial point) on rollouts without halvings: %.1f us;  per extra trial point "
          "(one residual evaluation) on the rollout with the most halvings: %.2f us" % (per_it, (slow[2] * 1e3 - slow[0] * per_it) / max(slow[1], 1)))
sim.close()
