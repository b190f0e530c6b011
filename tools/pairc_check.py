"""The two-point kernel of the full 32-link chain (rmx_pair32.h, RMX_PART 7: every front carries the next step's first point beside the
trial) against the one-point headline kernel (RMX_PAIRC=0): kernel time per K steps, bit-equality of states / Newton counts / halvings /
status / per-step energies, on the bench states, on wild states (line searches, diverging rollouts), with lu_mode 1 and with
compensated = 0.  RMX_PAIRC is read at every step call.  Usage: pairc_check.py [batch ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneChain, syntheticStates  # noqa: E402


def run(B, pair, K=100, wild=False, lu_mode=0, comp=1, tol=1e-9, reps=3, hist=False):
    os.environ["RMX_PAIRC"] = "1" if pair else "0"
    os.environ["RMX_W2_MAX"] = "0"            # (the one-point kernel, not the two-wave one, is the reference)
    sc = sceneChain(32)
    sc.init()
    q, qd = syntheticStates(sc.nr, B, sq=0.6, sv=4.0) if wild else syntheticStates(sc.nr, B)
    sim = BatchSim(sc, batch=B)
    sim.opts.tol = tol
    sim.opts.lu_mode = lu_mode
    sim.opts.compensated = comp
    sim.set_state(q, qd)
    sim.step_bdf1(5, h=1e-2)
    q0, qd0 = sim.get_state()
    best, out = None, None
    for _ in range(reps):
        sim.set_state(q0, qd0)
        out = sim.step_bdf1(K, h=1e-2, stats=True, history=hist)
        best = out["ms"] if best is None else min(best, out["ms"])
    qf, qdf = sim.get_state()
    sim.close()
    return best, qf, qdf, out


def same(a, b):
    ok = np.array_equal(a[1], b[1], equal_nan=True) and np.array_equal(a[2], b[2], equal_nan=True)
    for k in ("newton_iters", "ls_halvings", "status"):
        ok = ok and np.array_equal(a[3][k], b[3][k])
    for k in ("T", "V"):
        if k in a[3] and a[3][k] is not None:
            ok = ok and np.array_equal(a[3][k], b[3][k], equal_nan=True)
    return ok


def main():
    for B in [int(a) for a in sys.argv[1:]] or [1024, 128]:
        for K in (100, 20):
            one, two = run(B, False, K), run(B, True, K)
            print("chain32 B=%d K=%d: one point %.4f ms, two points %.4f ms (x%.3f), same bits %s, max|dq| %.2e, iterations %d vs %d"
                  % (B, K, one[0], two[0], one[0] / two[0], same(one, two), np.nanmax(np.abs(one[1] - two[1])), one[3]["newton_iters"].sum(),
                     two[3]["newton_iters"].sum()), flush=True)
    B = 256
    for name, kw in (("history", dict(hist=True, K=12)), ("wild tol 1e-6", dict(wild=True, tol=1e-6, K=12, hist=True)), ("lu_mode 1", dict(lu_mode=1, K=12)),
                     ("compensated 0", dict(comp=0, K=6)), ("wild lu_mode 1", dict(wild=True, tol=1e-6, lu_mode=1, K=6))):
        one, two = run(B, False, reps=1, **kw), run(B, True, reps=1, **kw)
        print("%-16s B=%d: same bits %s (max|dq| %.2e; halvings %d, status!=0 on %d rollouts, pivoted %d)" %
              (name, B, same(one, two), np.nanmax(np.abs(one[1] - two[1])), one[3]["ls_halvings"].sum(), int((one[3]["status"] != 0).sum()),
               int((one[3]["status"] & 16 != 0).sum())), flush=True)


if __name__ == "__main__":
    main()
