"""Per-phase shader-clock cycles of one Newton iteration for the benchmark workload (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from redmax_amd import BatchSim, sceneChain, sceneTree, syntheticStates  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32          # negative: the |n|-joint tree of BASELINE.json configs[2]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sc = sceneChain(n) if n > 0 else sceneTree(-n)
sc.init()
q, qd = syntheticStates(sc.nr, B)
if n < 0:
    q = q * 0.5 + sc.getQ()[0]
sim = BatchSim(sc, batch=B)
sim.set_state(q, qd)
sim.step_bdf1(10, h=1e-2)
c = sim.profile_phases(reps=20)
tot = c["eval_g"] + c["eval_gH"] + c["lu"] + c["reductions"]
print("cycles/wave: " + "  ".join("%s=%.0f" % (k, c[k]) for k in ("eval_g", "eval_gH", "lu", "reductions")) + "  sum=%.0f (%.1f us @2.4GHz)" % (tot, tot / 2400.0))
print("  (g,H) stages: " + "  ".join("%s=%.0f" % kv for kv in c["gH_stamps"].items()))
