"""Measurement build only (RMX_EXTRA_HIPCC_FLAGS=-DRMX_COOP_PROFILE): share of the cooperative launch its members spend inside
coop_exchange (config 5, 1024 x 100).  Usage: coop_profile.py out.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1]
os.environ["RMX_COOP_DUMP"] = out + ".dump"
import numpy as np
from tools.coop_check import run   # noqa: E402

r = run(1024, 100, 24, reps=1)
txt = open(out + ".dump").read()
lines = [l for l in txt.split("\n") if l.startswith("group")]
with open(out, "w") as f:
    f.write("launch %.2f ms\n" % r["ms"])
    for l in lines:
        p = l.split()
        roll, nx = int(p[3]), int(p[5])
        w = np.array([int(x) for x in p[7:]], dtype=float) * 1024 / 2.4e3      # us
        f.write("rollout %4d exchanges %5d iters %4d halvings %5d total %.2f ms | wait per member: min %.0f max %.0f us  = %.2f .. %.2f us per exchange\n" % (
            roll, nx, r["it"][roll], r["ls"][roll], r["tk"][roll] / 2.4e6, w.min(), w.max(), w.min() / max(nx, 1), w.max() / max(nx, 1)))
print(open(out).read())
