"""Per-rollout ticks / iterations / halvings of config 5 with and without park-and-relaunch -> gpurun_out/<tag>/coop_diag.npz"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from tools.coop_check import run   # noqa


def main():
    out = sys.argv[1]
    B, K = 1024, 100
    res = {}
    for park in (0, 24):
        r = run(B, K, park, reps=2)
        for k in ("it", "ls", "st", "tk"):
            res["%s_%d" % (k, park)] = r[k]
        res["ms_%d" % park] = r["ms"]
        print(park, r["ms"])
    np.savez(out, **res)


if __name__ == "__main__":
    main()
