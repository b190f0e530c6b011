import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from tools.coop_check import run
ref = run(1024, 100, 0)
r = run(1024, 100, 24)
f = np.nonzero(r["st"] & 512)[0]
print("faulted", f)
for i in f[:10]:
    print(i, "it", r["it"][i], ref["it"][i], "ls", r["ls"][i], ref["ls"][i], "st", r["st"][i], ref["st"][i])
d = np.nonzero((r["it"] != ref["it"]))[0]
print("differ", d[:20])
