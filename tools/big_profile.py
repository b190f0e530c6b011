"""Development aid: in-kernel phase timers of the large-tree kernels (rmx_big.hip built with -DRMX_BIG_PROFILE, linked as
redmax_amd/variants/libredmax_hip_bigprof.so): s_memtime ticks of block 0 in the evaluations, the solve and its parts, per launch.
    python tools/big_profile.py [links ...]

Building the variant (after __graft_entry__.build(), which leaves the other objects in build/):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iredmax_amd/csrc -DRMX_BIG_PROFILE -c -o build/variants/rmx_big_bigprof.o redmax_amd/csrc/rmx_big.hip
    hipcc --offload-arch=gfx950 -fPIC -shared -o redmax_amd/variants/libredmax_hip_bigprof.so $(ls build/*.o | grep -v rmx_big.o) build/variants/rmx_big_bigprof.o
RMX_PROFILE_LIB=<name> picks another redmax_amd/variants/libredmax_hip_<name>.so."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
from redmax_amd import _abi
_abi.LIB_PATH = os.path.join(%r, "redmax_amd", "variants", "libredmax_hip_%%s.so" %% os.environ.get("RMX_PROFILE_LIB", "bigprof"))
from redmax_amd import BatchSim, sceneChain, syntheticStates
n, B, tol, steps = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
sc = sceneChain(n); sc.init()
q, qd = syntheticStates(sc.nr, B)
sim = BatchSim(sc, batch=B)
sim.opts.tol = tol
sim.set_state(q, qd)
o = sim.step_bdf1(steps, h=1e-2, stats=True)
it = o["newton_iters"]
print("chain %%d B=%%d tol %%g: %%.3f ms per step; Newton iterations per step: rollout 0 %%.2f, mean %%.2f, max %%.2f; halvings rollout 0: %%d; bad %%d; pivoted %%d" %% (
    n, B, tol, o["ms"] / steps, it[0] / steps, it.mean() / steps, it.max() / steps, int(o["ls_halvings"][0]), int(((o["status"] & 15) != 0).sum()), int(((o["status"] & 16) != 0).sum())), flush=True)
sim.close()
''' % (ROOT, ROOT)

for n, tol in ((72, 1e-8), (128, 1e-7), (256, 1e-6)):
    if len(sys.argv) > 1 and str(n) not in sys.argv[1:]:
        continue
    for B in (256,):
        p = subprocess.run([sys.executable, "-c", CHILD, str(n), str(B), str(tol), "10"], capture_output=True, text=True)
        print(p.stdout.strip())
        if p.returncode != 0:
            print(p.stderr.strip()[-600:])
