export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r5w; mkdir -p $O
cat > /tmp/b512.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from redmax_amd import _abi
if os.environ.get("RMX_W2_LIB"): _abi.LIB_PATH = os.environ["RMX_W2_LIB"]
from redmax_amd import BatchSim, sceneChain, syntheticStates
sc = sceneChain(32); sc.init()
for B in (128, 512):
    q, qd = syntheticStates(sc.nr, B)      # the bench states: trajectory 0 is q = 0.1, qdot = 0
    sim = BatchSim(sc, batch=B); ms = []
    for r in range(5):
        sim.set_state(q, qd); sim.step_bdf1(10, h=1e-2)
        ms.append(sim.step_bdf1(100, h=1e-2, stats=True)["ms"])
    print("  B=%d: %.3f ms per 100 steps (min of 5)" % (B, min(ms)))
PY
echo "one wave"; RMX_W2_MAX=0 python /tmp/b512.py
echo "iter == predict (in-tree)"; python /tmp/b512.py
for v in 1 2; do echo "RMX_W2_PRED=$v"; RMX_W2_LIB=$PWD/redmax_amd/variants/libredmax_hip_w2cp$v.so python /tmp/b512.py; done
