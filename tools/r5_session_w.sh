export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -5
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from redmax_amd import BatchSim, sceneChain, syntheticStates
for B in (256, 512):
    r = {}
    for w2 in ("0", "100000"):
        os.environ["RMX_W2_MAX"] = w2
        sc = sceneChain(64); sc.init()
        q, qd = syntheticStates(sc.nr, B)
        sim = BatchSim(sc, batch=B); sim.set_state(q, qd); sim.step_bdf1(5, h=1e-2)
        best = min(sim.step_bdf1(100, h=1e-2, stats=True)["ms"] for _ in range(3))
        r[w2] = (best, sim.get_state()[0])
    print("chain64 B=%d: one wave %.3f ms, two waves %.3f ms per 100 steps (x%.3f), same bits %s" % (B, r["0"][0], r["100000"][0], r["0"][0] / r["100000"][0], np.array_equal(r["0"][1], r["100000"][1])))
PY
