#!/bin/bash
# round-3 session A: compensated iterate - tests, reference-tol statistics, quick timings
OUT=gpurun_out/r03a; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python tools/reference_tol_stats.py --rollouts 128 --steps 100 --json $OUT/reference_tol_stats.json > $OUT/reference_tol_stats.txt 2>&1; tail -20 $OUT/reference_tol_stats.txt
python - > $OUT/quick.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from redmax_amd import BatchSim, sceneChain, syntheticStates
sc = sceneChain(32); sc.init()
q, qd = syntheticStates(32, 1024)
for tol in (1e-8, 1e-9):
    for comp in (1, 0):
        sim = BatchSim(sc, batch=1024); sim.opts.tol = tol; sim.opts.compensated = comp
        ms = []
        for r in range(5):
            sim.set_state(q, qd); sim.step_bdf1(10, h=1e-2)
            o = sim.step_bdf1(100, h=1e-2, stats=True); ms.append(o["ms"])
        print("tol %g compensated %d: kernel ms/100 steps min %.3f median %.3f -> %.2f M rollout-steps/s; it/step %.3f halv/step %.3f bad rollouts %d" % (
            tol, comp, min(ms), np.median(ms), 102.4/np.median(ms), o["newton_iters"].sum()/102400, o["ls_halvings"].sum()/102400, ((o["status"]&15)!=0).sum()))
        sim.close()
PY
cat $OUT/quick.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log
