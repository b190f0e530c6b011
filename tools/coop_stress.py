"""Repeated config-5 launches with park and relaunch: every launch must reproduce the first one's iteration counts, with no cooperative
group giving up (status bit 512) and no launch near the groups' 2 s timeout.  Usage: coop_stress.py [launches]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
scene, h, integ, gen = bench.build_workload("ground", 32)
st = bench.GpuStepper(scene, 1024, 0, integ)
st.set_opts(h, 1e-9, 1)
q0, qd0 = gen(0, 1024)
ref = None
worst = 0.0
for i in range(N):
    st.set_state(q0, qd0)
    st.stats_reset()
    st.launch(100)
    ms = st.wait()
    s = st.stats()
    worst = max(worst, ms)
    key = (int(s["newton_iters"].sum()), int(s["ls_halvings"].sum()), int((s["status"] & 512).sum()))
    if ref is None:
        ref = key
    if key != ref or ms > 100.0:
        print("launch %d: %.2f ms %s (first launch %s)" % (i, ms, key, ref))
print("launches %d, slowest %.2f ms, counts %s" % (N, worst, ref))
