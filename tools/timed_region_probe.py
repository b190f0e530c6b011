"""What the contract's timed region costs beyond its kernel: bench.py's sequence (counter reset, W synchronous warm-up steps, device sync,
the K-step launch, wait, torch.cuda.synchronize) six times in one process - the FIRST pass pays 13 - 30 us of one-time set-up, which is why
bench.py rehearses the sequence once, untimed, before the real one.  GPU box only."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
torch.cuda.set_device(0); torch.cuda.synchronize()
import bench
scene, h, integ, gen = bench.build_workload("chain", 32)
st = bench.GpuStepper(scene, 1024, 0, integ); st.set_opts(h, 1e-9, 1)
q0, qd0 = gen(0, 1024); st.set_state(q0, qd0)
for _ in range(80):
    st.set_state(q0, qd0); st.launch(20); st.wait()
sim = st.sim
pc = time.perf_counter
def timed():
    torch.cuda.synchronize()
    t0 = pc(); st.launch(20); ms = st.wait(); torch.cuda.synchronize(); t1 = pc()
    return (t1 - t0) * 1e6 - ms * 1e3, ms * 1e3
def seq(mode):
    st.set_state(q0, qd0)
    st.stats_reset()
    if mode == "sync_api": st.warmup(5)
    elif mode == "async_api": sim.step_bdf1_async(5); sim.sync()
    elif mode == "sync_api+primer": st.warmup(4); sim.step_bdf1_async(1); sim.sync()
    elif mode == "no_stats_reset":
        pass
    if mode == "no_stats_reset": st.warmup(5)
    st.sync_device()
    return timed()
for k in range(6):
    r = seq("sync_api")
    print("sequence %d: overhead %.1f us, kernel %.1f us" % (k, r[0], r[1]))
