"""Why is the contract's timed launch (first one after warm-up + sync) a few % slower than its repeats?  Times the same K-step
launch after different preludes.  Diagnosis only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

def main():
    K, W, B = 100, 10, 1024
    scene, h, _, gen = bench.build_workload("chain", 32)
    st = bench.GpuStepper(scene, B, 0)
    st.set_opts(h, 1e-9, 1)
    q0, qd0 = gen(0, B)
    st.set_state(q0, qd0); st.warmup(W); qw, qdw = st.get_state()
    def timed(prelude):
        prelude()
        st.launch(K)
        return st.wait()
    def p_sync():                       # the contract's: warm-up kernel, sync, launch
        st.set_state(q0, qd0); st.warmup(W); st.sync_device()
    def p_copy():                       # the repeats': H2D copy of the post-warm-up state, launch
        st.set_state(qw, qdw)
    def p_copy_sync():
        st.set_state(qw, qdw); st.sync_device()
    def p_copy_sleep():
        st.set_state(qw, qdw); st.sync_device(); time.sleep(0.05)
    def p_sync_copy():                  # warm-up, sync, then a redundant copy of the same state
        st.set_state(q0, qd0); st.warmup(W); st.sync_device(); a, b = st.get_state(); st.set_state(a, b)
    def p_sync_sleep():
        st.set_state(q0, qd0); st.warmup(W); st.sync_device(); time.sleep(0.05)
    def p_reset_sync():
        st.set_state(q0, qd0); st.stats_reset(); st.warmup(W); st.sync_device()
    def p_reset_copy():
        st.stats_reset(); st.set_state(qw, qdw)
    def work():
        s = st.stats(); return int(s["newton_iters"].sum()), int(st.rollout_ticks().max())
    for name, p in [("reset+warmup+sync", p_reset_sync), ("reset+copy", p_reset_copy), ("reset+warmup+sync", p_reset_sync), ("reset+copy", p_reset_copy)]:
        ts = []
        for _ in range(4):
            t = timed(p); ts.append((t,) + work())
        print("%-24s %s" % (name, " ".join("%.4f/%d/%d" % t for t in ts)))
    for name, p in [("warmup+sync", p_sync), ("copy", p_copy), ("copy+sync", p_copy_sync), ("copy+sync+sleep50ms", p_copy_sleep),
                    ("warmup+sync+copy", p_sync_copy), ("warmup+sync+sleep50ms", p_sync_sleep), ("warmup+sync", p_sync), ("copy", p_copy)]:
        ts = [timed(p) for _ in range(6)]
        print("%-24s %s" % (name, " ".join("%.4f" % t for t in ts)))
main()
