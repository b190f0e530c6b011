"""s_memtime ticks per millisecond of kernel time for the bench workloads (rmx_step_ticks of the slowest rollout / HIP-event time):
the shader clock each launch actually ran at."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

def main():
    for wl, links, B in (("chain", 32, 1024), ("chain", 32, 128), ("tree64", 64, 512), ("tree64", 64, 1024), ("ground", 32, 1024), ("chain", 72, 256), ("chain", 256, 256)):
        if wl == "chain" and links > 64:
            from redmax_amd import sceneChain, syntheticStates
            scene = sceneChain(links); scene.init()
            h, integ, gen = 1e-2, "bdf1", (lambda first, count: syntheticStates(scene.nr, count, first=first))
            K, tol = 4, 1e-6
        else:
            scene, h, integ, gen = bench.build_workload(wl, links)
            K, tol = 100, 1e-9
        st = bench.GpuStepper(scene, B, 0, integ)
        st.set_opts(h, tol, 1)
        q0, qd0 = gen(0, B)
        for rep in range(3):
            st.set_state(q0, qd0)
            st.launch(K)
            ms = st.wait()
        tk = st.rollout_ticks().astype(np.float64)
        print("%-7s n=%3d B=%4d: kernel %8.3f ms, slowest rollout %.3e ticks -> %.0f ticks/ms" % (wl, links, B, ms, tk.max(), tk.max() / ms), flush=True)
        st.close()
main()
