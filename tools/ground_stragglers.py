"""Config 5 (32-link chain over frictional ground, BDF2, 1024 rollouts x 100 steps): where the launch time goes.  Per-rollout
Newton iterations, line-search halvings, status and s_memtime ticks of the slowest rollouts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    scene, h, integ, gen = bench.build_workload("ground", 32)
    B = 1024
    st = bench.GpuStepper(scene, B, 0, integ)
    st.set_opts(h, 1e-9, 1)
    q0, qd0 = gen(0, B)
    st.set_state(q0, qd0)
    st.stats_reset()
    st.launch(K)
    ms = st.wait()
    s = st.stats()
    tk = st.rollout_ticks().astype(np.float64)
    it, ls, stt = s["newton_iters"], s["ls_halvings"], s["status"]
    print("kernel %.2f ms; iterations %d, halvings %d; not converged %d" % (ms, it.sum(), ls.sum(), ((stt & 15) != 0).sum()))
    order = np.argsort(-tk)
    print("rank rollout  ms     iters  halvings status")
    for r in list(range(12)) + [50, 100, 512]:
        i = order[r]
        print("%4d %6d %7.2f %6d %8d %4d" % (r, i, ms * tk[i] / tk.max(), it[i], ls[i], stt[i]))
    # crude cost model: time = a * iters + b * halvings
    A = np.stack([it, ls], 1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, ms * tk / tk.max(), rcond=None)
    print("least squares: %.2f us per Newton iteration + %.2f us per line-search halving" % (1e3 * coef[0], 1e3 * coef[1]))
main()
