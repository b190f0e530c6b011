"""Cost of the one-workgroup-per-trajectory kernels (rmx_big.hip, trees of 65..256 nodes): kernel time per BDF1 step of an n-link
chain and of 20 free bodies at a few batch sizes, next to the 32-link chain on the one-wavefront kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
from redmax_amd import _abi  # noqa: E402
if os.environ.get("RMX_BENCH_LIB"):      # a build variant: redmax_amd/variants/libredmax_hip_<name>.so
    _abi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(_abi.__file__)), "variants", "libredmax_hip_%s.so" % os.environ["RMX_BENCH_LIB"])
from redmax_amd import BatchSim, sceneChain, syntheticStates  # noqa: E402

for n, tol in ((32, 1e-9), (64, 1e-9), (72, 1e-8), (128, 1e-7), (256, 1e-6)):
    if len(sys.argv) > 1 and str(n) not in sys.argv[1:]:
        continue
    sc = sceneChain(n)
    sc.init()
    for B, amp in ((256, 0.1), (1024, 0.1)) + (((256, 0.03),) if n == 256 else ()):
        # (256 links at U(-0.1, 0.1): 3 of 256 rollouts do not converge and the launch ends with them; U(-0.03, 0.03): every rollout valid)
        q, qd = syntheticStates(sc.nr, B, sq=amp, sv=amp)
        sim = BatchSim(sc, batch=B)
        sim.opts.tol = tol
        sim.set_state(q, qd)
        sim.step_bdf1(2, h=1e-2)
        o = sim.step_bdf1(20, h=1e-2, stats=True)
        tk = sim.step_ticks().astype(np.float64)
        per_it = tk / np.maximum(o["newton_iters"], 1)
        slow = int(np.argmax(tk))
        extra = "; ticks per Newton iteration: median %.0f k, slowest-in-time rollout %d: %.0f k x %.1f iterations per step, status %d" % (
            np.median(per_it) / 1e3, slow, per_it[slow] / 1e3, o["newton_iters"][slow] / 20, int(o["status"][slow]))
        print("chain %3d  B=%4d U(+-%g) tol %g: %.3f ms per step, %.2f Newton iterations per step, %.1f k rollout-steps/s, bad %d; slowest rollout %.1f iterations per step, %d rollouts redone with pivoting" % (
            n, B, amp, tol, o["ms"] / 20, o["newton_iters"].sum() / (20 * B), B * 20 / o["ms"], int(((o["status"] & 15) != 0).sum()),
            o["newton_iters"].max() / 20, int(((o["status"] & 16) != 0).sum())) + extra, flush=True)
        sim.close()
