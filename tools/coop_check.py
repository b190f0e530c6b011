"""Park and relaunch (rmx_device.h CoopCtx) against one wavefront per rollout throughout: config 5 (32-link chain over frictional
ground, BDF2), same launch with RMX_PARK_HALVINGS = 0 and with the default.  Final states must agree bit for bit, the per-rollout
Newton iteration / halving counts and status words must be equal.  Usage: coop_check.py [B] [K] [park_halvings ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench


def run(B, K, park, reps=1):
    os.environ["RMX_PARK_HALVINGS"] = str(park)
    scene, h, integ, gen = bench.build_workload("ground", 32)
    st = bench.GpuStepper(scene, B, 0, integ)
    st.set_opts(h, 1e-9, 1)
    q0, qd0 = gen(0, B)
    best = None
    for _ in range(reps):
        st.set_state(q0, qd0)
        st.stats_reset()
        st.launch(K)
        ms = st.wait()
        best = ms if best is None else min(best, ms)
    s = st.stats()
    q, qd = st.get_state()
    tk = st.rollout_ticks().astype(np.float64)
    return dict(ms=best, q=q, qd=qd, it=s["newton_iters"].copy(), ls=s["ls_halvings"].copy(), st=s["status"].copy(), tk=tk)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    parks = [int(a) for a in sys.argv[3:]] or [24]
    ref = run(B, K, 0, reps=2)
    print("one wavefront per rollout: %.2f ms, iterations %d, halvings %d, flagged %d" % (ref["ms"], ref["it"].sum(), ref["ls"].sum(), ((ref["st"] & 15) != 0).sum()))
    for p in parks:
        r = run(B, K, p, reps=2)
        same_q = np.array_equal(r["q"], ref["q"]) and np.array_equal(r["qd"], ref["qd"])
        print("park at %3d halvings: %.2f ms (%.2fx), states bit-identical %s, iterations equal %s, halvings equal %s, status equal %s, faults %d" % (
            p, r["ms"], ref["ms"] / r["ms"], same_q, np.array_equal(r["it"], ref["it"]), np.array_equal(r["ls"], ref["ls"]),
            np.array_equal(r["st"], ref["st"]), ((r["st"] & 512) != 0).sum()))
        if not same_q:
            bad = np.nonzero((r["q"] != ref["q"]).any(1))[0]
            print("   differing rollouts", len(bad), bad[:10], "max |dq|", np.abs(r["q"] - ref["q"]).max())
            print("   iters", r["it"][bad[:6]], ref["it"][bad[:6]], "halvings", r["ls"][bad[:6]], ref["ls"][bad[:6]], "status", r["st"][bad[:6]], ref["st"][bad[:6]])


if __name__ == "__main__":
    main()
