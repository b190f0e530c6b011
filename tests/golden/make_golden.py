"""Generates tests/golden/oracle_vectors.npz: trajectory / residual vectors produced by the CPU oracle AFTER it passed the
reference's known-answer energies (tests/test_oracle_kat.py).  The reference (MATLAB) holds no per-step vectors and cannot
be run here, so these are oracle outputs, frozen so that (i) the oracle cannot drift silently and (ii) the GPU tests have
a data-only check that does not need the oracle library.  Re-run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402
from redmax_amd.scenes import sceneChain, sceneChainGround, scenesRedMax, sceneTree, syntheticStates  # noqa: E402


def main():
    out = {}
    for sid in (0, 1, 2, 3, 14):
        sc = scenesRedMax(sid)
        sc.init()
        o = Oracle(sc.desc())
        _, V0 = o.energy()
        done = 0
        for k in (1, 10, sc.nsteps):
            st, T, V = o.step_bdf1(sc.h, k - done, history=True)
            done = k
            q, qd = o.get_state()
            out["scene%d_bdf1_step%d_q" % (sid, k)] = q
            out["scene%d_bdf1_step%d_qdot" % (sid, k)] = qd
            out["scene%d_bdf1_step%d_TV" % (sid, k)] = np.array([T[-1], V[-1], V0])
    # 32-chain (config 2): residual and Hessian at a seeded state, short rollouts from the synthetic states
    sc = sceneChain(32)
    sc.init()
    o = Oracle(sc.desc())
    rng = np.random.default_rng(424242)
    q0 = rng.uniform(-0.3, 0.3, 32)
    qd0 = rng.uniform(-0.5, 0.5, 32)
    q1 = q0 + 1e-2 * qd0 + rng.uniform(-1e-3, 1e-3, 32)
    g, H = o.eval_bdf1(q1, q0, qd0, 1e-2)
    out["chain32_eval_inputs"] = np.stack([q1, q0, qd0])
    out["chain32_eval_g"] = g
    out["chain32_eval_H"] = H
    q, qd = syntheticStates(32, 4)
    for b in range(4):
        o = Oracle(sc.desc())
        o.set_state(q[b], qd[b])
        o.step_bdf1(1e-2, 1)
        out["chain32_traj%d_step1_q" % b] = o.get_state()[0]
        o.step_bdf1(1e-2, 9)
        out["chain32_traj%d_step10_q" % b] = o.get_state()[0]
    # branching prismatic/revolute tree
    sc = sceneTree(15)
    sc.init()
    o = Oracle(sc.desc())
    o.step_bdf1(sc.h, 10)
    out["tree15_bdf1_step10_q"], out["tree15_bdf1_step10_qdot"] = o.get_state()
    # multi-DOF joints (scenes 4, 5, 6, 8), Euler-chart joints (7, 9: BDF2, scene 7 switches charts) and ground contact (11)
    for sid in (4, 5, 6, 8):
        sc = scenesRedMax(sid)
        sc.init()
        o = Oracle(sc.desc())
        o.step_bdf1(sc.h, 10)
        out["scene%d_bdf1_step10_q" % sid], out["scene%d_bdf1_step10_qdot" % sid] = o.get_state()
    for sid in (7, 9, 11):
        sc = scenesRedMax(sid)
        sc.init()
        o = Oracle(sc.desc())
        n = sc.nsteps if sid != 11 else 400
        o.step_bdf2(sc.h, n)
        out["scene%d_bdf2_end_q" % sid], out["scene%d_bdf2_end_qdot" % sid] = o.get_state()
        if sid != 11:
            out["scene%d_bdf2_end_charts" % sid] = o.charts()
    sc = sceneChainGround(6, ground_z=-1.0)
    sc.init()
    o = Oracle(sc.desc())
    rng = np.random.default_rng(77)
    q0 = rng.uniform(-0.4, 0.4, 6)
    qd0 = rng.normal(size=6) * 3
    q1 = q0 + sc.h * qd0
    g, H = o.eval_bdf1(q1, q0, qd0, sc.h)
    out["chain6ground_eval_inputs"] = np.stack([q1, q0, qd0])
    out["chain6ground_eval_g"] = g
    out["chain6ground_eval_H"] = H
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz"), **out)
    print("wrote %d arrays" % len(out))


if __name__ == "__main__":
    main()
