"""GPU parity for the multi-DOF joints: JointPlanar, JointTranslational, JointUniversal, JointFree2D are
lowered by rmx_model_create to chains of 1-DOF nodes with massless links that keep the reference's reduced numbering.

Tolerances (fp64): single evaluation |dg|/|g|, |dH|_F/|H|_F <= 1e-11 vs the oracle (whose own restatement of these joints is
pinned by the goldens, tests/test_oracle_kat.py); goldens through driverRedMaxBDF1/2: 1e-9 relative (reference: 1e-2 abs).
"""
import numpy as np
import pytest

from redmax_amd.scenes import COMPOSITE_SCENES, scenesRedMax

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("sid", COMPOSITE_SCENES)
def test_eval_matches_oracle(oracle_lib, sid):
    from redmax_amd import BatchSim
    sc = scenesRedMax(sid)
    sc.init()
    B = 4
    rng = np.random.default_rng(3)
    nr, h = sc.nr, sc.h
    q0 = rng.uniform(-0.7, 0.7, (B, nr))
    qd0 = rng.uniform(-1, 1, (B, nr))
    q1 = q0 + h * qd0 + rng.uniform(-1e-2, 1e-2, (B, nr))
    sim = BatchSim(sc, batch=B)
    assert sim.nr == nr and sim.nm == sc.nm
    assert list(sim.idxR()) == [j.idxR[0] if j.ndof else -1 for j in sc.joints]
    o = oracle_lib.Oracle(sc.desc())
    for eta, qA, qB in ((h, q0, q0 + h * qd0), (2 * h / 3, q0 + 1e-3 * rng.normal(size=(B, nr)), q0 + 0.9 * h * qd0)):
        g, H = sim.eval_residual(q1, qA, qB, eta)
        for b in range(B):
            go, Ho = o.eval_residual(q1[b], qA[b], qB[b], eta)
            assert _rel(g[b], go) <= 1e-11
            assert _rel(H[b], Ho) <= 1e-11
    sim.set_state(q1, qd0)
    T, V = sim.energy()
    for b in range(B):
        o.set_state(q1[b], qd0[b])
        To, Vo = o.energy()
        assert abs(T[b] - To) <= 1e-11 * max(abs(To), 1) and abs(V[b] - Vo) <= 1e-11 * max(abs(Vo), 1)
    sim.close()


@pytest.mark.parametrize("sid", COMPOSITE_SCENES)
def test_goldens_through_the_drivers(sid):
    from redmax_amd import driverRedMaxBDF1, driverRedMaxBDF2
    for drv, k in ((driverRedMaxBDF1, 0), (driverRedMaxBDF2, 1)):
        sc, H, passed = drv(sid, verbose=False)
        assert passed
        assert abs(H - sc.Hexpected[k]) <= 1e-9 * abs(sc.Hexpected[k])
        assert sc.solverInfo["status"] & 7 == 0


def test_universal_chain_with_joint_springs_matches_oracle(oracle_lib):
    """Stiffness/damping act on every DOF of a multi-DOF joint with qRest = initial q per DOF (Joint.m:157, 446)."""
    from redmax_amd import BatchSim
    sc = scenesRedMax(8)
    for j in sc.joints:
        j.setStiffness(2e4)
        j.setDamping(3e2)
    sc.init()
    sim = BatchSim(sc, batch=1)
    q0, qd0 = sc.getQ()
    sim.set_state(q0[None], qd0[None])
    out = sim.step_bdf1(40, h=sc.h, stats=True, history=True)
    q, qd = sim.get_state()
    o = oracle_lib.Oracle(sc.desc())
    st, To, Vo = o.step_bdf1(sc.h, 40, history=True)
    qo, qdo = o.get_state()
    assert _rel(q[0], qo) <= 1e-9 and _rel(qd[0], qdo) <= 1e-8
    assert np.abs(out["V"][:, 0] - Vo).max() <= 1e-9 * np.abs(Vo).max()
    sim.close()
