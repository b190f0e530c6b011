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


# ---------------------------------------------------------------- JointSpherical / JointFree3D (Euler charts, reparam_)

@pytest.mark.parametrize("sid", [7, 9])
def test_spherical_goldens_through_the_drivers(sid):
    """Scenes 7 and 9 (scenesRedMax.m:204-224, 248-260); scene 7 under BDF2 switches Euler charts on the device
    (JointSpherical.reparam_ :63-102) - without the switch it ends 1.2e-3 away from the golden."""
    from redmax_amd import driverRedMaxBDF1, driverRedMaxBDF2
    from redmax_amd.scenes import scenesRedMax as _s
    for drv, k in ((driverRedMaxBDF1, 0), (driverRedMaxBDF2, 1)):
        sc, H, passed = drv(sid, verbose=False)
        assert passed
        assert abs(H - sc.Hexpected[k]) <= 1e-9 * abs(sc.Hexpected[k])
        assert sc.solverInfo["status"] & 7 == 0
        assert bool(sc.solverInfo["status"] & 32) == ((sid, k) == (7, 1))        # RMX_ST_CHART
    _s(0)


def test_spherical_chart_switch_matches_oracle(oracle_lib):
    """Scene 7, BDF2, whole run with per-step history: charts, q and qdot (coordinates in the current chart) agree with the
    oracle's restatement of reparam_ step by step through both chart switches."""
    from redmax_amd import BatchSim
    sc = scenesRedMax(7)
    sc.init()
    q0, qd0 = sc.getQ()
    sim = BatchSim(sc, batch=2)
    assert sim.nsph == 2
    Q0 = np.stack([q0, q0 * 1.01])
    Qd0 = np.stack([qd0, qd0 * 0.99])
    sim.set_state(Q0, Qd0)
    assert np.all(sim.charts() == 7)
    out = sim.step_bdf2(sc.nsteps, h=sc.h, stats=True, history="full")
    charts = sim.charts()
    for b in range(2):
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(Q0[b], Qd0[b])
        worst = 0.0
        for k in range(sc.nsteps):
            o.step_bdf2(sc.h, 1, step0=k)
            qo, qdo = o.get_state()
            worst = max(worst, _rel(out["q"][k, b], qo))
            assert _rel(out["q"][k, b], qo) <= 1e-7, (b, k)
            assert _rel(out["qdot"][k, b], qdo) <= 1e-6, (b, k)
            assert list(out["charts"][k, b]) == list(o.charts()), (b, k)      # rmx_history.charts: the chart of every step
        assert list(charts[b]) == list(o.charts())
        assert list(out["charts"][-1, b]) == list(charts[b]) and list(out["charts"][0, b]) == [7, 7]
        assert len({tuple(c) for c in out["charts"][:, b]}) >= 2             # the run does switch charts
    assert list(charts[0]) == [7, 10]                                # XYZ, YXZ
    assert np.all(out["status"] & 32)
    sim.close()


@pytest.mark.parametrize("sid", [7, 9])
def test_spherical_eval_in_other_charts_matches_oracle(oracle_lib, sid):
    """rmx_eval / rmx_energy with the joints declared in non-default charts (rmx_set_charts) vs the oracle in the same charts."""
    from redmax_amd import BatchSim
    sc = scenesRedMax(sid)
    sc.init()
    nr, h = sc.nr, sc.h
    rng = np.random.default_rng(12)
    B = 4
    sim = BatchSim(sc, batch=B)
    o = oracle_lib.Oracle(sc.desc())
    for trial, charts in enumerate(([7] * sim.nsph, [1, 12][:sim.nsph], [5, 9][:sim.nsph], [10, 3][:sim.nsph])):
        q0 = rng.uniform(-0.6, 0.6, (B, nr))
        q0[:, :] += 0.9 * (np.arange(nr) % 3 == 1)                   # keep the middle angles away from 0 (proper-Euler lock)
        qd0 = rng.uniform(-1, 1, (B, nr))
        q1 = q0 + h * qd0
        sim.set_state(q1, qd0)
        sim.set_charts(charts)
        o.set_charts(charts)
        g, H = sim.eval_bdf1(q1, q0, qd0, h)
        T, V = sim.energy()
        for b in range(B):
            go, Ho = o.eval_bdf1(q1[b], q0[b], qd0[b], h)
            assert _rel(g[b], go) <= 1e-11 and _rel(H[b], Ho) <= 1e-11, (charts, b)
            o.set_state(q1[b], qd0[b])
            To, Vo = o.energy()
            assert abs(T[b] - To) <= 1e-11 * max(abs(To), 1) and abs(V[b] - Vo) <= 1e-11 * max(abs(Vo), 1)
    sim.close()


def test_spherical_is_refused_by_euler_and_adjoint():
    from redmax_amd import BatchSim, RedMaxHipError
    sc = scenesRedMax(9)
    sc.init()
    sim = BatchSim(sc, batch=1)
    with pytest.raises(RedMaxHipError):
        sim.step_euler(1, 1e-2)
    sim.close()
