"""GPU parity for ground frictional contact (ForceGroundCuboid.m:54-183, SURVEY.md §8(f)-3 / BASELINE.json configs[4]):
the HIP path through the C ABI (rmx_model_set_ground_contact + rmx_eval / rmx_step_bdf1 / rmx_step_bdf2 / rmx_energy) vs
the CPU oracle, and the reference's scene-11 goldens through driverRedMaxBDF1/2.

Tolerances (fp64):
  single evaluation : |dg|/|g| <= 1e-11, |dH|_F/|H|_F <= 1e-11 (penetrating states, both friction branches)
  energies          : 1e-11 relative
  scene 11 goldens  : BDF1 1e-9 relative; BDF2 1e-6 relative (the oracle itself lands 9e-8 from the golden through 1200
                      steps of impact and stick/slip switching; the reference's criterion is 1e-2 absolute, Scene.m:173)
  rollouts          : contact switching (d <= 0, static vs dynamic friction) makes a trajectory only piecewise smooth, so two
                      correct solvers that stop Newton at |g| < tol differ by (tol-sized) x (growth through impacts):
                      |dq| <= 1e-6 |q| over 150 steps through impact, energy history to 1e-6 relative of its range
"""
import numpy as np
import pytest

from redmax_amd.scenes import sceneChain, sceneChainGround, scenesRedMax, syntheticStates

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _penetrating_states(name, nr, h, B, rng):
    q0 = np.empty((B, nr))
    qd0 = np.empty((B, nr))
    for b in range(B):
        if name == "11":
            q0[b] = np.array([rng.uniform(-1, 1), rng.uniform(-0.4, 0.4), 0.3])           # JointFree2D: x, y, theta
            qd0[b] = rng.normal(size=nr) * (50 if b % 2 else 0.5)
        else:
            q0[b] = rng.uniform(-0.4, 0.4, nr)
            qd0[b] = rng.normal(size=nr) * (5 if b % 2 else 0.05)
    return q0, qd0, q0 + h * qd0


@pytest.mark.parametrize("name", ["11", "chain6ground", "chain32ground"])
def test_contact_eval_matches_oracle(oracle_lib, name):
    from redmax_amd import BatchSim
    sc = scenesRedMax(11) if name == "11" else sceneChainGround(6 if name == "chain6ground" else 32, ground_z=-1.0)
    sc.init()
    B = 6
    rng = np.random.default_rng(5)
    nr, h = sc.nr, sc.h
    q0, qd0, q1 = _penetrating_states(name, nr, h, B, rng)
    sim = BatchSim(sc, batch=B)
    o = oracle_lib.Oracle(sc.desc())
    o_free = oracle_lib.Oracle(dict(sc.desc(), contact=None))
    touched = 0
    for eta, qA, qB in ((h, q0, q0 + 0.5 * h * qd0), (2 * h / 3, q0 + 1e-3 * rng.normal(size=(B, nr)), q0)):
        g, H = sim.eval_residual(q1, qA, qB, eta)
        g_only = sim.eval_residual(q1, qA, qB, eta, want_H=False)
        for b in range(B):
            go, Ho = o.eval_residual(q1[b], qA[b], qB[b], eta)
            assert _rel(g[b], go) <= 1e-11
            assert _rel(H[b], Ho) <= 1e-11
            assert _rel(g_only[b], go) <= 1e-11
            touched += _rel(o_free.eval_residual(q1[b], qA[b], qB[b], eta)[0], go) > 1e-6
    assert touched >= B                                   # contact forces really were in play
    sim.set_state(q1, qd0)
    T, V = sim.energy()
    for b in range(B):
        o.set_state(q1[b], qd0[b])
        To, Vo = o.energy()
        assert abs(T[b] - To) <= 1e-11 * max(abs(To), 1) and abs(V[b] - Vo) <= 1e-11 * max(abs(Vo), 1)
    sim.close()


def test_scene11_goldens_through_the_drivers():
    """driverRedMaxBDF1(11) / driverRedMaxBDF2(11): Hexpected of scenesRedMax.m:292-293."""
    from redmax_amd import driverRedMaxBDF1, driverRedMaxBDF2
    sc, H, passed = driverRedMaxBDF1(11, verbose=False)
    assert passed and abs(H - sc.Hexpected[0]) <= 1e-9 * abs(sc.Hexpected[0])
    sc, H, passed = driverRedMaxBDF2(11, verbose=False)
    assert passed and abs(H - sc.Hexpected[1]) <= 1e-6 * abs(sc.Hexpected[1])
    assert sc.solverInfo["status"] & 7 == 0


@pytest.mark.parametrize("integ", ["bdf1", "bdf2"])
def test_config5_chain_ground_rollout_matches_oracle(oracle_lib, integ):
    """BASELINE.json configs[4] at test size: the 32-link chain falls onto the frictional ground; 150 steps of h = 5e-4 carry
    it through the first impacts.  Trajectory 0 is the scene's own initial state, the others synthetic."""
    from redmax_amd import BatchSim
    sc = sceneChainGround(32)
    sc.init()
    B, nsteps = 3, 150
    q0, qd0 = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1)     # tip deflection ~0.3 units: the chains start above the ground
    q0[0], qd0[0] = sc.getQ()
    sim = BatchSim(sc, batch=B)
    sim.set_state(q0, qd0)
    step = sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2
    out = step(nsteps, h=sc.h, stats=True, history=True)
    q, qd = sim.get_state()
    assert np.all(out["status"] & 7 == 0)
    for b in range(B):
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(q0[b], qd0[b])
        res = (o.step_bdf1 if integ == "bdf1" else o.step_bdf2)(sc.h, nsteps, history=True)
        st, To, Vo = res
        qo, qdo = o.get_state()
        assert st.diverged == 0 and st.not_converged == 0
        assert Vo.max() - Vo.min() > 1e3                    # the chain did hit the ground
        assert _rel(q[b], qo) <= 1e-6
        assert _rel(qd[b], qdo) <= 1e-5
        Hg, Ho = out["T"][:, b] + out["V"][:, b], To + Vo
        assert np.abs(Hg - Ho).max() <= 1e-6 * (np.abs(Ho).max() + 1)
    sim.close()


@pytest.mark.parametrize("integ", ["bdf1", "bdf2"])
def test_hand_over_from_free_flight_to_contact(oracle_lib, integ):
    """A call on a contact scene is two launches: the lean kernel (plain evaluation + lowest-corner clearance test) takes every
    trajectory up to the step where a cuboid comes within reach of the ground, the kernel with the contact terms takes it from
    there.  An 8-link chain dropped from three heights reaches the ground at three different steps of the same call: the
    per-step history (q, qdot, T, V) across each hand-over is the oracle's, the BDF2 start step is taken once, and the same
    rollout cut into two calls at an arbitrary step gives the same states."""
    from redmax_amd import BatchSim
    nsteps, cut = 120, 37
    for gz, lo, hi in ((-0.8, 30, 60), (-1.2, 60, 90), (-30.0, None, None)):      # the last one never gets there
        sc = sceneChainGround(8, ground_z=gz)
        sc.init()
        rng = np.random.default_rng(77)       # near horizontal: the lowest corners start 0.5 below the root
        q0, qd0 = 1e-3 * rng.normal(size=(2, sc.nr)), 0.05 * rng.normal(size=(2, sc.nr))
        sim = BatchSim(sc, batch=2)
        sim.set_state(q0, qd0)
        step = sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2
        out = step(nsteps, h=sc.h, stats=True, history="full")
        q, qd = sim.get_state()
        assert np.all(out["status"] & 5 == 0)
        scf = sceneChain(8)
        scf.init()
        simf = BatchSim(scf, batch=2)
        simf.set_state(q0, qd0)
        free = (simf.step_bdf1 if integ == "bdf1" else simf.step_bdf2)(nsteps, h=sc.h, history="full")
        simf.close()
        for b in range(2):
            o = oracle_lib.Oracle(sc.desc())
            o.set_state(q0[b], qd0[b])
            st, To, Vo = (o.step_bdf1 if integ == "bdf1" else o.step_bdf2)(sc.h, nsteps, history=True)
            qo, qdo = o.get_state()
            assert st.diverged == 0
            # "Newton did not converge" (a line search stalled at the reference's tol = 1e-9 on a contact kink) is the
            # reference's behaviour too: the oracle and the GPU must agree on it; such a rollout is compared more loosely
            stalled = st.not_converged > 0
            assert bool(out["status"][b] & 2) == stalled
            # the first step at which the trajectory leaves the one of the same chain without a ground: the hand-over step
            first = np.flatnonzero(np.abs(out["q"][:, b] - free["q"][:, b]).max(axis=1) > 1e-9)
            if lo is None:
                assert first.size == 0
            else:
                assert lo <= first[0] <= hi, first[:3]
            tolq = 1e-5 if stalled else 1e-7
            assert _rel(q[b], qo) <= tolq and _rel(qd[b], qdo) <= 10 * tolq
            Hg, Ho = out["T"][:, b] + out["V"][:, b], To + Vo
            assert np.abs(Hg - Ho).max() <= tolq * (np.abs(Ho).max() + 1)
            if not stalled:
                assert abs(int(out["newton_iters"][b]) - st.newton_iters) <= 2
        # the same rollout in two calls
        sim2 = BatchSim(sc, batch=2)
        sim2.set_state(q0, qd0)
        step2 = sim2.step_bdf1 if integ == "bdf1" else sim2.step_bdf2
        o1 = step2(cut, h=sc.h, history="full")
        o2 = step2(nsteps - cut, h=sc.h, history="full")
        qq, qqd = sim2.get_state()
        assert _rel(qq, q) <= 1e-10 and _rel(qqd, qd) <= 1e-9
        assert np.abs(np.concatenate([o1["q"], o2["q"]]) - out["q"]).max() <= 1e-10 * (np.abs(out["q"]).max() + 1)
        sim.close()
        sim2.close()


@pytest.mark.parametrize("integ", ["bdf1", "bdf2"])
def test_contact_on_a_tree_of_more_than_32_nodes(oracle_lib, integ):
    """40-link chain over the ground: the 64-lane kernels.  Their Hessian stage has two forms - matrix cores with H left in LDS for
    the block-column solve (the evaluation without contact terms: the lean launch) and the v_readlane columns with the rows handed
    over in registers (the launch with the contact terms, whether or not a corner touches at an iterate) - and a rollout that
    swings into the ground, bounces and lifts off again goes through all of them: single evaluations with penetrating corners, then
    120 steps from above the ground, against the oracle.  No solve may fall back to partial pivoting (status bit 16: a solve that
    eliminated anything but H would trip the growth guard) and the Newton iteration counts must be the oracle's."""
    from redmax_amd import BatchSim
    sc = sceneChainGround(40, ground_z=-1.0)
    sc.init()
    rng = np.random.default_rng(11)
    B = 3
    q0, qd0, q1 = _penetrating_states("chain", sc.nr, sc.h, B, rng)
    sim = BatchSim(sc, batch=B)
    g, H = sim.eval_bdf1(q1, q0, qd0, sc.h)
    for b in range(B):
        o = oracle_lib.Oracle(sc.desc())
        go, Ho = o.eval_bdf1(q1[b], q0[b], qd0[b], sc.h)
        assert _rel(g[b], go) <= 1e-11 and _rel(H[b], Ho) <= 1e-11
    q0 = 2e-3 * rng.normal(size=(B, sc.nr))
    qd0 = 0.05 * rng.normal(size=(B, sc.nr))
    sim.set_state(q0, qd0)
    nsteps = 120
    out = (sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2)(nsteps, h=sc.h, stats=True, history=True)
    q, qd = sim.get_state()
    assert np.all(out["status"] & 5 == 0)
    assert np.all(out["status"] & 16 == 0), out["status"]
    for b in range(B):
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(q0[b], qd0[b])
        st, To, Vo = (o.step_bdf1 if integ == "bdf1" else o.step_bdf2)(sc.h, nsteps, history=True)
        qo, qdo = o.get_state()
        assert st.diverged == 0
        stalled = st.not_converged > 0
        assert bool(out["status"][b] & 2) == stalled
        if not stalled:
            assert abs(int(out["newton_iters"][b]) - st.newton_iters) <= max(2, 0.02 * st.newton_iters), (b, out["newton_iters"][b], st.newton_iters)
        assert Vo.max() - Vo.min() > 1e2                      # the chain did reach the ground
        tolq = 1e-5 if stalled else 1e-7
        assert _rel(q[b], qo) <= tolq and _rel(qd[b], qdo) <= 10 * tolq
        Hg, Ho = out["T"][:, b] + out["V"][:, b], To + Vo
        assert np.abs(Hg - Ho).max() <= tolq * (np.abs(Ho).max() + 1)
    sim.close()


def _free_box_with_arm_over_ground():
    """A free-flying box (JointFree3D) carrying an arm on a JointSpherical, ForceGroundCuboid on both bodies: multi-DOF joints with
    Euler-chart switching AND ground contact in one scene (the reference has no such scene; the oracle is the check)."""
    from redmax_amd.redmax import BodyCuboid, ForceGroundCuboid, JointFree3D, JointSpherical, Scene
    from redmax_amd.scenes import _T
    sc = Scene()
    sc.name = "free box with a spherical-jointed arm over ground"
    sc.grav = np.array([0.0, 0.0, -980.0])
    sc.h, sc.tEnd = 5e-4, 0.2
    b0 = BodyCuboid(1.0, [4, 3, 2])
    j0 = JointFree3D(None, b0)
    j0.setJointTransform(_T([0, 0, 5.0]))
    b0.setBodyTransform(np.eye(4))
    b1 = BodyCuboid(1.0, [6, 1, 1])
    j1 = JointSpherical(j0, b1)
    j1.setJointTransform(_T([2, 0, 0]))
    b1.setBodyTransform(_T([3, 0, 0]))
    j0.q[:] = [0.3, -0.2, 0.1, 0, 0, 0]
    j0.qdot[:] = [6.0, -9.0, 4.0, 5.0, 0, 0]
    j1.q[:] = [0.2, 0.9, -0.1]
    j1.qdot[:] = [9.0, -24.0, 6.0]
    sc.bodies += [b0, b1]
    sc.joints += [j0, j1]
    for b in sc.bodies:
        f = ForceGroundCuboid(b)
        f.setTransform(np.eye(4))
        f.setStiffness(1e5, 1e2)
        f.setDamping(3e1)
        f.setFriction(0.5)
        sc.forces.append(f)
    sc.init()
    return sc


@pytest.mark.parametrize("integ", ["bdf1", "bdf2"])
def test_spherical_joints_and_ground_contact_together(oracle_lib, integ):
    """400 steps: ~200 of free flight with tumbling (lean launch, chart switches), then touch-down and sliding (the launch with
    the contact terms): per-step energies, final state, charts and Newton iteration count against the oracle."""
    from redmax_amd import BatchSim
    sc = _free_box_with_arm_over_ground()
    nsteps = 400
    q0, qd0 = sc.getQ()
    sim = BatchSim(sc, batch=2)
    sim.set_state(np.stack([q0, q0]), np.stack([qd0, 0.5 * qd0]))
    out = (sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2)(nsteps, h=sc.h, stats=True, history="full")
    q, qd = sim.get_state()
    charts = sim.charts()
    assert np.all(out["status"] & 7 == 0)
    for b, scale in enumerate((1.0, 0.5)):
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(q0, scale * qd0)
        st, To, Vo = (o.step_bdf1 if integ == "bdf1" else o.step_bdf2)(sc.h, nsteps, history=True)
        qo, qdo = o.get_state()
        assert st.diverged == 0 and st.not_converged == 0
        Ho = To + Vo
        touch = np.flatnonzero(np.abs(np.diff(Ho)) > 1e-3 * np.abs(Ho).max())
        assert touch.size and 100 < touch[0] < 350          # free flight first, then the ground
        assert _rel(q[b], qo) <= 1e-6 and _rel(qd[b], qdo) <= 1e-5
        Hg = out["T"][:, b] + out["V"][:, b]
        assert np.abs(Hg - Ho).max() <= 1e-6 * (np.abs(Ho).max() + 1)
        assert abs(int(out["newton_iters"][b]) - st.newton_iters) <= 3
        assert list(charts[b]) == list(o.charts())
    sim.close()


def test_contact_is_refused_by_euler_and_adjoint():
    from redmax_amd import BatchSim, RedMaxHipError
    sc = sceneChainGround(4)
    sc.init()
    sim = BatchSim(sc, batch=1)
    with pytest.raises(RedMaxHipError):
        sim.step_euler(1, 1e-2)
    task = {"body": 3, "xlocal": [5.0, 0, 0], "xtarget": [10.0, 0, -10.0], "step": 1, "pscale": 1e5, "wreg": 1e-2, "wpos": 1e2}
    with pytest.raises(RedMaxHipError):
        sim.adjoint_bdf1(1, sc.h, task, np.zeros((1, sc.nr)))
    sim.close()


def test_contact_free_scene_takes_the_plain_kernels(oracle_lib):
    """All flags zero: rmx_model_set_ground_contact removes the contact again (the plain instantiations run)."""
    from redmax_amd import BatchSim
    sc = sceneChainGround(6, ground_z=-1.0)
    sc.init()
    d = dict(sc.desc())
    d["contact"] = np.zeros_like(np.asarray(d["contact"]))
    sim = BatchSim(d, batch=1)
    rng = np.random.default_rng(1)
    q0, qd0 = rng.uniform(-0.4, 0.4, (1, sc.nr)), rng.normal(size=(1, sc.nr))
    g, H = sim.eval_bdf1(q0 + sc.h * qd0, q0, qd0, sc.h)
    o = oracle_lib.Oracle(dict(d, contact=None))
    go, Ho = o.eval_bdf1((q0 + sc.h * qd0)[0], q0[0], qd0[0], sc.h)
    assert _rel(g[0], go) <= 1e-11 and _rel(H[0], Ho) <= 1e-11
    sim.close()


@pytest.mark.parametrize("integ", ["bdf1", "bdf2"])
def test_force_objects_with_their_own_ground_frames(oracle_lib, integ):
    """Every ForceGroundCuboid object holds its own E, kn, kt, mu, kd (ForceGroundCuboid.m:6-13).  An 8-link chain whose even bodies
    meet a floor and whose odd bodies meet a tilted, softer, nearly frictionless plane: g, H at penetrating states (1e-11) and a
    rollout through the contact (1e-6 |q|) against the oracle; and per-body copies of ONE frame reproduce the shared-frame results
    bit for bit."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChainTwoGrounds
    sc = sceneChainTwoGrounds(8, ground_z=-1.0)
    sc.init()
    d = sc.desc()
    assert d.get("ground_body") is not None
    B = 4
    rng = np.random.default_rng(11)
    nr, h = sc.nr, sc.h
    q0, qd0, q1 = _penetrating_states("chain8two", nr, h, B, rng)
    sim = BatchSim(sc, batch=B)
    o = oracle_lib.Oracle(d)
    o_shared = oracle_lib.Oracle(dict(d, ground_body=None))
    differs = 0
    g, H = sim.eval_residual(q1, q0, q0 + 0.5 * h * qd0, h)
    for b in range(B):
        go, Ho = o.eval_residual(q1[b], q0[b], q0[b] + 0.5 * h * qd0[b], h)
        assert _rel(g[b], go) <= 1e-11 and _rel(H[b], Ho) <= 1e-11
        differs += _rel(o_shared.eval_residual(q1[b], q0[b], q0[b] + 0.5 * h * qd0[b], h)[0], go) > 1e-6
    assert differs >= 2                                    # the second plane really is in play
    K = 60
    sim.set_state(q1, qd0)
    out = (sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2)(K, h=h, stats=True)
    q, qd = sim.get_state()
    for b in range(B):
        ob = oracle_lib.Oracle(d)
        ob.set_state(q1[b], qd0[b])
        (ob.step_bdf1 if integ == "bdf1" else ob.step_bdf2)(h, K)
        qo, qdo = ob.get_state()
        assert _rel(q[b], qo) <= 1e-6 and _rel(qd[b], qdo) <= 1e-4
    sim.close()
    # n copies of one frame = the shared frame, bit for bit
    sg = sceneChainGround(8, ground_z=-1.0)
    sg.init()
    dg = sg.desc()
    n = len(dg["contact"])
    gg = dg["ground"]
    sims = []
    for gb in (None, {"E": np.stack([gg["E"]] * n), "kn": np.full(n, gg["kn"]), "kt": np.full(n, gg["kt"]), "mu": np.full(n, gg["mu"]),
                      "kd": np.full(n, gg["kd"])}):
        s2 = BatchSim(dict(dg, ground_body=gb), batch=B)
        sims.append(s2.eval_residual(q1, q0, q0 + 0.5 * h * qd0, h))
        s2.close()
    assert np.array_equal(sims[0][0], sims[1][0]) and np.array_equal(sims[0][1], sims[1][1])


@pytest.mark.parametrize("name", ["chain6ground", "chain32ground", "chain8two", "11"])
def test_compute_values_hook_with_ground_contact(oracle_lib, name):
    """rmx_eval_mfd on scenes with ForceGroundCuboid: the contact wrench is part of f, its damping block J' Dm J part of D
    (ForceGroundCuboid.m:104-106, 135-150), both friction branches; M, f, D vs the oracle's computeValues at penetrating states."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChainTwoGrounds
    if name == "11":
        sc = scenesRedMax(11)
    elif name == "chain8two":
        sc = sceneChainTwoGrounds(8, ground_z=-1.0)
    else:
        sc = sceneChainGround(6 if name == "chain6ground" else 32, ground_z=-1.0)
    sc.init()
    B = 6
    rng = np.random.default_rng(21)
    _, qd, q = _penetrating_states(name, sc.nr, sc.h, B, rng)
    sim = BatchSim(sc, batch=B)
    M, f, D = sim.eval_mfd(q, qd)
    o_free = oracle_lib.Oracle(dict(sc.desc(), contact=None))
    touched = 0
    for b in range(B):
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(q[b], qd[b])
        Mo, fo, _, _, Do = o.compute_values(deriv=True)
        assert _rel(M[b], Mo) <= 1e-11 and _rel(f[b], fo) <= 1e-11, (name, b, _rel(M[b], Mo), _rel(f[b], fo))
        assert _rel(D[b], Do) <= 1e-11, (name, b, _rel(D[b], Do))
        o_free.set_state(q[b], qd[b])
        touched += _rel(o_free.compute_values(deriv=True)[4], Do) > 1e-6
    assert touched >= 1                                   # the contact damping really was in D
    sim.close()


@pytest.mark.parametrize("name", ["chain6ground", "chain8two", "11"])
def test_compute_values_full_output_with_ground_contact(oracle_lib, name):
    """rmx_compute_values on scenes with ForceGroundCuboid at penetrating states: K carries J' Km J and the dJdq' fm term of the contact
    wrench (driverRedMaxBDF1.m:239-243, ForceGroundCuboid.m:104-150), both friction branches; K, D, dMv vs the oracle."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChainTwoGrounds
    if name == "11":
        sc = scenesRedMax(11)
    elif name == "chain8two":
        sc = sceneChainTwoGrounds(8, ground_z=-1.0)
    else:
        sc = sceneChainGround(6, ground_z=-1.0)
    sc.init()
    B = 6
    rng = np.random.default_rng(22)
    _, qd, q = _penetrating_states(name, sc.nr, sc.h, B, rng)
    v = rng.standard_normal((B, sc.nr)) * 1e-2
    sim = BatchSim(sc, batch=B)
    out = sim.compute_values(q, qd, v=v)
    o_free = oracle_lib.Oracle(dict(sc.desc(), contact=None))
    touched = 0
    for b in range(B):
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(q[b], qd[b])
        Mo, fo, dMo, Ko, Do = o.compute_values(deriv=True)
        scale = np.linalg.norm(Mo) + np.linalg.norm(Do) + np.linalg.norm(Ko)
        assert _rel(out["M"][b], Mo) <= 1e-11 and _rel(out["f"][b], fo) <= 1e-11
        assert np.linalg.norm(out["D"][b] - Do) <= 1e-11 * scale
        assert np.linalg.norm(out["K"][b] - Ko) <= 1e-10 * scale, (name, b, np.linalg.norm(out["K"][b] - Ko) / scale)
        assert np.linalg.norm(out["dMv"][b] - np.einsum("rci,c->ri", dMo, v[b])) <= 1e-10 * scale
        o_free.set_state(q[b], qd[b])
        touched += _rel(o_free.compute_values(deriv=True)[3], Ko) > 1e-6
    assert touched >= 1                                   # the contact stiffness really was in K
    sim.close()


def test_config5_flagged_rollouts_replay_on_the_oracle(oracle_lib):
    """BASELINE.json configs[4] (32-link chain over the frictional ground, BDF2, h = 5e-4, 100 steps): the rollouts whose Newton creeps
    through its 320 iterations on one step (status MAXITER; the default launch parks them and finishes them in cooperative groups of
    wavefronts, rmx_ct32.h) against the literal oracle: the Newton iteration and line-search halving counts (equal on most, within 1 % on all), the same
    'did not converge' verdict, q to 1e-6 (1e-7 on the rollouts that converge on every step).  (the manual script tests/config5_check.py of rounds 2-4, collected.)"""
    from concurrent.futures import ThreadPoolExecutor
    from redmax_amd import BatchSim, sceneChainGround, syntheticStates
    B, K = 256, 100
    sc = sceneChainGround(32)
    sc.init()
    q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1)
    q[0], qd[0] = sc.getQ()
    sim = BatchSim(sc, batch=B)
    sim.set_state(q, qd)
    out = sim.step_bdf2(K, h=sc.h, stats=True)
    qg, _ = sim.get_state()
    sim.close()
    it, ls, st = out["newton_iters"], out["ls_halvings"], out["status"]
    assert (st & 512).max() == 0                      # no cooperative group gave up
    assert (st & (64 | 256)).max() == 0               # the launch's internal hand-over bits (left the lean solve, parked) never reach the caller
    flagged = np.nonzero(st & 2)[0]
    assert len(flagged) >= 4, len(flagged)            # the workload does have such rollouts (12 of the first 256 when written)
    pick = list(flagged[:5]) + [1, 2]

    def replay(b):
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(q[b], qd[b])
        s = o.step_bdf2(sc.h, K)
        return s, o.get_state()[0]
    with ThreadPoolExecutor(max_workers=len(pick)) as ex:
        res = list(ex.map(replay, pick))
    same_counts = 0
    for b, (s, qo) in zip(pick, res):
        print("rollout %3d: iterations gpu %d oracle %d, halvings gpu %d oracle %d, oracle not converged %d" % (
            b, it[b], s.newton_iters, ls[b], s.ls_halvings, s.not_converged))
        # On a step where Newton creeps through its 320 iterations the counts are decided by f < f0 at the 15th digit, 20 times per line
        # search: identical to the oracle's on most of these rollouts (524 / 524, 521 / 521 ...), within 1 % on the others (538 vs 535
        # iterations on rollout 20, 3311 vs 3319 halvings on rollout 11 when written); identical on the rollouts that converge throughout
        assert abs(int(it[b]) - s.newton_iters) <= 0.01 * s.newton_iters and abs(int(ls[b]) - s.ls_halvings) <= 0.01 * s.ls_halvings, b
        same_counts += int(it[b] == s.newton_iters)
        if not (st[b] & 2):
            assert it[b] == s.newton_iters and ls[b] == s.ls_halvings, b
        assert bool(st[b] & 2) == (s.not_converged > 0) and s.diverged == 0
        # (a step both sides end with 'did not converge' leaves its state to the last bits of 320 creeping iterations: 1.5e-7 on
        # rollout 11 when written; the rollouts that converge throughout to 1e-7)
        assert np.linalg.norm(qg[b] - qo) <= (1e-6 if st[b] & 2 else 1e-7) * np.linalg.norm(qo), b
    assert same_counts >= (len(pick) + 1) // 2, same_counts


def test_floor_and_wall_on_every_body(oracle_lib):
    """Several ForceGroundCuboid objects on one body (Force.m:26-56: the reference's forces are a list; refused up to round 4).  A 6-link
    chain whose every body carries a floor and a wall: the host mirror lists each second force as a fixed, massless child of the body's
    joint (one force per listing entry is what the C ABI takes), the ORACLE adds both objects to the same body literally
    (Scene.desc_literal()).  g, H at states inside both planes, energies, BDF1 and BDF2 rollouts."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChainFloorAndWall
    sc = sceneChainFloorAndWall(6, ground_z=-1.0)
    sc.init()
    lit = sc.desc_literal()
    assert sc.desc()["njoints"] == 12 and lit["njoints"] == 6
    B, nr, h = 3, sc.nr, sc.h
    rng = np.random.default_rng(12)
    q0 = rng.uniform(0.05, 0.35, (B, nr))
    qd0 = rng.uniform(-3, 3, (B, nr))
    q1 = q0 + h * qd0 + 1e-4 * rng.standard_normal((B, nr))
    sim = BatchSim(sc, batch=B)
    assert sim.nr == nr
    g, H = sim.eval_residual(q1, q0, q0 + h * qd0, h)
    sim.set_state(q0, qd0)
    T, V = sim.energy()
    o_floor = oracle_lib.Oracle(dict(lit, extra_forces=None))
    for b in range(B):
        o = oracle_lib.Oracle(lit)
        go, Ho = o.eval_residual(q1[b], q0[b], q0[b] + h * qd0[b], h)
        assert _rel(g[b], go) <= 1e-10 and _rel(H[b], Ho) <= 1e-10, b
        o.set_state(q0[b], qd0[b])
        To, Vo = o.energy()
        assert abs(T[b] - To) <= 1e-10 * max(abs(To), 1) and abs(V[b] - Vo) <= 1e-10 * max(abs(Vo), 1)
        o_floor.set_state(q0[b], qd0[b])
        assert Vo > o_floor.energy()[1] > 0             # the wall is in play, and so is the floor
    for integ in ("bdf1", "bdf2"):
        sim.set_state(q0, qd0)
        out = (sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2)(40, h=h, stats=True)
        qg, _ = sim.get_state()
        for b in range(B):
            o = oracle_lib.Oracle(lit)
            o.set_state(q0[b], qd0[b])
            st = (o.step_bdf1 if integ == "bdf1" else o.step_bdf2)(h, 40)
            qo, _ = o.get_state()
            if st.diverged or st.not_converged:
                assert out["status"][b] & 3
                continue
            assert out["status"][b] & 7 == 0, (integ, b)
            assert out["newton_iters"][b] == st.newton_iters, (integ, b, out["newton_iters"][b], st.newton_iters)
            assert _rel(qg[b], qo) <= 1e-7, (integ, b, _rel(qg[b], qo))
    sim.close()


def test_one_launch_cooperative_path_at_odd_sizes(monkeypatch):
    """The default launch of a chain with ground contact is ONE kernel in which the rollouts and, dispatched behind them, the cooperative
    groups run side by side (rmx_kernels.hip k_ground32).  Sizes it must survive: more rollouts than SIMDs (1500 on 1024), a handful (37:
    fewer rollouts than a group has members to spare), and two shards of 600 on ONE device - two such launches competing for the SIMDs.
    Every case against one wavefront per rollout: states bit for bit, no group gave up (status 512)."""
    from redmax_amd import BatchSim, GroupSim, sceneChainGround, syntheticStates
    sc = sceneChainGround(32)
    sc.init()

    def run(B, park, group=False):
        monkeypatch.setenv("RMX_PARK_HALVINGS", park)
        q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1)
        q[0], qd[0] = sc.getQ()
        sim = GroupSim(sc, B, devices=(0, 0)) if group else BatchSim(sc, batch=B)
        sim.set_state(q, qd)
        out = sim.step(100, integrator=2, h=sc.h) if group else sim.step_bdf2(100, h=sc.h, stats=True)
        qf, qdf = sim.get_state()
        sim.close()
        return qf, qdf, out
    for B in (1500, 37):
        ref = run(B, "0")
        got = run(B, "24")
        assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]), B
        assert np.array_equal(ref[2]["newton_iters"], got[2]["newton_iters"]) and (got[2]["status"] & 512).max() == 0, B
    ref = run(1200, "24")
    got = run(1200, "24", group=True)
    assert np.array_equal(ref[0], got[0]) and (got[2]["status"] & 512).max() == 0
    # ... and the same two launches with (nearly) every rollout parked - a threshold of one halving -: the SIMDs of the device are divided
    # among the live batches that may hold cooperative groups (launch_step), so the groups of both launches are resident together and
    # neither waits for members the other one keeps from being dispatched (round-5 advice); one wavefront per rollout is the reference
    ref = run(600, "0")
    got = run(600, "1", group=True)
    assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
    assert np.array_equal(ref[2]["newton_iters"], got[2]["newton_iters"]) and (got[2]["status"] & 512).max() == 0
