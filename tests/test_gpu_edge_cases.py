"""Edge cases of the boundary: smallest / largest trees, fixed-joint padding, listing order, empty work, bad input."""
import ctypes as C
import math

import numpy as np
import pytest

from redmax_amd import se3
from redmax_amd.redmax import BodyCuboid, JointFixed, JointRevolute, Scene
from redmax_amd.scenes import sceneChain, scenesRedMax

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _single_revolute():
    """scenesRedMax.m:13-26 (scene -2, 'Single revolute'): the smallest tree, nr = 1."""
    sc = Scene()
    b = BodyCuboid(1.0, [2, 0.2, 0.2])
    j = JointRevolute(None, b, [0, 1, 0])
    j.setJointTransform(np.eye(4))
    j.qdot[0] = 1.0
    b.setBodyTransform(se3.transform(p=[1, 0, 0]))
    sc.bodies, sc.joints = [b], [j]
    return sc


def test_single_joint_scene(oracle_lib):
    from redmax_amd import BatchSim
    sc = _single_revolute()
    sc.init()
    sim = BatchSim(sc, batch=2)
    q0, qd0 = sc.getQ()
    sim.set_state(np.stack([q0, q0 + 0.3]), np.stack([qd0, qd0]))
    out = sim.step_bdf1(50, h=sc.h, stats=True)
    qg, qdg = sim.get_state()
    for b, dq in enumerate((0.0, 0.3)):
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(q0 + dq, qd0)
        o.step_bdf1(sc.h, 50)
        qo, qdo = o.get_state()
        assert _rel(qg[b], qo) <= 1e-10 and _rel(qdg[b], qdo) <= 1e-8
    assert (out["status"] & 15 == 0).all()


def test_maximum_size_chain64(oracle_lib):
    """n = 64 nodes is the per-wavefront maximum (all 64 lanes are nodes, 4 DPP rows in the chain scans)."""
    from redmax_amd import BatchSim
    sc = sceneChain(64)
    sc.init()
    rng = np.random.default_rng(12)
    q = rng.uniform(-0.05, 0.05, (2, 64))
    qd = rng.uniform(-0.05, 0.05, (2, 64))
    sim = BatchSim(sc, batch=2)
    sim.opts.tol = 1e-7        # |g| starts ~1e5 on this 6.4 m chain: the reference's 1e-9 is below its fp64 noise floor
    g, H = sim.eval_bdf1(q + 1e-3, q, qd, 1e-2)
    sim.set_state(q, qd)
    out = sim.step_bdf1(3, h=1e-2, stats=True)
    qg, _ = sim.get_state()
    oracle_lib.set_newton(tol=1e-7)
    try:
        for b in range(2):
            o = oracle_lib.Oracle(sc.desc())
            go, Ho = o.eval_bdf1(q[b] + 1e-3, q[b], qd[b], 1e-2)
            assert _rel(g[b], go) <= 1e-11 and _rel(H[b], Ho) <= 1e-11
            o.set_state(q[b], qd[b])
            o.step_bdf1(1e-2, 3)
            assert _rel(qg[b], o.get_state()[0]) <= 1e-8
    finally:
        oracle_lib.set_newton()
    assert not (out["status"] & 5).any()


def test_too_many_joints_is_an_error_not_a_fallback():
    from redmax_amd import BatchSim, RedMaxHipError
    sc = sceneChain(257)          # 65..256 nodes run on the one-workgroup-per-trajectory kernels (tests/test_gpu_big_trees.py)
    sc.init()
    with pytest.raises(RedMaxHipError, match="njoints"):
        BatchSim(sc, batch=1)


def test_listing_that_is_not_depth_first_is_reordered_inside(oracle_lib):
    """The C ABI accepts any parent-before-child listing (the reference silently assumes depth-first order, Joint.m:134-146);
    reduced indices still follow the LISTING (Scene.m:69-71).  Scene 2 listed breadth-first with an extra grandchild."""
    from redmax_amd import BatchSim
    sc = scenesRedMax(2)
    # add a grandchild under joint 3 so that depth-first order differs from this listing [1,2,3,4,5(child of 3)]
    b5 = BodyCuboid(1.0, [1, 1, 4])
    j5 = JointRevolute(sc.joints[2], b5, [0, 1, 0])
    j5.setJointTransform(se3.transform(p=[0, 0, -10]))
    b5.setBodyTransform(se3.transform(p=[0, 0, -2]))
    j5.q[0] = 0.2
    sc.bodies.append(b5)
    sc.joints.append(j5)
    # bypass Scene.init()'s ordering check: build the descriptor by hand in this (non depth-first) listing order
    for b in sc.bodies:
        b.computeInertia()
    nr = 0
    for j in reversed(sc.joints):
        j.idxR = list(range(nr, nr + j.ndof))
        nr += j.ndof
        j.qRest = float(j.q[0])
    sc.nr, sc.nm = nr, 6 * len(sc.joints)
    d = sc.desc()
    assert list(d["parent"]) == [-1, 0, 1, 1, 2]
    rng = np.random.default_rng(2)
    q = rng.uniform(-0.5, 0.5, (2, nr))
    qd = rng.uniform(-1, 1, (2, nr))
    sim = BatchSim(d, batch=2)
    assert list(sim.idxR()) == [4, 3, 2, 1, 0]
    g, H = sim.eval_bdf1(q + 1e-3, q, qd, 1e-2)
    sim.set_state(q, qd)
    sim.step_bdf1(5, h=1e-2)
    qg, _ = sim.get_state()
    for b in range(2):
        o = oracle_lib.Oracle(d)
        go, Ho = o.eval_bdf1(q[b] + 1e-3, q[b], qd[b], 1e-2)
        assert _rel(g[b], go) <= 1e-11 and _rel(H[b], Ho) <= 1e-11
        o.set_state(q[b], qd[b])
        o.step_bdf1(1e-2, 5)
        assert _rel(qg[b], o.get_state()[0]) <= 1e-10


def test_zero_steps_and_roundtrip_state():
    from redmax_amd import BatchSim
    sc = scenesRedMax(1)
    sc.init()
    sim = BatchSim(sc, batch=3)
    q = np.arange(9, dtype=float).reshape(3, 3) / 10
    qd = -q
    sim.set_state(q, qd)
    sim.step_bdf1(0, h=1e-2)
    q2, qd2 = sim.get_state()
    assert np.array_equal(q, q2) and np.array_equal(qd, qd2)


def test_bad_arguments_return_errors():
    from redmax_amd import BatchSim, RedMaxHipError, _abi
    sc = scenesRedMax(0)
    sc.init()
    sim = BatchSim(sc, batch=1)
    with pytest.raises(RedMaxHipError):
        sim.step_bdf1(1, h=-1.0)
    L = _abi.lib()
    assert L.rmx_step_bdf1(None, None, 1, None, None, None) < 0
    assert b"null" in L.rmx_last_error()
    d = dict(sc.desc())
    d["parent"] = np.array([-1, 0, 5, 2, 3], dtype=np.int32)      # forward reference
    with pytest.raises(RedMaxHipError, match="parent-before-child"):
        BatchSim(d, batch=1)


def test_nonconverging_solve_is_a_status_not_an_error():
    """The reference prints 'Newton did not converge' and continues (driverRedMaxBDF1.m:150-153)."""
    from redmax_amd import BatchSim
    sc = scenesRedMax(0)
    sc.init()
    sim = BatchSim(sc, batch=1)
    sim.opts.tol = 1e-30           # unreachable
    q0, qd0 = sc.getQ()
    sim.set_state(q0[None], qd0[None])
    out = sim.step_bdf1(2, h=sc.h, stats=True)
    assert out["status"][0] & 2
    q, _ = sim.get_state()
    assert np.isfinite(q).all()


@pytest.mark.parametrize("name", ["2", "chain8", "4", "8", "9", "11", "chain6ground"])
def test_hessian_is_the_jacobian_of_g_on_the_device(name):
    """The reference's own gradient check inside newton (testGrad, driverRedMaxBDF1.m:104-115) run on the HIP path alone:
    central differences of rmx_eval's g reproduce its H (1e-6 relative, the pass criterion of Scene.printError, Scene.m:424-450).
    No oracle involved; covers revolute/prismatic trees, multi-DOF joints, Euler-chart joints and ground contact."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChainGround
    if name == "chain8":
        sc = sceneChain(8, axis=(0.3, 1.0, 0.2))
    elif name == "chain6ground":
        sc = sceneChainGround(6, ground_z=-1.0)
    else:
        sc = scenesRedMax(int(name))
    sc.init()
    nr, h = sc.nr, sc.h
    rng = np.random.default_rng(21)
    q0 = rng.uniform(-0.4, 0.4, nr)
    if name == "11":
        q0 = np.array([0.2, -0.1, 0.3])                      # corners below the ground plane
    qd0 = rng.uniform(-1, 1, nr)
    q1 = q0 + h * qd0
    B = 2 * nr + 1
    X = np.tile(q1, (B, 1))
    step = 1e-6
    for i in range(nr):
        X[1 + 2 * i, i] += step
        X[2 + 2 * i, i] -= step
    sim = BatchSim(sc, batch=B)
    g, H = sim.eval_bdf1(X, np.tile(q0, (B, 1)), np.tile(qd0, (B, 1)), h)
    Hfd = np.stack([(g[1 + 2 * i] - g[2 + 2 * i]) / (2 * step) for i in range(nr)], axis=1)
    assert _rel(Hfd, H[0]) < 1e-6
    sim.close()


def test_planar_joint_with_a_skew_plane_matches_oracle(oracle_lib):
    """JointPlanar(parent, body, plane) with non-orthogonal in-plane directions (JointPlanar.m:11-19 only normalises the
    columns): rmx_model_desc.plane reaches the lowered prismatic pair."""
    from redmax_amd import BatchSim
    from redmax_amd.redmax import JointPlanar
    sc = Scene()
    b1 = BodyCuboid(1.0, [4, 4, 1])
    j1 = JointPlanar(None, b1, plane=np.array([[1.0, 0.2, 0.0], [0.3, 1.0, 0.5]]).T)
    j1.setJointTransform(se3.transform(R=se3.aaToMat([0, 0, 1], 0.3)))
    b1.setBodyTransform(se3.transform(p=[0.5, 0, 0]))
    b2 = BodyCuboid(1.0, [1, 1, 6])
    j2 = JointRevolute(j1, b2, [0, 1, 0])
    j2.setJointTransform(se3.transform(p=[-2, 0, 0]))
    b2.setBodyTransform(se3.transform(p=[0, 0, -3]))
    j1.setStiffness(5e3)
    j1.q[:] = [0.3, -0.2]
    j1.qdot[:] = [1.0, 2.0]
    j2.q[0] = 0.4
    sc.bodies, sc.joints = [b1, b2], [j1, j2]
    sc.init()
    assert sc.nr == 3 and [j.idxR for j in sc.joints] == [[1, 2], [0]]
    sim = BatchSim(sc, batch=1)
    q0, qd0 = sc.getQ()
    sim.set_state(q0[None], qd0[None])
    o = oracle_lib.Oracle(sc.desc())
    g, H = sim.eval_bdf1((q0 + 1e-2 * qd0)[None], q0[None], qd0[None], 1e-2)
    go, Ho = o.eval_bdf1(q0 + 1e-2 * qd0, q0, qd0, 1e-2)
    assert _rel(g[0], go) <= 1e-11 and _rel(H[0], Ho) <= 1e-11
    sim.step_bdf1(20, h=1e-2)
    o.set_state(q0, qd0)                                             # eval_bdf1 left the oracle at the evaluation point
    o.step_bdf1(1e-2, 20)
    assert _rel(sim.get_state()[0][0], o.get_state()[0]) <= 1e-9     # the joint spring pulls towards qRest = initial q (both DOFs)
    sim.close()


def test_lowered_tree_larger_than_the_node_limit_is_an_error():
    from redmax_amd import BatchSim, RedMaxHipError
    from redmax_amd.redmax import JointFree3D
    sc = Scene()
    prev = None
    for i in range(43):                                       # 43 x 6 DOF = 258 nodes after lowering; the limit is 256
        b = BodyCuboid(1.0, [1, 1, 1])                        # (66 nodes - 11 bodies - run: tests/test_gpu_big_trees.py, free20)
        j = JointFree3D(prev, b)
        j.setJointTransform(se3.transform(p=[1, 0, 0]))
        sc.bodies.append(b)
        sc.joints.append(j)
        prev = j
    sc.init()
    with pytest.raises(RedMaxHipError, match="1-DOF nodes"):
        BatchSim(sc, batch=1)


def test_chart_api_validation():
    from redmax_amd import BatchSim, RedMaxHipError
    sc = scenesRedMax(7)
    sc.init()
    sim = BatchSim(sc, batch=3)
    assert sim.nsph == 2 and sim.charts().shape == (3, 2) and np.all(sim.charts() == 7)
    sim.set_charts([1, 12])
    assert np.array_equal(sim.charts(), np.tile([1, 12], (3, 1)))
    with pytest.raises(RedMaxHipError):
        sim.set_charts([0, 13])
    q0, qd0 = sc.getQ()
    sim.set_state(np.tile(q0, (3, 1)), np.tile(qd0, (3, 1)))          # a new state is read in CHART_XYZ
    assert np.all(sim.charts() == 7)
    sim.close()
    plain = BatchSim(scenesRedMaxInit(0), batch=1)
    assert plain.nsph == 0 and plain.charts().shape == (1, 0)
    plain.close()


def scenesRedMaxInit(sid):
    sc = scenesRedMax(sid)
    sc.init()
    return sc


@pytest.mark.parametrize("name", ["chain20", "tree25", "chain21fixed", "chain17ground"])
def test_partially_filled_32_lane_trees_match_oracle(oracle_lib, name):
    """17 <= n <= 31 nodes run the n <= 32 code (MFMA Hessian, two lanes per component in the subtree scan) with idle node
    slots: (g, H), energies and a short rollout vs the oracle; also with fixed joints (no DOF) and with ground contact."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChainGround, sceneTree
    if name == "chain20":
        sc = sceneChain(20, axis=(0.2, 1.0, -0.3))
    elif name == "tree25":
        sc = sceneTree(25)
    elif name == "chain17ground":
        sc = sceneChainGround(17, ground_z=-1.0)
    else:
        sc = Scene()
        for i in range(21):                                   # scene 0's pattern (scenesRedMax.m:52-79): every other joint fixed
            sc.bodies.append(BodyCuboid(1.0, [4, 1, 1]))
            parent = sc.joints[i - 1] if i else None
            j = JointRevolute(parent, sc.bodies[-1], [0, 1, 0]) if i % 2 == 0 else JointFixed(parent, sc.bodies[-1])
            j.setJointTransform(np.eye(4) if i == 0 else se3.transform(p=[4, 0, 0]))
            sc.bodies[-1].setBodyTransform(se3.transform(p=[2, 0, 0]))
            sc.joints.append(j)
    sc.init()
    nr, h = sc.nr, sc.h
    rng = np.random.default_rng(17)
    B = 3
    q0 = rng.uniform(-0.3, 0.3, (B, nr))
    qd0 = rng.uniform(-1, 1, (B, nr))
    q1 = q0 + h * qd0 + rng.uniform(-1e-3, 1e-3, (B, nr))
    sim = BatchSim(sc, batch=B)
    o = oracle_lib.Oracle(sc.desc())
    g, H = sim.eval_bdf1(q1, q0, qd0, h)
    for b in range(B):
        go, Ho = o.eval_bdf1(q1[b], q0[b], qd0[b], h)
        assert _rel(g[b], go) <= 1e-11 and _rel(H[b], Ho) <= 1e-11
    sim.set_state(q0, qd0)
    out = sim.step_bdf1(8, h=h, stats=True)
    qg, qdg = sim.get_state()
    assert np.all(out["status"] & 7 == 0)
    for b in range(B):
        o.set_state(q0[b], qd0[b])
        o.step_bdf1(h, 8)
        qo, qdo = o.get_state()
        assert np.linalg.norm(qg[b] - qo) <= 1e-9 * np.linalg.norm(qo) + 1e-10
    sim.close()
