"""The algebra the HIP kernels implement (tests/proto_worldframe.py: world-frame recursive Newton-Euler with analytic
derivatives, no J / dJdq) equals the reference's literal tensor formulation (the oracle) to roundoff - on every
in-scope scene, skew axes, the 32-chain and a branching prismatic/revolute tree."""
import numpy as np
import pytest

import proto_worldframe as pw
from redmax_amd import se3
from redmax_amd.scenes import sceneChain, sceneChainGround, scenesRedMax, sceneTree


def _scene(name):
    if name == "chain32":
        return sceneChain(32)
    if name == "chain8skew":
        return sceneChain(8, axis=(0.3, 1.0, 0.2))
    if name == "tree15":
        return sceneTree(15)
    return scenesRedMax(int(name))


@pytest.mark.parametrize("name", ["0", "1", "2", "3", "14", "chain8skew", "chain32", "tree15", "4", "5", "6", "8"])
def test_worldframe_equals_tensor_formulation(oracle_lib, name):
    sc = _scene(name)
    sc.init()
    d = sc.desc()
    o = oracle_lib.Oracle(d)
    m = pw.build_model(oracle_lib.lower_composite(d))     # scenes 4, 5, 6, 8: multi-DOF joints as 1-DOF chains, massless links
    rng = np.random.default_rng(11)
    nr, h = o.nr, sc.h
    q0 = rng.uniform(-0.7, 0.7, nr)
    qd0 = rng.uniform(-1, 1, nr)
    q1 = q0 + h * qd0 + rng.uniform(-1e-2, 1e-2, nr)
    if name == "14":
        q1 = rng.uniform(-2.0, 0.5, nr)
    g, H = o.eval_bdf1(q1, q0, qd0, h)
    g2, H2 = pw.eval_world(m, q1, q0, q0 + h * qd0, h)
    assert np.linalg.norm(g - g2) <= 1e-12 * np.linalg.norm(g)
    assert np.linalg.norm(H - H2) <= 1e-12 * np.linalg.norm(H)
    # a BDF2-style residual (eta = 2h/3, qA != q0) goes through the same code
    qA = q0 + 1e-3 * rng.normal(size=nr)
    qB = q0 + h * qd0 * 0.9
    g, H = o.eval_residual(q1, qA, qB, 2 * h / 3)
    g2, H2 = pw.eval_world(m, q1, qA, qB, 2 * h / 3)
    assert np.linalg.norm(g - g2) <= 1e-12 * np.linalg.norm(g)
    assert np.linalg.norm(H - H2) <= 1e-12 * np.linalg.norm(H)
    # energies
    o.set_state(q1, qd0)
    T, V = o.energy()
    T2, V2 = pw.energy_world(m, q1, qd0)
    assert abs(T - T2) <= 1e-12 * max(abs(T), 1) and abs(V - V2) <= 1e-12 * max(abs(V), 1)


@pytest.mark.parametrize("name", ["11", "chain6ground"])
def test_worldframe_contact_equals_tensor_formulation(oracle_lib, name):
    """Ground contact (ForceGroundCuboid.m:54-183): world-frame blocks inside the recursion == the oracle's literal
    J' (fm, Km, Dm) J assembly, on states where corners penetrate in both friction branches."""
    sc = scenesRedMax(11) if name == "11" else sceneChainGround(6, ground_z=-1.0)
    sc.init()
    d = sc.desc()
    o = oracle_lib.Oracle(d)
    m = pw.build_model(oracle_lib.lower_composite(d))    # scene 11's JointFree2D as its 1-DOF chain
    rng = np.random.default_rng(5)
    nr, h = o.nr, sc.h
    hits = 0
    for trial in range(6):
        if name == "11":
            q0 = np.array([rng.uniform(-1, 1), rng.uniform(-0.4, 0.4), 0.3])         # JointFree2D: x, y, theta
            qd0 = rng.normal(size=nr) * (50 if trial % 2 else 0.5)
        else:
            q0 = rng.uniform(-0.4, 0.4, nr)
            qd0 = rng.normal(size=nr) * (5 if trial % 2 else 0.05)
        q1 = q0 + h * qd0
        for eta, qA, qB in ((h, q0, q0 + h * qd0 * 0.5), (2 * h / 3, q0 + 1e-3 * rng.normal(size=nr), q0)):
            g, H = o.eval_residual(q1, qA, qB, eta)
            g2, H2 = pw.eval_world(m, q1, qA, qB, eta)
            assert np.linalg.norm(g - g2) <= 1e-12 * np.linalg.norm(g)
            assert np.linalg.norm(H - H2) <= 1e-12 * np.linalg.norm(H)
        o.set_state(q1, qd0)
        T, V = o.energy()
        T2, V2 = pw.energy_world(m, q1, qd0)
        assert abs(T - T2) <= 1e-12 * max(abs(T), 1) and abs(V - V2) <= 1e-12 * max(abs(V), 1)
        hits += pw.energy_world(m, q1, qd0)[1] != pw.energy_world(dict(m, contact=None), q1, qd0)[1]
    assert hits >= 3                                         # the contact branch really ran


def test_world_contact_blocks_are_the_congruence_of_the_body_blocks():
    """contact_world == Ad' (fm, Km, Dm) Ad of the literal body-frame ForceGroundCuboid.m:76-150 blocks."""
    rng = np.random.default_rng(0)
    g = {"E": se3.transform(R=se3.aaToMat([1, 0.2, 0], -1.2), p=[0.1, -0.2, 0.3]), "kn": 1e5, "kt": 1e2, "mu": 0.5, "kd": 3e1}
    m = {"ground": g, "sides": [np.array([3.0, 1.0, 2.0])]}
    seen = 0
    for trial in range(8):
        E = se3.transform(R=se3.aaToMat(rng.normal(size=3), rng.uniform(-2, 2)), p=rng.normal(size=3) * 0.5)
        phi_w = rng.normal(size=6) * (50 if trial % 2 else 1)
        Ad = se3.Ad(se3.inv(E))
        fm, Km, Dm, V = pw.contact_blocks_body(m, 0, E, Ad @ phi_w)
        F, K, D, V2 = pw.contact_world(m, 0, E, phi_w)
        seen += V > 0
        for a, b in ((F, Ad.T @ fm), (K, Ad.T @ Km @ Ad), (D, Ad.T @ Dm @ Ad)):
            assert np.linalg.norm(a - b) <= 1e-13 * max(np.linalg.norm(b), 1e-300)
        assert V == pytest.approx(V2, rel=1e-14)
    assert seen >= 4
