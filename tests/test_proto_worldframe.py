"""The algebra the HIP kernels implement (tests/proto_worldframe.py: world-frame recursive Newton-Euler with analytic
derivatives, no J / dJdq) equals the reference's literal tensor formulation (the oracle) to roundoff - on every
in-scope scene, skew axes, the 32-chain and a branching prismatic/revolute tree."""
import numpy as np
import pytest

import proto_worldframe as pw
from redmax_amd.scenes import sceneChain, scenesRedMax, sceneTree


def _scene(name):
    if name == "chain32":
        return sceneChain(32)
    if name == "chain8skew":
        return sceneChain(8, axis=(0.3, 1.0, 0.2))
    if name == "tree15":
        return sceneTree(15)
    return scenesRedMax(int(name))


@pytest.mark.parametrize("name", ["0", "1", "2", "3", "14", "chain8skew", "chain32", "tree15"])
def test_worldframe_equals_tensor_formulation(oracle_lib, name):
    sc = _scene(name)
    sc.init()
    d = sc.desc()
    o = oracle_lib.Oracle(d)
    m = pw.build_model(d)
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
