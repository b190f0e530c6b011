"""Randomised parity: random trees (topology, joint types incl. the multi-DOF and Euler-chart ones, skew axes and planes, random
joint / body frames, cuboid sizes, joint springs / dampers / limits, optional ground contact) - HIP path vs the oracle on
(g, H), energies and a short BDF1 / BDF2 rollout.  Fixed seeds; tolerances as in test_gpu_parity.py."""
import math

import numpy as np
import pytest

from redmax_amd import se3
from redmax_amd.redmax import (BodyCuboid, ForceGroundCuboid, JointFixed, JointFree2D, JointFree3D, JointPlanar, JointPrismatic,
                               JointRevolute, JointSpherical, JointTranslational, JointUniversal, Scene)

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _random_scene(seed, contact=False, big=False):
    rng = np.random.default_rng(seed)
    sc = Scene()
    sc.h = 5e-3
    njoints = int(rng.integers(20, 30)) if big else int(rng.integers(3, 12))
    max_nodes = 62 if big else 30
    kinds = ["rev", "rev", "rev", "pri", "fix", "planar", "universal", "trans", "free2d", "sph", "free3d"]
    nodes = 0
    for i in range(njoints):
        kind = kinds[int(rng.integers(len(kinds)))] if i else kinds[int(rng.integers(3))]
        cost = {"planar": 2, "universal": 2, "trans": 3, "free2d": 3, "sph": 3, "free3d": 6}.get(kind, 1)
        if nodes + cost > max_nodes:
            kind, cost = "rev", 1
        nodes += cost
        body = BodyCuboid(float(rng.uniform(0.5, 2.0)), rng.uniform(0.5, 4.0, 3))
        # parent: a random earlier joint that keeps the listing depth-first (the last joint or one of its ancestors)
        parent = None
        if i:
            chain = [sc.joints[-1]]
            while chain[-1].parent is not None:
                chain.append(chain[-1].parent)
            parent = chain[int(rng.integers(len(chain)))]
        axis = rng.normal(size=3)
        if kind == "rev":
            j = JointRevolute(parent, body, axis)
        elif kind == "pri":
            j = JointPrismatic(parent, body, axis)
            j.setStiffness(float(rng.uniform(1e3, 1e4)))
        elif kind == "fix":
            j = JointFixed(parent, body)
        elif kind == "planar":
            j = JointPlanar(parent, body, rng.normal(size=(3, 2)))
            j.setStiffness(float(rng.uniform(1e3, 1e4)))
        elif kind == "universal":
            j = JointUniversal(parent, body)
        elif kind == "trans":
            j = JointTranslational(parent, body)
            j.setStiffness(float(rng.uniform(1e3, 1e4)))
        elif kind == "free2d":
            j = JointFree2D(parent, body)
            j.setStiffness(float(rng.uniform(1e3, 1e4)))
        elif kind == "sph":
            j = JointSpherical(parent, body)
        else:
            j = JointFree3D(parent, body)
            j.setStiffness(float(rng.uniform(1e3, 1e4)))
        j.setJointTransform(se3.transform(R=se3.aaToMat(rng.normal(size=3), rng.uniform(-1, 1)), p=rng.uniform(-3, 3, 3)))
        body.setBodyTransform(se3.transform(R=se3.aaToMat(rng.normal(size=3), rng.uniform(-1, 1)), p=rng.uniform(-2, 2, 3)))
        if rng.random() < 0.4:
            j.setDamping(float(rng.uniform(1e1, 1e3)))
        if kind == "rev" and rng.random() < 0.3:
            j.setLimitLower(-0.2)
            j.setLimitUpper(0.3)
            j.setLimitStiffness(1e5)
            j.setLimitDamping(1e2)
        if j.ndof:
            j.q[:j.ndof] = rng.uniform(-0.3, 0.3, j.ndof)
            j.qdot[:j.ndof] = rng.uniform(-1, 1, j.ndof)
        sc.bodies.append(body)
        sc.joints.append(j)
    if contact:
        for b in sc.bodies:
            if rng.random() < 0.6:
                f = ForceGroundCuboid(b)
                f.setTransform(se3.transform(p=[0, 0, -1.0]))
                f.setStiffness(1e5, 1e2)
                f.setDamping(3e1)
                f.setFriction(0.5)
                sc.forces.append(f)
    sc.init()
    return sc


@pytest.mark.parametrize("seed", list(range(100, 116)) + [200, 201, 202, 203])
def test_random_tree_matches_oracle(oracle_lib, seed):
    """seeds >= 200: 20-30 joints / 33-62 nodes (the 64-lane code paths, 201 and 203 with ground contact)."""
    from redmax_amd import BatchSim
    sc = _random_scene(seed, contact=(seed % 4 == 3) or seed == 201, big=seed >= 200)
    nr, h = sc.nr, sc.h
    rng = np.random.default_rng(seed + 1000)
    B = 2
    q0 = np.tile(sc.getQ()[0], (B, 1)) + rng.uniform(-0.05, 0.05, (B, nr))
    qd0 = np.tile(sc.getQ()[1], (B, 1)) + rng.uniform(-0.2, 0.2, (B, nr))
    q1 = q0 + h * qd0 + rng.uniform(-1e-3, 1e-3, (B, nr))
    sim = BatchSim(sc, batch=B)
    o = oracle_lib.Oracle(sc.desc())
    assert sim.nr == o.nr == nr
    for eta, qA, qB in ((h, q0, q0 + h * qd0), (2 * h / 3, q0 + 1e-3 * rng.normal(size=(B, nr)), q0 + 0.8 * h * qd0)):
        g, H = sim.eval_residual(q1, qA, qB, eta)
        for b in range(B):
            go, Ho = o.eval_residual(q1[b], qA[b], qB[b], eta)
            assert _rel(g[b], go) <= 1e-10, (seed, b)
            assert _rel(H[b], Ho) <= 1e-10, (seed, b)
    sim.set_state(q1, qd0)
    T, V = sim.energy()
    for b in range(B):
        o.set_state(q1[b], qd0[b])
        To, Vo = o.energy()
        assert abs(T[b] - To) <= 1e-10 * max(abs(To), 1) and abs(V[b] - Vo) <= 1e-10 * max(abs(Vo), 1)
    # the oracle's four rollouts (2 integrators x B) side by side on host threads (ctypes releases the GIL; every Oracle object owns its
    # scene): a state on which the reference's Newton creeps for its 10 nr iterations costs the test one such rollout, not four in a row
    def _oracle_rollout(integ, b):
        oo = oracle_lib.Oracle(sc.desc())
        oo.set_state(q0[b], qd0[b])
        st = (oo.step_bdf1 if integ == "bdf1" else oo.step_bdf2)(h, 6)
        return oo, st
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=2 * B) as ex:
        fut = {(integ, b): ex.submit(_oracle_rollout, integ, b) for integ in ("bdf1", "bdf2") for b in range(B)}
        orc = {k: f.result() for k, f in fut.items()}
    for integ in ("bdf1", "bdf2"):
        sim.set_state(q0, qd0)
        out = (sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2)(6, h=h, stats=True)
        qg, qdg = sim.get_state()
        charts = sim.charts()
        for b in range(B):
            oo, st = orc[(integ, b)]
            qo, qdo = oo.get_state()
            if st.diverged or st.not_converged:
                assert out["status"][b] & 3                 # the reference algorithm fails on this state: so must we
                continue
            assert out["status"][b] & 7 == 0, (seed, integ, b)
            assert list(charts[b]) == list(oo.charts())
            assert np.linalg.norm(qg[b] - qo) <= 1e-7 * np.linalg.norm(qo) + 1e-9, (seed, integ, b)
    sim.close()


@pytest.mark.parametrize("seed", [100, 103, 105, 107, 110, 111, 200, 201])
def test_random_tree_compute_values(oracle_lib, seed):
    """rmx_compute_values on the random trees above (every joint class, joint springs / dampers / limits, ground contact on the
    seeds with seed % 4 == 3 and on 201, 33..62 nodes from 200 on): the full [M, f, dMdq v, K, D] of computeValues
    (driverRedMaxBDF1.m:188-243) against the oracle's literal dense path, relative to |M| + |D| + |K|."""
    from redmax_amd import BatchSim
    sc = _random_scene(seed, contact=(seed % 4 == 3) or seed == 201, big=seed >= 200)
    nr = sc.nr
    rng = np.random.default_rng(seed + 2000)
    B = 2
    q = np.tile(sc.getQ()[0], (B, 1)) + rng.uniform(-0.05, 0.05, (B, nr))
    qd = np.tile(sc.getQ()[1], (B, 1)) + rng.uniform(-0.2, 0.2, (B, nr))
    v = 1e-2 * rng.standard_normal((B, nr))
    sim = BatchSim(sc, batch=B)
    out = sim.compute_values(q, qd, v=v)
    charts = sim.charts()
    for b in range(B):
        o = oracle_lib.Oracle(sc.desc())
        if len(charts[b]):
            o.set_charts(list(charts[b]))
        o.set_state(q[b], qd[b])
        Mo, fo, dMo, Ko, Do = o.compute_values(deriv=True)
        scale = np.linalg.norm(Mo) + np.linalg.norm(Do) + np.linalg.norm(Ko)
        assert np.linalg.norm(out["M"][b] - Mo) <= 1e-10 * scale, (seed, b)
        assert np.linalg.norm(out["f"][b] - fo) <= 1e-10 * max(np.linalg.norm(fo), 1.0), (seed, b)
        assert np.linalg.norm(out["D"][b] - Do) <= 1e-10 * scale, (seed, b)
        assert np.linalg.norm(out["K"][b] - Ko) <= 1e-9 * scale, (seed, b, np.linalg.norm(out["K"][b] - Ko) / scale)
        assert np.linalg.norm(out["dMv"][b] - np.einsum("rci,c->ri", dMo, v[b])) <= 1e-9 * scale, (seed, b)
    sim.close()
