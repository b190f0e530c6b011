"""Branching trees of 33..64 nodes, at most 7 levels deep: the guarded solve runs along the tree (tree_solve64 in rmx_device.h: a
multifrontal elimination, leaves first - no fill outside the root paths) instead of through the dense 64 x 64 block-column elimination.
Another elimination order, hence another rounding: the two agree to roundoff, and both against the oracle's dense partial-pivoting LU
(driverRedMaxBDF1.m:117).  RMX_TREE_SOLVE=0 (read at model creation) keeps the dense solve for every tree."""
import numpy as np
import pytest

from redmax_amd import se3
from redmax_amd.redmax import BodyCuboid, JointFixed, JointPrismatic, JointRevolute, Scene

pytestmark = pytest.mark.gpu


def _tree_scene(seed, n, max_depth, max_children, fixed=False):
    rng = np.random.default_rng(seed)
    # the shape first (a random parent among the nodes that can still take a child), then the depth-first listing the scene needs
    par, dep, nch = [-1], [0], [0]
    for i in range(1, n):
        ok = [k for k in range(i) if dep[k] < max_depth and nch[k] < max_children]
        p = ok[int(rng.integers(len(ok)))]
        par.append(p); dep.append(dep[p] + 1); nch.append(0); nch[p] += 1
    kids = [[k for k in range(n) if par[k] == i] for i in range(n)]
    order, stack = [], [0]
    while stack:
        i = stack.pop()
        order.append(i)
        stack.extend(reversed(kids[i]))
    sc = Scene()
    sc.h = 5e-3
    joint_of = {}
    for i in order:
        body = BodyCuboid(float(rng.uniform(0.5, 2.0)), rng.uniform(0.5, 3.0, 3))
        parent = joint_of[par[i]] if par[i] >= 0 else None
        kind = rng.random()
        if fixed and par[i] >= 0 and kind < 0.12:
            j = JointFixed(parent, body)                  # (a node without a DOF: unit diagonal, no coupling)
        elif kind < 0.75:
            j = JointRevolute(parent, body, rng.normal(size=3))
        else:
            j = JointPrismatic(parent, body, rng.normal(size=3))
            j.setStiffness(float(rng.uniform(1e3, 1e4)))
        j.setJointTransform(se3.transform(R=se3.aaToMat(rng.normal(size=3), rng.uniform(-1, 1)), p=rng.uniform(-3, 3, 3)))
        body.setBodyTransform(se3.transform(R=se3.aaToMat(rng.normal(size=3), rng.uniform(-1, 1)), p=rng.uniform(-2, 2, 3)))
        if rng.random() < 0.4:
            j.setDamping(float(rng.uniform(1e1, 1e3)))
        if j.ndof:
            j.q[:1] = rng.uniform(-0.3, 0.3, 1)
            j.qdot[:1] = rng.uniform(-1, 1, 1)
        joint_of[i] = j
        sc.bodies.append(body)
        sc.joints.append(j)
    sc.init()
    return sc, max(dep), max(nch)


def _run(sc, q0, qd0, integ, K, monkeypatch, tree):
    from redmax_amd import BatchSim
    monkeypatch.setenv("RMX_TREE_SOLVE", "1" if tree else "0")
    sim = BatchSim(sc, batch=q0.shape[0])
    sim.set_state(q0, qd0)
    out = (sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2)(K, h=sc.h, stats=True)
    q, qd = sim.get_state()
    sim.close()
    return q, qd, out


@pytest.mark.parametrize("seed,n,max_depth,max_children", [(1, 64, 7, 2), (2, 64, 5, 4), (3, 33, 3, 4), (4, 50, 6, 3), (5, 47, 7, 4), (6, 64, 3, 4)])
def test_tree_solve_agrees_with_the_dense_solve_and_the_oracle(oracle_lib, seed, n, max_depth, max_children, monkeypatch):
    sc, dmax, cmax = _tree_scene(seed, n, max_depth, max_children, fixed=seed % 2 == 0)      # even seeds: some JointFixed nodes
    assert 1 <= dmax <= 7 and cmax <= 4 and sc.nr <= n
    n = sc.nr
    rng = np.random.default_rng(seed + 50)
    B, K = 3, 6
    q0 = np.tile(sc.getQ()[0], (B, 1)) + rng.uniform(-0.05, 0.05, (B, n))
    qd0 = np.tile(sc.getQ()[1], (B, 1)) + rng.uniform(-0.2, 0.2, (B, n))
    for integ in ("bdf1", "bdf2"):
        qt, qdt, ot = _run(sc, q0, qd0, integ, K, monkeypatch, True)
        qd_, qdd, od = _run(sc, q0, qd0, integ, K, monkeypatch, False)
        assert np.isfinite(qt).all() and (ot["status"] & 16 == 0).all()            # no solve fell back to the pivot search
        assert np.abs(qt - qd_).max() <= 1e-10 * np.abs(qd_).max() and np.abs(qdt - qdd).max() <= 1e-8 * max(np.abs(qdd).max(), 1.0)
        assert np.array_equal(ot["status"], od["status"])
        assert np.abs(ot["newton_iters"].astype(int) - od["newton_iters"].astype(int)).max() <= 1
        for b in range(B):
            o = oracle_lib.Oracle(sc.desc())
            o.set_state(q0[b], qd0[b])
            st = (o.step_bdf1 if integ == "bdf1" else o.step_bdf2)(sc.h, K)
            if st.diverged or st.not_converged:
                assert ot["status"][b] & 3
                continue
            qo, _ = o.get_state()
            assert ot["status"][b] & 7 == 0
            assert np.linalg.norm(qt[b] - qo) <= 1e-7 * np.linalg.norm(qo) + 1e-9, (seed, integ, b)


def test_trees_the_solve_does_not_take_keep_the_dense_solve(monkeypatch):
    """deeper than 7 levels, or a node with more than 4 children: the model carries no tree for the solve - same bits either way"""
    for seed, n, md, mc in ((11, 40, 12, 2), (12, 40, 3, 7)):
        sc, dmax, cmax = _tree_scene(seed, n, md, mc)
        assert dmax > 7 or cmax > 4
        rng = np.random.default_rng(seed)
        q0 = np.tile(sc.getQ()[0], (2, 1)) + rng.uniform(-0.05, 0.05, (2, n))
        qd0 = np.tile(sc.getQ()[1], (2, 1))
        a = _run(sc, q0, qd0, "bdf1", 4, monkeypatch, True)
        b = _run(sc, q0, qd0, "bdf1", 4, monkeypatch, False)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
