"""Trees of more than 64 nodes: the one-workgroup-per-trajectory kernels (redmax_amd/csrc/rmx_big.hip).  The reference has no size
limit (the one-wavefront kernels stop at 64 nodes, a JointFree3D body being six of them): a 128-link chain, a 150-joint
revolute / prismatic binary tree and a scene of 20 free-flying bodies (JointFree3D: Euler charts with switching) against the oracle -
single evaluations (g, H), BDF1 and BDF2 rollouts with per-step energies, Newton counts."""
import math

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _free_bodies(nb=20):
    """A fixed hub carrying nb free-flying cuboids (JointFree3D each): 1 + 6 nb nodes after lowering, nb Euler-chart groups."""
    from redmax_amd.redmax import BodyCuboid, JointFixed, JointFree3D, Scene
    from redmax_amd.scenes import _T
    sc = Scene()
    sc.name = "%d free bodies" % nb
    sc.h = 2e-2
    sc.grav = np.array([0.0, 0.0, -9.8])
    hub = BodyCuboid(1.0, [1, 1, 1])
    j0 = JointFixed(None, hub)
    j0.setJointTransform(np.eye(4))
    hub.setBodyTransform(np.eye(4))
    sc.bodies.append(hub)
    sc.joints.append(j0)
    rng = np.random.default_rng(5)
    for i in range(nb):
        b = BodyCuboid(1.0, [1.0 + 0.1 * (i % 3), 0.8, 0.6 + 0.05 * (i % 4)])
        j = JointFree3D(j0, b)
        j.setJointTransform(_T([3.0 * (i % 5), 3.0 * (i // 5), 0.0]))
        b.setBodyTransform(np.eye(4))
        j.q[:] = np.concatenate([rng.uniform(-0.2, 0.2, 3), rng.uniform(-0.4, 0.4, 3)])
        j.qdot[:] = np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-2.5, 2.5, 3)])       # fast spins: charts will switch
        sc.bodies.append(b)
        sc.joints.append(j)
    return sc


def _binary_tree(n=150, depth=7):
    """A binary tree of n revolute (axes cycling y, x, z) and prismatic (every 7th, with a spring) joints, listed depth-first."""
    from redmax_amd.redmax import BodyCuboid, JointPrismatic, JointRevolute, Scene
    from redmax_amd.scenes import _T
    sc = Scene()
    sc.name = "%d-joint binary tree" % n
    axes = ([0, 1, 0], [1, 0, 0], [0, 0, 1])
    left = [n]

    def rec(par, side, d):
        if left[0] <= 0:
            return
        i = len(sc.joints)
        left[0] -= 1
        b = BodyCuboid(1.0, [4, 1, 1])
        if i % 7 == 5:
            j = JointPrismatic(par, b, [1, 0, 0])
            j.setStiffness(1e3)
        else:
            j = JointRevolute(par, b, axes[i % 3])
        j.setJointTransform(np.eye(4) if par is None else _T([4, 0.5 * side, 0]))
        b.setBodyTransform(_T([2, 0, 0]))
        j.q[0] = 0.1 * math.sin(i)
        sc.bodies.append(b)
        sc.joints.append(j)
        if d > 0:
            rec(j, 1, d - 1)
            rec(j, -1, d - 1)
    rec(None, 0, depth)
    return sc


def _scene(name):
    from redmax_amd.scenes import sceneChain
    if name == "chain128":
        return sceneChain(128)
    if name == "tree150":
        return _binary_tree(150)
    return _free_bodies(20)


@pytest.mark.parametrize("name", ["chain128", "tree150", "free20"])
def test_big_tree_eval_matches_oracle(oracle_lib, name):
    """g and H of one evaluation against the literal oracle (its O(n^4) tensor path takes seconds per evaluation at these sizes: one
    trajectory) and against the tensor-free CPU code (fixed / revolute / prismatic joints only)."""
    from redmax_amd import BatchSim
    sc = _scene(name)
    sc.init()
    B = 2
    rng = np.random.default_rng(3)
    qs, qds = sc.getQ()
    h = sc.h
    q0 = qs[None, :] + rng.uniform(-0.1, 0.1, (B, sc.nr))
    qd0 = qds[None, :] + rng.uniform(-0.5, 0.5, (B, sc.nr))
    q1 = q0 + h * qd0 + rng.uniform(-1e-3, 1e-3, (B, sc.nr))
    sim = BatchSim(sc, batch=B)
    g, H = sim.eval_bdf1(q1, q0, qd0, h)
    g1 = sim.eval_bdf1(q1, q0, qd0, h, want_H=False)
    assert np.array_equal(g, g1)
    sim.close()
    o = oracle_lib.Oracle(sc.desc())
    go, Ho = o.eval_bdf1(q1[0], q0[0], qd0[0], h)
    assert _rel(g[0], go) <= 1e-10 and _rel(H[0], Ho) <= 1e-10, (name, _rel(g[0], go), _rel(H[0], Ho))
    if name != "free20":
        for b in range(B):
            gt, Ht = oracle_lib.tensorfree_eval(sc.desc(), q1[b], q0[b], q0[b] + h * qd0[b], h)
            assert _rel(g[b], gt) <= 1e-11 and _rel(H[b], Ht) <= 1e-11, (name, b)


@pytest.mark.parametrize("name,K,tol", [("chain128", 8, 1e-7), ("tree150", 8, 1e-9)])
def test_big_tree_bdf1_rollout_matches_tensor_free(oracle_lib, name, K, tol):
    """BDF1 rollouts of a 128-link chain and a 150-joint binary tree against the tensor-free CPU code in the same iterate mode (the literal
    oracle needs minutes per evaluation here and, on plain doubles, cannot meet tol = 1e-9 on a 1280 cm cgs chain at all: the
    lattice of doubles): final state, Newton counts, statuses.  The 1280 cm chain runs tol = 1e-7: |g| starts at
    ~1e6 there and its evaluation noise reaches 1e-9, where iteration counts depend on the last bits of either side."""
    from redmax_amd import BatchSim
    sc = _scene(name)
    sc.init()
    B = 3
    rng = np.random.default_rng(8)
    qs, qds = sc.getQ()
    q = np.ascontiguousarray(qs[None, :] + rng.uniform(-0.05, 0.05, (B, sc.nr)))
    qd = np.ascontiguousarray(qds[None, :] + rng.uniform(-0.1, 0.1, (B, sc.nr)))
    sim = BatchSim(sc, batch=B)
    sim.opts.tol = tol
    sim.set_state(q, qd)
    out = sim.step_bdf1(K, h=sc.h, stats=True)
    qg, qdg = sim.get_state()
    sim.close()
    qc, qdc = q.copy(), qd.copy()
    ref = oracle_lib.tensorfree_batch_step_bdf1(sc.desc(), qc, qdc, sc.h, K, nthreads=4, tol=tol, compensated=True)
    assert (out["status"] & 15 == 0).all() and (ref["status"] & 15 == 0).all()
    for b in range(B):
        assert _rel(qg[b], qc[b]) <= 1e-9 and _rel(qdg[b], qdc[b]) <= 1e-7, (name, b, _rel(qg[b], qc[b]))
    assert np.abs(out["newton_iters"] - ref["newton_iters"]).max() <= 1, (out["newton_iters"], ref["newton_iters"])


def test_chain72_bdf2_matches_literal_oracle(oracle_lib):
    """Just above the one-wavefront limit the literal oracle is still affordable: a 72-link chain, SDIRK2 start step + BDF2, per-step
    energies.  (tol = 1e-7 on both sides: a 720 cm cgs chain on plain doubles - the oracle's arithmetic - sits on the lattice at 1e-9.)"""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChain
    sc = sceneChain(72)
    sc.init()
    K, tol = 4, 1e-7
    rng = np.random.default_rng(2)
    q = rng.uniform(-0.05, 0.05, (1, sc.nr))
    qd = rng.uniform(-0.1, 0.1, (1, sc.nr))
    sim = BatchSim(sc, batch=1)
    sim.opts.tol = tol
    sim.set_state(q, qd)
    out = sim.step_bdf2(K, h=sc.h, stats=True, history=True)
    qg, qdg = sim.get_state()
    sim.close()
    oracle_lib.set_newton(tol=tol)
    o = oracle_lib.Oracle(sc.desc())
    o.set_state(q[0], qd[0])
    st, To, Vo = o.step_bdf2(sc.h, K, history=True)
    oracle_lib.set_newton()
    qo, qdo = o.get_state()
    assert st.diverged == 0 and st.not_converged == 0 and (out["status"] & 15 == 0).all()
    assert _rel(qg[0], qo) <= 1e-8 and _rel(qdg[0], qdo) <= 1e-6
    assert np.abs(out["T"][:, 0] + out["V"][:, 0] - To - Vo).max() <= 1e-8 * (np.abs(To + Vo).max() + 1.0)
    assert abs(int(out["newton_iters"][0]) - st.newton_iters) <= 1


def test_chain72_bdf1_rollout_matches_literal_oracle(oracle_lib):
    """The BDF1 rollout of a tree of more than 64 nodes against the LITERAL oracle (the tensor-free code of the test above is the
    kernels' own algorithm): 72-link chain, 6 steps of the reference's simLoop, 2 rollouts, per-step energies, final state, Newton
    counts.  tol = 1e-7 on both sides as in the BDF2 test below (the lattice of doubles at 1e-9 on a 720 cm chain)."""
    from redmax_amd import BatchSim, syntheticStates
    from redmax_amd.scenes import sceneChain
    sc = sceneChain(72)
    sc.init()
    B, K, tol = 2, 6, 1e-7
    q, qd = syntheticStates(sc.nr, B, first=40)
    sim = BatchSim(sc, batch=B)
    sim.opts.tol = tol
    sim.set_state(q, qd)
    out = sim.step_bdf1(K, h=sc.h, stats=True, history=True)
    qg, qdg = sim.get_state()
    sim.close()
    oracle_lib.set_newton(tol=tol)
    try:
        for b in range(B):
            o = oracle_lib.Oracle(sc.desc())
            o.set_state(q[b], qd[b])
            st, To, Vo = o.step_bdf1(sc.h, K, history=True)
            qo, qdo = o.get_state()
            assert st.diverged == 0 and st.not_converged == 0 and out["status"][b] & 15 == 0
            assert _rel(qg[b], qo) <= 1e-8 and _rel(qdg[b], qdo) <= 1e-6, (b, _rel(qg[b], qo), _rel(qdg[b], qdo))
            assert np.abs(out["T"][:, b] + out["V"][:, b] - To - Vo).max() <= 1e-8 * (np.abs(To + Vo).max() + 1.0)
            assert abs(int(out["newton_iters"][b]) - st.newton_iters) <= 2
    finally:
        oracle_lib.set_newton()


@pytest.mark.parametrize("integ,nbodies", [("bdf1", 12), ("bdf2", 20)] + ([("bdf1", 20)] if os.environ.get("RMX_FULL_TESTS") == "1" else []))
def test_free_bodies_rollout_matches_oracle(oracle_lib, integ, nbodies):
    """20 (12) free-flying bodies (JointFree3D: 121 (73) nodes after lowering, 20 (12) Euler-chart groups), fast spins so that charts
    switch inside the rollout: final state, per-step energies, Newton counts and the charts themselves against the literal oracle
    (whose O(n^4) tensor path takes 27 s for the 20-body rollout: BDF1 runs the 12-body scene unless RMX_FULL_TESTS=1)."""
    from redmax_amd import BatchSim
    sc = _free_bodies(nbodies)
    sc.h = 2e-2
    for j in sc.joints[1:]:
        j.qdot[3:] *= 2.0              # up to 5 rad/s: |det T| <= 0.5 within a dozen steps
    sc.init()
    K = 14
    qs, qds = sc.getQ()
    sim = BatchSim(sc, batch=1)
    sim.set_state(qs[None, :], qds[None, :])
    out = (sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2)(K, h=sc.h, stats=True, history=True)
    qg, qdg = sim.get_state()
    charts = sim.charts()
    sim.close()
    o = oracle_lib.Oracle(sc.desc())
    o.set_state(qs, qds)
    st, To, Vo = (o.step_bdf1 if integ == "bdf1" else o.step_bdf2)(sc.h, K, history=True)
    qo, qdo = o.get_state()
    assert st.diverged == 0 and st.not_converged == 0 and (out["status"] & 15 == 0).all()
    assert st.chart_switches > 0 and (out["status"] & 32).all()          # charts did switch
    assert np.array_equal(charts[0], o.charts())
    assert _rel(qg[0], qo) <= 1e-8 and _rel(qdg[0], qdo) <= 1e-6, (_rel(qg[0], qo), _rel(qdg[0], qdo))
    assert np.abs(out["T"][:, 0] + out["V"][:, 0] - To - Vo).max() <= 1e-8 * (np.abs(To + Vo).max() + 1.0)
    assert int(out["newton_iters"][0]) == st.newton_iters


def test_more_than_21_euler_chart_joints(oracle_lib):
    """JointSpherical.m has no cap on the number of spherical joints of a scene; up to ABI 108 the library refused more than 21 (the size of
    the one-wavefront kernels' chart tables) although trees of 256 nodes are accepted.  26 free-flying bodies (157 nodes, 26 Euler-chart
    groups): g, H against the literal oracle, and a BDF1 rollout with fast spins - charts, final state, Newton counts."""
    from redmax_amd import BatchSim
    sc = _free_bodies(26)
    sc.h = 2e-2
    for j in sc.joints[1:]:
        j.qdot[3:] *= 2.0
    sc.init()
    qs, qds = sc.getQ()
    h = sc.h
    rng = np.random.default_rng(2)
    q1 = qs + h * qds + 1e-3 * rng.standard_normal(sc.nr)
    sim = BatchSim(sc, batch=1)
    assert sim.nsph == 26
    g, H = sim.eval_bdf1(q1[None, :], qs[None, :], qds[None, :], h)
    o = oracle_lib.Oracle(sc.desc())
    go, Ho = o.eval_bdf1(q1, qs, qds, h)
    assert _rel(g[0], go) <= 1e-10 and _rel(H[0], Ho) <= 1e-10
    K = 5
    sim.set_state(qs[None, :], qds[None, :])
    out = sim.step_bdf1(K, h=h, stats=True)
    qg, qdg = sim.get_state()
    charts = sim.charts()
    sim.close()
    o.set_state(qs, qds)
    st = o.step_bdf1(h, K)
    qo, qdo = o.get_state()
    assert st.diverged == 0 and st.not_converged == 0 and (out["status"] & 15 == 0).all()
    assert np.array_equal(charts[0], o.charts())
    assert _rel(qg[0], qo) <= 1e-8 and _rel(qdg[0], qdo) <= 1e-6
    assert int(out["newton_iters"][0]) == st.newton_iters


def test_big_tree_refusals():
    """What the one-workgroup kernels do not cover is refused loudly, not silently dropped."""
    from redmax_amd import BatchSim
    from redmax_amd._abi import RedMaxHipError
    from redmax_amd.scenes import sceneChain, sceneChainGround
    sc = sceneChain(100)
    sc.init()
    sim = BatchSim(sc, batch=1)
    with pytest.raises(RedMaxHipError):
        sim.step_euler(1, 1e-2)
    sim.close()
    big = sceneChain(300)
    big.init()
    with pytest.raises(RedMaxHipError):
        BatchSim(big, batch=1)


def test_compute_values_on_a_big_tree(oracle_lib):
    """computeValues' full output (driverRedMaxBDF1.m:188-243) for a tree of more than 64 nodes: no M / D kernel exists at that size, so
    rmx_compute_values (and rmx_eval_mfd through it) takes M, D, K apart from H(eta; v = 0) at eta = 1, 2, 1/2 and dMdq v from one more
    evaluation.  72-link chain against the literal oracle (its dense tensor path), relative to |M| + |D| + |K|."""
    from redmax_amd import BatchSim, syntheticStates
    from redmax_amd.scenes import sceneChain
    sc = sceneChain(72)
    for j in sc.joints:                      # some joint damping and stiffness, so that D and K have more in them than the Coriolis terms
        j.setDamping(3.0e2)
        j.setStiffness(2.0e3)
    sc.init()
    B = 2
    q, qd = syntheticStates(sc.nr, B, first=3)
    rng = np.random.default_rng(9)
    v = rng.standard_normal((B, sc.nr)) * 1e-2
    sim = BatchSim(sc, batch=B)
    out = sim.compute_values(q, qd, v=v)
    M2, f2, D2 = sim.eval_mfd(q, qd)
    for b in range(B):
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(q[b], qd[b])
        Mo, fo, dMo, Ko, Do = o.compute_values(deriv=True)
        scale = np.linalg.norm(Mo) + np.linalg.norm(Do) + np.linalg.norm(Ko)
        assert np.linalg.norm(out["M"][b] - Mo) <= 1e-10 * scale and np.linalg.norm(out["f"][b] - fo) <= 1e-10 * np.linalg.norm(fo)
        assert np.linalg.norm(out["D"][b] - Do) <= 1e-10 * scale and np.linalg.norm(out["K"][b] - Ko) <= 1e-10 * scale
        assert np.linalg.norm(out["dMv"][b] - np.einsum("rci,c->ri", dMo, v[b])) <= 1e-10 * scale
        assert np.array_equal(M2[b], out["M"][b]) and np.array_equal(D2[b], out["D"][b]) and np.array_equal(f2[b], out["f"][b])
    sim.close()


def _ground_states(sc, B, rng, depth=0.6):
    """States of a chain over the ground with bodies IN the ground: the chain lies along x at the joints' height, so a floor just below
    z = 0 is penetrated by the lower corners of every cuboid; random joint angles and velocities on top."""
    q = 0.02 * rng.standard_normal((B, sc.nr))
    qd = 0.5 * rng.standard_normal((B, sc.nr))
    return q, qd


def test_ground_contact_on_a_big_tree_eval(oracle_lib):
    """ForceGroundCuboid on a tree of more than 64 nodes (the reference has no size limit, ForceGroundCuboid.m:54-153): g and H of a
    72-link chain whose bodies penetrate a frictional floor, both friction branches, against the literal oracle."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChainGround
    sc = sceneChainGround(72, ground_z=-0.3)
    sc.init()
    B = 2
    rng = np.random.default_rng(4)
    q0, qd0 = _ground_states(sc, B, rng)
    h = sc.h
    q1 = q0 + h * qd0 + 1e-4 * rng.standard_normal((B, sc.nr))
    sim = BatchSim(sc, batch=B)
    g, H = sim.eval_residual(q1, q0, q0 + h * qd0, h)
    o_free = oracle_lib.Oracle(dict(sc.desc(), contact=None))
    for b in range(B):
        o = oracle_lib.Oracle(sc.desc())
        go, Ho = o.eval_residual(q1[b], q0[b], q0[b] + h * qd0[b], h)
        assert _rel(g[b], go) <= 1e-10 and _rel(H[b], Ho) <= 1e-10, (b, _rel(g[b], go), _rel(H[b], Ho))
        gf, Hf = o_free.eval_residual(q1[b], q0[b], q0[b] + h * qd0[b], h)
        assert _rel(gf, go) > 1e-3 and _rel(Hf, Ho) > 1e-6          # the contact terms really are in g and H
    sim.close()


@pytest.mark.parametrize("integ", ["bdf1", "bdf2"])
def test_ground_contact_on_a_big_tree_rollout(oracle_lib, integ):
    """A 72-link chain dropped onto the floor, a few steps of BDF1 / BDF2 (scene 11's step) through touch-down against the literal
    oracle: final state, Newton counts."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChainGround
    sc = sceneChainGround(72, ground_z=-0.52)
    sc.init()
    B = 1
    rng = np.random.default_rng(12)
    q = 0.01 * rng.standard_normal((B, sc.nr))
    qd = 0.2 * rng.standard_normal((B, sc.nr))
    nsteps = 3
    sim = BatchSim(sc, batch=B)
    sim.set_state(q, qd)
    out = (sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2)(nsteps, h=sc.h, stats=True)
    qg, qdg = sim.get_state()
    sim.close()
    o = oracle_lib.Oracle(sc.desc())
    o.set_state(q[0], qd[0])
    st = o.step_bdf1(sc.h, nsteps) if integ == "bdf1" else o.step_bdf2(sc.h, nsteps)
    qo, qdo = o.get_state()
    assert (out["status"] & 15 == 0).all()
    assert _rel(qg[0], qo) <= 1e-8 and _rel(qdg[0], qdo) <= 1e-6, (_rel(qg[0], qo), _rel(qdg[0], qdo))
    assert abs(int(out["newton_iters"][0]) - int(st.newton_iters)) <= 1
    of = oracle_lib.Oracle(dict(sc.desc(), contact=None))
    of.set_state(q[0], qd[0])
    (of.step_bdf1 if integ == "bdf1" else of.step_bdf2)(sc.h, nsteps)
    assert _rel(of.get_state()[1], qdo) > 1e-3                 # the floor really was felt


@pytest.mark.parametrize("name", ["chain72", "chain128", "tree150", "chain200", "free20"])
def test_big_tree_lu_modes_agree(name):
    """The two linear solves of the large-tree kernels - lu_mode 0: blocked elimination on the diagonal under the multiplier guard
    (big_solve_diag: H in LDS with 16-column panels at 72 / 128 links and 20 free bodies, H in HBM with 32-column panels at 150 / 200 DOFs),
    lu_mode 1: partial pivoting always (MATLAB mldivide, big_solve / big_solve_blocked) - give the same rollout to rounding, with the
    same Newton iteration counts."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChain
    sc = sceneChain(int(name[5:])) if name.startswith("chain") else _scene(name)
    sc.init()
    B, K = 4, 6
    rng = np.random.default_rng(21)
    qs, qds = sc.getQ()
    q = np.ascontiguousarray(qs[None, :] + rng.uniform(-0.05, 0.05, (B, sc.nr)))
    qd = np.ascontiguousarray(qds[None, :] + rng.uniform(-0.1, 0.1, (B, sc.nr)))
    res = []
    for mode in (0, 1):
        sim = BatchSim(sc, batch=B)
        sim.opts.tol = 1e-7 if name.startswith("chain") else 1e-9
        sim.opts.lu_mode = mode
        sim.set_state(q, qd)
        out = sim.step_bdf1(K, h=sc.h, stats=True)
        res.append((sim.get_state(), out))
        sim.close()
    (s0, o0), (s1, o1) = res
    assert (o0["status"] & 15 == 0).all() and (o1["status"] & 15 == 0).all()
    assert (o1["status"] & 16 == 0).all()            # RMX_ST_PIVOTED reports a FALLBACK: never in lu_mode 1
    for b in range(B):
        assert _rel(s0[0][b], s1[0][b]) <= 1e-9 and _rel(s0[1][b], s1[1][b]) <= 1e-7, (name, b, _rel(s0[0][b], s1[0][b]))
    assert np.abs(o0["newton_iters"] - o1["newton_iters"]).max() <= 1, (o0["newton_iters"], o1["newton_iters"])


def test_big_tree_guard_falls_back_to_pivoting():
    """The 256-link chain at U(-0.1, 0.1) (tools/big_tree_bench.py): some of its rollouts whip hard enough for a multiplier of the
    equilibrated H to pass the guard's bound (profiles/r05d_big_profile_diag_solver.txt: blocks 3, 8, 15, 17, 21 within ten steps).  Those
    solves are redone with partial pivoting (RMX_ST_PIVOTED) and the rollout goes on as it does with pivoting throughout: every
    rollout that converges in both modes ends in the same state."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChain, syntheticStates
    sc = sceneChain(256)
    sc.init()
    B, K = 32, 10
    q, qd = syntheticStates(sc.nr, B)
    res = []
    for mode in (0, 1):
        sim = BatchSim(sc, batch=B)
        sim.opts.tol = 1e-6
        sim.opts.lu_mode = mode
        sim.set_state(q, qd)
        out = sim.step_bdf1(K, h=1e-2, stats=True)
        res.append((sim.get_state(), out))
        sim.close()
    (s0, o0), (s1, o1) = res
    assert (o0["status"] & 16 != 0).any(), "no solve tripped the guard: the fallback went untested"
    assert (o1["status"] & 16 == 0).all()
    assert np.isfinite(s0[0]).all() and np.isfinite(s0[1]).all()
    good = ((o0["status"] & 15) == 0) & ((o1["status"] & 15) == 0)
    assert good.sum() >= B - 4, (o0["status"], o1["status"])
    for b in np.nonzero(good)[0]:
        assert _rel(s0[0][b], s1[0][b]) <= 1e-6, (b, _rel(s0[0][b], s1[0][b]), int(o0["status"][b]))
