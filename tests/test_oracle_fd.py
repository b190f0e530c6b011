"""Finite-difference self-tests of the oracle's analytic derivatives, mirroring Scene.test
(matlab-diff/+redmax/Scene.m:224-378; pass criterion 1e-6 relative, Scene.printError :424-450)."""
import numpy as np
import pytest

from redmax_amd.scenes import sceneChain, sceneChainGround, scenesRedMax

SQE = np.sqrt(np.finfo(float).eps)


def _err(v0, v1):
    e = np.linalg.norm(v1 - v0)
    n0, n1 = np.linalg.norm(v0), np.linalg.norm(v1)
    if n0 > 1e-4 and n1 > 1e-4:
        e /= min(n0, n1)
    return e


@pytest.mark.parametrize("sid", [0, 1, 2, 3, "chain6"])
def test_scene_test_fd_identities(oracle_lib, sid):
    sc = sceneChain(6, axis=(0.2, 1.0, 0.1)) if sid == "chain6" else scenesRedMax(sid)
    sc.init()
    o = oracle_lib.Oracle(sc.desc())
    rng = np.random.default_rng(2)
    nr = o.nr
    q = rng.uniform(-0.8, 0.8, nr)
    qd = rng.uniform(-1, 1, nr)
    o.set_state(q, qd)
    J, Jdot, dJdq, dJdotdq = o.jacobian(deriv=True)
    M, f, dMdq, K, D = o.compute_values()
    # Jdot  (Scene.m:275-283)
    o.set_state(q + SQE * qd, qd)
    J_, _ = o.jacobian()
    assert _err((J_ - J) / SQE, Jdot) < 1e-6
    K_ = np.zeros((nr, nr))
    D_ = np.zeros((nr, nr))
    for i in range(nr):
        q_ = q.copy()
        q_[i] += SQE
        o.set_state(q_, qd)
        J_, Jdot_ = o.jacobian()
        assert _err((J_ - J) / SQE, dJdq[:, :, i]) < 1e-6            # dJ/dq   (:286-299)
        assert _err((Jdot_ - Jdot) / SQE, dJdotdq[:, :, i]) < 2e-6   # dJdot/dq
        M_, f_ = o.compute_values(deriv=False)
        assert _err((M_ - M) / SQE, dMdq[:, :, i]) < 1e-6            # dM/dq   (:302-313)
        # K, D (:343-376).  f is O(1e5..1e6) here, so the reference's one-sided sqrt(eps) difference carries
        # ~eps|f|/sqrt(eps) ~ 1e-2 of roundoff; central differences with a larger step test the same identity cleanly.
        hk, hd = 1e-5, 1e-4
        fp = []
        for sgn in (+1, -1):
            q_ = q.copy()
            q_[i] += sgn * hk
            o.set_state(q_, qd)
            fp.append(o.compute_values(deriv=False)[1])
        K_[:, i] = (fp[0] - fp[1]) / (2 * hk)
        fp = []
        for sgn in (+1, -1):
            qd_ = qd.copy()
            qd_[i] += sgn * hd
            o.set_state(q, qd_)
            fp.append(o.compute_values(deriv=False)[1])
        D_[:, i] = (fp[0] - fp[1]) / (2 * hd)
    assert _err(K_, K) < 1e-6
    assert _err(D_, D) < 1e-6


def test_newton_hessian_is_jacobian_of_g(oracle_lib):
    """testGrad of newton (driverRedMaxBDF1.m:104-115): H = dg/dq1."""
    sc = scenesRedMax(2)
    sc.init()
    o = oracle_lib.Oracle(sc.desc())
    rng = np.random.default_rng(4)
    nr, h = o.nr, sc.h
    q0, qd0 = rng.uniform(-0.5, 0.5, nr), rng.uniform(-1, 1, nr)
    q1 = q0 + h * qd0
    g, H = o.eval_bdf1(q1, q0, qd0, h)
    H_ = np.zeros_like(H)
    for i in range(nr):
        x = q1.copy()
        x[i] += SQE
        H_[:, i] = (o.eval_bdf1(x, q0, qd0, h, want_H=False) - g) / SQE
    assert _err(H_, H) < 1e-6


@pytest.mark.parametrize("two", [False, True])
def test_ground_contact_fd(oracle_lib, two):
    """ForceGroundCuboid's K and D against central differences of f (Scene.test K/D checks, Scene.m:343-376) on a chain
    whose corners penetrate the ground, plus H = dg/dq1 of the BDF1 residual through contact.  two: force objects with different
    frames and constants (a floor for the even bodies, a tilted softer plane for the odd ones)."""
    from redmax_amd.scenes import sceneChainTwoGrounds
    sc = sceneChainTwoGrounds(4, ground_z=-1.0) if two else sceneChainGround(4, ground_z=-1.0)
    sc.init()
    assert (sc.desc().get("ground_body") is not None) == two
    o = oracle_lib.Oracle(sc.desc())
    rng = np.random.default_rng(8)
    nr, h = o.nr, sc.h
    q = rng.uniform(0.1, 0.4, nr)
    qd = rng.uniform(-3, 3, nr)
    o.set_state(q, qd)
    M, f, dMdq, K, D = o.compute_values()
    _, V = o.energy()
    assert V > 0
    K_, D_ = np.zeros((nr, nr)), np.zeros((nr, nr))
    for i in range(nr):
        for A_, x, step in ((K_, 0, 1e-6), (D_, 1, 1e-5)):
            fp = []
            for sgn in (+1, -1):
                q_, qd_ = q.copy(), qd.copy()
                (q_ if x == 0 else qd_)[i] += sgn * step
                o.set_state(q_, qd_)
                fp.append(o.compute_values(deriv=False)[1])
            A_[:, i] = (fp[0] - fp[1]) / (2 * step)
    assert _err(K_, K) < 1e-6
    assert _err(D_, D) < 1e-6
    q1 = q + h * qd
    g, H = o.eval_bdf1(q1, q, qd, h)
    H_ = np.zeros_like(H)
    for i in range(nr):
        gp = []
        for sgn in (+1, -1):
            x = q1.copy()
            x[i] += sgn * 1e-7
            gp.append(o.eval_bdf1(x, q, qd, h, want_H=False))
        H_[:, i] = (gp[0] - gp[1]) / 2e-7
    assert _err(H_, H) < 1e-6


def test_per_body_ground_frames_equal_the_shared_frame_when_they_are_the_same(oracle_lib):
    """orc_set_ground_contact_body with n copies of one frame / parameter set = orc_set_ground_contact, bit for bit."""
    sc = sceneChainGround(5, ground_z=-1.0)
    sc.init()
    d = sc.desc()
    n = len(d["contact"])
    g = d["ground"]
    d2 = dict(d, ground_body={"E": np.stack([g["E"]] * n), "kn": np.full(n, g["kn"]), "kt": np.full(n, g["kt"]), "mu": np.full(n, g["mu"]),
                              "kd": np.full(n, g["kd"])})
    rng = np.random.default_rng(3)
    q = rng.uniform(0.1, 0.4, sc.nr)
    qd = rng.uniform(-3, 3, sc.nr)
    out = []
    for dd in (d, d2):
        o = oracle_lib.Oracle(dd)
        out.append(o.eval_bdf1(q + sc.h * qd, q, qd, sc.h))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_several_ground_forces_on_one_body(oracle_lib):
    """The reference keeps its force objects in a list (Force.m:26-56): a body may carry a floor AND a wall.  The oracle restates that
    literally (orc_add_ground_contact: both objects' fm, Km, Dm into the same body rows); the product's host mirror lists the second
    force as a fixed, massless child of the body's joint (Scene.desc(), what the C ABI takes: one force per listing entry).  Here, on the
    CPU: (i) the two descriptions give the same g, H, energies on the oracle (the lowering is exact), (ii) the literal H is the
    derivative of the literal g (central differences) at a state where floor and wall are both penetrated."""
    from redmax_amd.scenes import sceneChainFloorAndWall
    sc = sceneChainFloorAndWall(4, ground_z=-1.0)
    sc.init()
    d, lit = sc.desc(), sc.desc_literal()
    assert d["njoints"] == 8 and lit["njoints"] == 4 and len(lit["extra_forces"]) == 4
    o_lit, o_low = oracle_lib.Oracle(lit), oracle_lib.Oracle(d)
    assert o_lit.nr == o_low.nr == 4
    rng = np.random.default_rng(3)
    h = sc.h
    q = rng.uniform(0.1, 0.4, 4)
    qd = rng.uniform(-3, 3, 4)
    q1 = q + h * qd
    for o in (o_lit, o_low):
        o.set_state(q, qd)
    Vl, Vw = o_lit.energy()[1], o_low.energy()[1]
    o_floor = oracle_lib.Oracle(dict(lit, extra_forces=None))
    o_floor.set_state(q, qd)
    assert Vl > o_floor.energy()[1] > 0                 # both the floor and the wall are in play
    assert abs(Vl - Vw) <= 1e-12 * abs(Vl)
    g, H = o_lit.eval_bdf1(q1, q, qd, h)
    g2, H2 = o_low.eval_bdf1(q1, q, qd, h)
    assert _err(g2, g) < 1e-12 and _err(H2, H) < 1e-12
    H_ = np.zeros_like(H)
    for i in range(4):
        gp = []
        for sgn in (+1, -1):
            x = q1.copy()
            x[i] += sgn * 1e-7
            gp.append(o_lit.eval_bdf1(x, q, qd, h, want_H=False))
        H_[:, i] = (gp[0] - gp[1]) / 2e-7
    assert _err(H_, H) < 1e-6
