"""Adjoint BDF1 (BASELINE.json configs[3]) and adjoint BDF2 (driverRedMaxAdjointBDF2.m, scene 101) on the oracle.  The reference holds NO golden numbers for this
path: its only check is the finite-difference identity of taskObjective's testGrad switch
(driverRedMaxAdjointBDF1.m:46-61), reproduced here - "parity unpinned" beyond it."""
import numpy as np
import pytest

from redmax_amd.scenes import sceneAdjointChain, scenesRedMax


@pytest.mark.parametrize("n", [2, 5])
def test_adjoint_gradient_matches_finite_differences(oracle_lib, n):
    sc = scenesRedMax(100) if n == 2 else sceneAdjointChain(n)
    sc.init()
    o = oracle_lib.Oracle(sc.desc())
    rng = np.random.default_rng(1)
    p = 0.1 * rng.standard_normal(o.nr)
    nsteps = 20
    task = dict(sc.task, t=nsteps * sc.h)
    P, dPdp, st = o.adjoint_bdf1(sc.h, nsteps, task, p)
    assert st.diverged == 0 and st.not_converged == 0
    assert np.isfinite(P) and P > 0
    fd = np.zeros_like(p)
    eps = 1e-6
    for i in range(o.nr):
        pp, pm = p.copy(), p.copy()
        pp[i] += eps
        pm[i] -= eps
        fd[i] = (o.adjoint_bdf1(sc.h, nsteps, task, pp)[0] - o.adjoint_bdf1(sc.h, nsteps, task, pm)[0]) / (2 * eps)
    err = np.linalg.norm(fd - dPdp) / np.linalg.norm(fd)
    assert err < 1e-6, (err, fd, dPdp)      # Scene.printError threshold (Scene.m:425)


def test_adjoint_zero_parameters_is_plain_rollout(oracle_lib):
    """With p = 0 the forward pass is the unforced chain: the state after the call equals a line-search-free rollout."""
    sc = scenesRedMax(100)
    sc.init()
    o = oracle_lib.Oracle(sc.desc())
    task = dict(sc.task, t=10 * sc.h)
    P, dPdp, _ = o.adjoint_bdf1(sc.h, 10, task, np.zeros(o.nr))
    q_adj, _ = o.get_state()
    o2 = oracle_lib.Oracle(sc.desc())
    o2.step_bdf1(sc.h, 10)
    q_ref, _ = o2.get_state()
    assert np.linalg.norm(q_adj - q_ref) <= 1e-9 * np.linalg.norm(q_ref)


@pytest.mark.parametrize("n,nsteps", [(2, 20), (2, 100), (5, 12)])
def test_adjoint_bdf2_gradient_matches_finite_differences(oracle_lib, n, nsteps):
    """driverRedMaxAdjointBDF2.m's testGrad identity (:46-61) for TaskBDF2 / TaskBDF2PointPos (scene 101 and its n-link form): the
    reference drops dg/dqa of the SDIRK start step and uses the BDF2 dg/dp for every step (TaskBDF2.m:52-55, TaskBDF2PointPos.m:97-106),
    so its own gradient is exact only up to the start step's share, which decays like 1/nsteps: 25 % over 3 steps, 4.7 % over 20,
    0.8 % over the scene's own 100 (measured on this restatement, which keeps the reference's approximations: the tolerance is
    what the reference's formulas achieve).  The off-diagonal blocks of the BDF2 steps themselves are exact derivatives
    (d g_k / d q_{k-1} = -8/3 M + 8/9 h D etc.), which is why the error does not grow with the horizon."""
    sc = scenesRedMax(101) if n == 2 else sceneAdjointChain(n, bdf2=True)
    sc.init()
    o = oracle_lib.Oracle(sc.desc())
    rng = np.random.default_rng(2)
    p = 0.1 * rng.standard_normal(o.nr)
    task = dict(sc.task, t=nsteps * sc.h)
    P, dPdp, st = o.adjoint_bdf2(sc.h, nsteps, task, p)
    assert st.diverged == 0 and st.not_converged == 0 and np.isfinite(P) and P > 0
    fd = np.zeros_like(p)
    eps = 1e-6
    for i in range(o.nr):
        pp, pm = p.copy(), p.copy()
        pp[i] += eps
        pm[i] -= eps
        fd[i] = (o.adjoint_bdf2(sc.h, nsteps, task, pp)[0] - o.adjoint_bdf2(sc.h, nsteps, task, pm)[0]) / (2 * eps)
    err = np.linalg.norm(fd - dPdp) / np.linalg.norm(fd)
    print("adjoint BDF2 n=%d nsteps=%d: |fd - dPdp|/|fd| = %.3e" % (n, nsteps, err))
    assert err < 1.2 / nsteps, (err, fd, dPdp)


def test_adjoint_bdf2_zero_parameters_is_plain_bdf2_rollout(oracle_lib):
    sc = scenesRedMax(101)
    sc.init()
    o = oracle_lib.Oracle(sc.desc())
    P, dPdp, _ = o.adjoint_bdf2(sc.h, 10, dict(sc.task, t=10 * sc.h), np.zeros(o.nr))
    q_adj, _ = o.get_state()
    o2 = oracle_lib.Oracle(sc.desc())
    o2.step_bdf2(sc.h, 10)
    q_ref, _ = o2.get_state()
    assert np.linalg.norm(q_adj - q_ref) <= 1e-9 * np.linalg.norm(q_ref)
