"""Adjoint BDF1 (BASELINE.json configs[3]) on the oracle.  The reference holds NO golden numbers for this
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
