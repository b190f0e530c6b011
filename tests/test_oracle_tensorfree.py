"""oracle/redmax_tensorfree.c ("Baseline B"): the tensor-free CPU implementation that bench.py times next to
the literal restatement must compute the same thing.  It is checked against the KAT-pinned literal oracle: g and H of the
generic implicit residual to 1e-11, BDF1 rollouts to 1e-9 with the same Newton iteration counts."""
import numpy as np
import pytest

from oracle import oracle as orc
from redmax_amd.scenes import sceneChain, scenesRedMax, sceneTree, syntheticStates


def _scene(name):
    if name == "chain32":
        return sceneChain(32)
    if name == "chain8skew":
        return sceneChain(8, axis=(0.3, 1.0, 0.2))
    if name == "tree15":
        return sceneTree(15)
    return scenesRedMax(int(name))


@pytest.mark.parametrize("name", ["0", "1", "2", "3", "14", "chain8skew", "tree15", "chain32"])
def test_eval_matches_the_literal_oracle(name):
    sc = _scene(name)
    sc.init()
    rng = np.random.default_rng(3)
    q0, qd0 = sc.getQ()
    for trial in range(3):
        q = q0 + 0.3 * rng.standard_normal(sc.nr)
        qA = q - 1e-2 * (qd0 + rng.standard_normal(sc.nr))
        qB = qA + 1e-3 * rng.standard_normal(sc.nr)
        eta = [1e-2, 2e-2 / 3, 3e-3][trial]
        o = orc.Oracle(sc.desc())
        go, Ho = o.eval_residual(q, qA, qB, eta)
        g, H = orc.tensorfree_eval(sc.desc(), q, qA, qB, eta)
        assert np.linalg.norm(g - go) <= 1e-11 * np.linalg.norm(go), (name, trial)
        assert np.linalg.norm(H - Ho) <= 1e-11 * np.linalg.norm(Ho), (name, trial)
        g1 = orc.tensorfree_eval(sc.desc(), q, qA, qB, eta, want_H=False)
        assert np.array_equal(g1, g)


@pytest.mark.parametrize("name,nsteps", [("2", 30), ("14", 40), ("chain32", 5)])
def test_rollout_matches_the_literal_oracle(name, nsteps):
    sc = _scene(name)
    sc.init()
    B = 3
    q, qd = syntheticStates(sc.nr, B, first=2)
    if name != "chain32":
        s0, sd0 = sc.getQ()
        q, qd = q + s0, qd + sd0
    qa, qda = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
    qb, qdb = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
    ca = orc.batch_step_bdf1(sc.desc(), qa, qda, sc.h, nsteps, nthreads=2, counters=True)
    cb = orc.tensorfree_batch_step_bdf1(sc.desc(), qb, qdb, sc.h, nsteps, nthreads=2)
    for b in range(B):
        assert np.linalg.norm(qa[b] - qb[b]) <= 1e-9 * np.linalg.norm(qa[b]) + 1e-10, (name, b)
    if name != "chain32":          # on the 320 cm chain |g| < 1e-9 is decided by roundoff: counts may differ
        assert np.array_equal(ca["newton_iters"], cb["newton_iters"])
        assert (cb["status"] == 0).all() and (ca["bad"] == 0).all()


def test_compensated_iterate_converges_where_plain_doubles_stick():
    """The 32-link chain at the reference's tol = 1e-9 over the reference's 100 steps (CPU side of tests/test_gpu_reference_tol.py):
    on plain doubles the world-frame evaluation sticks on the lattice of doubles on a sizeable fraction of the trajectory-steps (the
    literal oracle escapes through its own evaluation noise); with the compensated iterate x + xlo - the mode
    the HIP kernels run by default - every step converges in no more iterations than the literal oracle needs, to the same state."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from reference_tol import run_stats
    r = run_stats(8, 100, tol=1e-9, gpu=False, first=1)
    tf, tfp, lit = r["tensor_free"], r["tensor_free_plain"], r["literal"]
    assert tf["bad_frac"] == 0.0 and tf["halvings_per_step"] <= 0.01
    assert tfp["bad_frac"] >= 0.03 and tfp["bad_frac_second_half"] >= 5 * tfp["bad_frac_first_half"]
    assert lit["bad_frac"] <= 0.02
    assert tf["iters_per_step"] <= lit["iters_per_step"] + 0.05
    for a in (tf, tfp):
        e = np.linalg.norm(a["q"] - lit["q"], axis=1) / np.linalg.norm(lit["q"], axis=1)
        assert e.max() <= 1e-10, e.max()


def test_compensated_low_part_enters_v_and_qdot_only():
    """otf_eval_lo: the low-order part of the iterate shifts v = x - qB and qdot = (x - qA)/eta and nothing else, i.e.
    g(x, lo) = g(x) + (M - eta D) lo to first order, with M - eta D read off two evaluations of H."""
    sc = _scene("chain32")
    sc.init()
    rng = np.random.default_rng(5)
    q = rng.uniform(-0.3, 0.3, sc.nr)
    qA = q - 1e-2 * rng.uniform(-0.1, 0.1, sc.nr)
    qB = qA + 1e-4 * rng.uniform(-0.1, 0.1, sc.nr)
    eta = 1e-2
    lo = 0.5 * np.spacing(q) * rng.uniform(-1, 1, sc.nr)
    g0 = orc.tensorfree_eval(sc.desc(), q, qA, qB, eta, want_H=False)
    g1 = orc.tensorfree_eval_lo(sc.desc(), q, lo, qA, qB, eta, want_H=False)
    # the same shift applied to qA, qB in exact arithmetic: v and qdot see q + lo, the geometry sees q.  Scale the test up so that
    # the shift is representable: lo -> 2^30 lo, compare with shifted qA, qB
    big = lo * 2.0 ** 30
    gA = orc.tensorfree_eval_lo(sc.desc(), q, big, qA, qB, eta, want_H=False)
    gB = orc.tensorfree_eval(sc.desc(), q, qA - big, qB - big, eta, want_H=False)
    assert np.linalg.norm(gA - gB) <= 1e-9 * np.linalg.norm(gA - g0)
    assert 0 < np.linalg.norm(g1 - g0) < 1e-7           # a sub-ulp shift of x moves g by ~|M| ulp(x): the lattice spacing of g
