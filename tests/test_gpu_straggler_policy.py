"""rmx_opts.ls_fail_limit, the opt-in straggler policy for batches (include/redmax_hip.h; NOT reference behaviour, off by default).

BASELINE.json configs[4] (32-link chain over frictional ground, BDF2, 1024 rollouts x 100 steps): ~5 % of the synthetic rollouts hit
a step where every line search of newton() (driverRedMaxBDF1.m:126-138) runs out its 20 trials without a decrease - a non-smooth
point of the residual - and the reference repeats that up to iterMax = 320 times: ~4000 extra evaluations for ONE step, and the
launch of the whole batch waits for those wavefronts (p50 3 ms, max 41 ms per 100 steps).  With ls_fail_limit = N the Newton loop
of such a step ends at its N-th failed line search.  Checked here:
  * off (0) is the default; rollouts the policy did not touch are BIT-identical with and without it;
  * every rollout it did touch carries RMX_ST_MAXITER | RMX_ST_LS_CUT, and the slowest wavefront gets >= 3x faster;
  * the same option restated in the oracle (orc_set_ls_fail_limit) gives the same trajectory (1e-6 |q|, the contact tolerance of
    tests/test_gpu_contact.py) on rollouts the policy cut.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ST_MAXITER, ST_LS_CUT = 2, 128


def _ground_batch(B, K, limit):
    import bench
    scene, h, integ, gen = bench.build_workload("ground", 32)
    assert integ == "bdf2"
    from redmax_amd import BatchSim
    sim = BatchSim(scene, batch=B)
    sim.opts.h, sim.opts.tol = h, 1e-9
    assert sim.opts.ls_fail_limit == 0                     # rmx_opts_default: the reference's loop
    sim.opts.ls_fail_limit = limit
    q0, qd0 = gen(0, B)
    sim.set_state(q0, qd0)
    out = sim.step_bdf2(K, stats=True)
    q, qd = sim.get_state()
    ticks = sim.step_ticks().astype(np.float64)
    sim.close()
    return scene, h, q0, qd0, q, qd, out, ticks


def test_ls_fail_limit_cuts_the_stragglers_and_nothing_else(oracle_lib):
    B, K = 1024, 100
    scene, h, q0, qd0, qa, qda, sa, ta = _ground_batch(B, K, 0)
    _, _, _, _, qb, qdb, sb, tb = _ground_batch(B, K, 2)
    assert (sa["status"] & ST_LS_CUT).max() == 0            # never set without the option
    cut = (sb["status"] & ST_LS_CUT) != 0
    assert cut.sum() >= 8, cut.sum()                        # the workload does have such rollouts (48 of 1024 when written)
    assert ((sb["status"][cut] & ST_MAXITER) != 0).all()
    # untouched rollouts: same decisions, same bits
    assert np.array_equal(qa[~cut], qb[~cut]) and np.array_equal(qda[~cut], qdb[~cut])
    assert np.array_equal(sa["newton_iters"][~cut], sb["newton_iters"][~cut])
    # every rollout that fails to converge without the option is one the option cuts (it fails through its line searches)
    slow = (sa["status"] & ST_MAXITER) != 0
    assert (cut[slow]).all()
    assert sb["newton_iters"][cut].sum() < 0.5 * sa["newton_iters"][cut].sum()
    print("ls_fail_limit=2: %d of %d rollouts cut; slowest wavefront %.3g -> %.3g ticks, total Newton iterations %d -> %d"
          % (cut.sum(), B, ta.max(), tb.max(), sa["newton_iters"].sum(), sb["newton_iters"].sum()))
    # (3 x before round 5; the default launch itself now parks its creeping rollouts for the cooperative line search, 42 -> 16 M ticks
    # measured on the slowest wavefront)
    assert tb.max() < ta.max() / 2.0
    # the oracle with the same option, on two of the rollouts that were cut
    oracle_lib.set_newton(tol=1e-9)
    oracle_lib.set_ls_fail_limit(2)
    try:
        for b in np.flatnonzero(cut)[:2]:
            o = oracle_lib.Oracle(scene.desc())
            o.set_state(q0[b], qd0[b])
            st = o.step_bdf2(h, K)
            qo, qdo = o.get_state()
            assert st.not_converged >= 1
            assert np.linalg.norm(qb[b] - qo) <= 1e-6 * np.linalg.norm(qo), (b, np.linalg.norm(qb[b] - qo) / np.linalg.norm(qo))
            assert np.linalg.norm(qdb[b] - qdo) <= 1e-4 * max(np.linalg.norm(qdo), 1.0)
    finally:
        oracle_lib.set_ls_fail_limit(0)


def test_negative_limit_is_refused():
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChain
    sc = sceneChain(4)
    sc.init()
    sim = BatchSim(sc, batch=2)
    sim.opts.ls_fail_limit = -1
    with pytest.raises(Exception):
        sim.step_bdf1(1)
    sim.close()
