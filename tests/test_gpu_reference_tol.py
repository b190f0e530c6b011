"""BASELINE.json configs[1] at the reference's own Newton constant, tol = 1e-9 (driverRedMaxBDF1.m:95), for the reference's rollout
length (100 steps = tEnd 1, Scene.m:117): does the GPU fail where - and as often as - the reference algorithm does?

What decides it: g depends on x at the resolution of one ulp through M (x - qB), and |M| ulp(q) ~ 1e-9 on
this 320 cm cgs chain, so on the lattice of doubles |g| < 1e-9 holds only at lucky points.  The literal oracle (= the reference's
arithmetic) finds them because its own evaluation noise dithers the Newton update; the world-frame evaluation has a smoother error
and, on plain doubles, sticks on >= 10 % of the trajectory-steps.  The kernels therefore carry the iterate as x + xlo
(rmx_opts.compensated = 1, the default).  Three things are pinned here, on 64 of the benchmark's rollouts (every 16th global index):

  * default mode: no failed step at all, never more failures or iterations than the literal oracle, and the same counts per
    (rollout, step) as the tensor-free CPU code in the same mode on >= 99 % of trajectory-steps;
  * rmx_opts.compensated = 0 (plain doubles): the GPU fails as often as the tensor-free CPU code in that mode (same algorithm,
    same lattice: failure fraction, iterations and halvings within a stated band) - and far more often than the literal oracle,
    which is the finding that motivates the default;
  * the lattice itself: |g| at the lattice neighbours of a converged state is the same on both sides (GPU evaluation vs literal
    oracle differ by < 5e-10, medians of |g| within 20 %): neither side has the higher roundoff floor.
Final states of all five runs agree to <= 1e-10 relative (SURVEY 8(d) allows 1e-8)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_chain32_reference_tol_failure_statistics(oracle_lib):
    from reference_tol import run_stats
    B, K = 64, 100
    r = run_stats(B, K, tol=1e-9, stride=16)
    gpu, gpl, tf, tfp, lit = (r[k] for k in ("gpu", "gpu_plain", "tensor_free", "tensor_free_plain", "literal"))
    for k, v in r.items():
        print("%-18s it/step %.3f halvings/step %.3f failed steps %.4f (first half %.4f, second half %.4f) rollouts with a failed step %d" % (
            k, v["iters_per_step"], v["halvings_per_step"], v["bad_frac"], v["bad_frac_first_half"], v["bad_frac_second_half"], v["rollouts_with_a_bad_step"]))
    # ---- the default (compensated iterate): converges on every step, at the cost of <= the reference algorithm's iterations
    assert gpu["bad_frac"] == 0.0 and tf["bad_frac"] == 0.0
    assert gpu["bad_frac"] <= lit["bad_frac"] <= 0.02
    assert 3.5 <= gpu["iters_per_step"] <= lit["iters_per_step"] + 0.05
    assert gpu["halvings_per_step"] <= 0.01
    same = gpu["iters"] == tf["iters"]
    assert same.mean() >= 0.99, same.mean()
    assert abs(gpu["iters"].sum() - tf["iters"].sum()) <= 0.002 * tf["iters"].sum()
    # ---- plain doubles: the GPU fails where its CPU twin fails, and as often
    assert 0.05 <= gpl["bad_frac"] <= 0.25 and 0.05 <= tfp["bad_frac"] <= 0.25
    assert abs(gpl["bad_frac"] - tfp["bad_frac"]) <= 0.05
    assert gpl["bad_frac_first_half"] <= 0.01 and tfp["bad_frac_first_half"] <= 0.01       # the lattice binds once the chain has swung out
    assert abs(gpl["iters_per_step"] - tfp["iters_per_step"]) <= 0.15 * tfp["iters_per_step"]
    assert abs(gpl["halvings_per_step"] - tfp["halvings_per_step"]) <= 0.25 * tfp["halvings_per_step"]
    assert gpl["bad_frac"] >= 5.0 * max(lit["bad_frac"], 0.005)       # ... which is far more often than the reference algorithm
    # ---- final states
    for a in (gpu, gpl, tf, tfp):
        e = np.linalg.norm(a["q"] - lit["q"], axis=1) / np.linalg.norm(lit["q"], axis=1)
        ed = np.linalg.norm(a["qd"] - lit["qd"], axis=1) / np.linalg.norm(lit["qd"], axis=1)
        assert e.max() <= 1e-10 and ed.max() <= 1e-8, (e.max(), ed.max())


def test_chain32_roundoff_floor_is_the_lattice_on_both_sides(oracle_lib):
    """|g| over the lattice neighbours of a converged state of the late rollout (step 80), evaluated by the GPU (rmx_eval) and by the
    literal oracle at the same points: same values (the evaluations differ by far less than tol), same distribution - the 'floor' is
    H rho, rho the rounding of x, not evaluation noise of either side."""
    from redmax_amd import BatchSim, sceneChain, syntheticStates
    sc = sceneChain(32)
    sc.init()
    h, B, NP = 1e-2, 8, 24
    q, qd = syntheticStates(32, B, first=1)
    oracle_lib.set_newton(tol=1e-8)
    qc, qdc = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
    oracle_lib.batch_step_bdf1(sc.desc(), qc, qdc, h, 80, nthreads=os.cpu_count())
    q0, qd0 = qc.copy(), qdc.copy()
    oracle_lib.batch_step_bdf1(sc.desc(), qc, qdc, h, 1, nthreads=os.cpu_count())        # qc: a converged x of step 81
    oracle_lib.set_newton()
    rng = np.random.default_rng(11)
    sim = BatchSim(sc, batch=B)
    ng, nl, dif = [], [], []
    o = [oracle_lib.Oracle(sc.desc()) for _ in range(B)]
    for t in range(NP):
        x = qc + rng.integers(-2, 3, qc.shape) * np.spacing(qc)
        g = sim.eval_bdf1(x, q0, qd0, h, want_H=False)
        for b in range(B):
            gl = o[b].eval_bdf1(x[b], q0[b], qd0[b], h, want_H=False)
            ng.append(np.linalg.norm(g[b]))
            nl.append(np.linalg.norm(gl))
            dif.append(np.linalg.norm(g[b] - gl))
    sim.close()
    ng, nl, dif = np.array(ng), np.array(nl), np.array(dif)
    print("|g| over %d lattice neighbours: GPU median %.3e, literal median %.3e, fraction below 1e-9: %.3f / %.3f, max |g_gpu - g_lit| %.2e" % (
        ng.size, np.median(ng), np.median(nl), (ng < 1e-9).mean(), (nl < 1e-9).mean(), dif.max()))
    assert dif.max() <= 5e-10
    assert abs(np.median(ng) - np.median(nl)) <= 0.2 * np.median(nl)
    assert np.median(nl) >= 1e-9          # the lattice spacing alone puts a typical neighbour above the reference's tol
