"""Reduced slices of the two long-running checks that used to be scripts only (tests/soak_parity.py, tests/fuzz_more.py):

* soak: 256 of the 1024 benchmark rollouts (every 4th global index, incl. the deterministic rollout 0) x 25 BDF1 steps at the
  benchmark's Newton tolerance against the CPU oracle (OpenMP over the host cores), with the PER-ROLLOUT NEWTON ITERATION COUNTS
  compared per (rollout, step) - SURVEY.md §8(d) expects identical counts on >= 99 % of trajectory-steps.  tol = 1e-8 is above the fp64 noise floor of
  |g| on this chain, so the counts are reproducible, which they are not at the reference's 1e-9.
* fuzz: 36 more random trees (30 small, 6 with 33..62 nodes) through test_gpu_fuzz's parity check; besides parity this keeps the
  DPP-fused elimination (fmsub_rowbcast: a DPP read right after a VALU write of the same register sees the old value) exercised
  on many matrix sizes and pivot patterns.
The full-size versions remain runnable: python tests/soak_parity.py, python tests/fuzz_more.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_chain32_soak_slice_with_newton_counts(oracle_lib):
    from redmax_amd import BatchSim, sceneChain, syntheticStates
    sc = sceneChain(32)
    sc.init()
    B, K, h, tol = 256, 25, 1e-2, 1e-8
    q = np.empty((B, 32))
    qd = np.empty((B, 32))
    for i in range(B):
        q[i], qd[i] = (a[0] for a in syntheticStates(32, 1, first=4 * i))
    sim = BatchSim(sc, batch=B)
    sim.opts.tol = tol
    sim.set_state(q, qd)
    oracle_lib.set_newton(tol=tol)
    qc, qdc = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
    it_g = np.zeros((K, B), dtype=np.int64)
    it_o = np.zeros((K, B), dtype=np.int64)
    ls_g = ls_o = 0
    status = np.zeros(B, dtype=np.int64)
    bad = 0
    for s in range(K):          # one step per call on both sides, so the counts can be compared per (rollout, step)
        out = sim.step_bdf1(1, h=h, stats=True)
        cnt = oracle_lib.batch_step_bdf1(sc.desc(), qc, qdc, h, 1, nthreads=os.cpu_count(), counters=True)
        it_g[s], it_o[s] = out["newton_iters"], cnt["newton_iters"]
        ls_g += int(out["ls_halvings"].sum())
        ls_o += int(cnt["ls_halvings"].sum())
        status |= out["status"]
        bad += int(cnt["bad"].sum())
    oracle_lib.set_newton()
    qg, qdg = sim.get_state()
    sim.close()
    eq = np.linalg.norm(qg - qc, axis=1) / np.linalg.norm(qc, axis=1)
    ed = np.linalg.norm(qdg - qdc, axis=1) / np.maximum(np.linalg.norm(qdc, axis=1), 1e-30)
    assert (status & 15 == 0).all() and bad == 0
    assert eq.max() <= 1e-10 and ed.max() <= 1e-8, (eq.max(), ed.max())
    same = it_g == it_o
    print("newton counts: %d/%d trajectory-steps identical (%d/%d rollouts on every step); gpu %d vs oracle %d iterations; "
          "halvings gpu %d oracle %d" % (same.sum(), same.size, same.all(axis=0).sum(), B, it_g.sum(), it_o.sum(), ls_g, ls_o))
    assert same.mean() >= 0.99, (same.mean(), np.argwhere(~same)[:20].tolist())
    assert abs(int(it_g.sum()) - int(it_o.sum())) <= 0.002 * it_o.sum()


# seeds whose oracle rollouts take 38 - 80 s each on the GPU box (GPUTEST r04 / r5a durations): run with RMX_FULL_TESTS=1 only
# (round-4 review: the GPU suite under 600 s of the driver's 1200 s step limit)
_SLOW_SEEDS = (307, 319, 323, 605)


@pytest.mark.parametrize("seed", [s for s in list(range(300, 330)) + list(range(600, 606))
                                  if s not in _SLOW_SEEDS or os.environ.get("RMX_FULL_TESTS") == "1"])
def test_random_tree_matches_oracle_extended(oracle_lib, seed, monkeypatch):
    import test_gpu_fuzz as tf
    if seed >= 600:       # the suite's convention for the 33..62-node trees
        orig = tf._random_scene
        monkeypatch.setattr(tf, "_random_scene", lambda s, contact=False, big=False: orig(s, contact=contact, big=True))
    tf.test_random_tree_matches_oracle(oracle_lib, seed)      # a random contact scene of more than 64 nodes runs on the large-tree kernels
