"""BASELINE.json configs[2], [3] and [4] at their FULL per-GPU sizes through size-independent properties (the oracle cannot follow at
these sizes in test time), plus an oracle comparison on a small sample of the same batch.

  configs[2]  64-joint revolute/prismatic tree, BDF1, 512 rollouts per GPU (4096 / 8)
  configs[3]  adjoint BDF1 forward + backward, 16-DOF chain, 512 rollouts
  configs[4]  32-link chain over frictional ground, BDF2, 1024 rollouts
Properties: every rollout converges (or fails exactly where the reference algorithm does), finite results, shard invariance (a
rollout's result does not depend on its batch neighbours: bit-identical when a slice is recomputed alone), determinism, and for
the adjoint the finite-difference identity of the reference's own testGrad (driverRedMaxAdjointBDF1.m:46-61) on the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _tree_states(sc, B):
    qs, _ = sc.getQ()
    q0, qd0 = np.empty((B, sc.nr)), np.empty((B, sc.nr))
    for i in range(B):
        rng = np.random.default_rng(20240 + i)
        q0[i] = qs + rng.uniform(-0.05, 0.05, sc.nr)
        qd0[i] = rng.uniform(-0.1, 0.1, sc.nr)
    return q0, qd0


def test_config3_tree64_full_size(oracle_lib):
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneTree
    sc = sceneTree(64)
    sc.init()
    B, K = 512, 10
    q, qd = _tree_states(sc, B)
    sim = BatchSim(sc, batch=B)
    sim.set_state(q, qd)
    out = sim.step_bdf1(K, h=1e-2, stats=True, history=True)          # the reference's own Newton constants (tol = 1e-9)
    qa, qda = sim.get_state()
    assert (out["status"] & 15 == 0).all() and np.isfinite(qa).all() and np.isfinite(qda).all()
    T, V = sim.energy()
    assert np.allclose(T, out["T"][-1], rtol=1e-12, atol=1e-9) and np.allclose(V, out["V"][-1], rtol=1e-12, atol=1e-9)
    sub = BatchSim(sc, batch=64)
    sub.set_state(q[128:192], qd[128:192])
    sub.step_bdf1(K, h=1e-2)
    qs, qds = sub.get_state()
    assert np.array_equal(qs, qa[128:192]) and np.array_equal(qds, qda[128:192])
    sim.set_state(q, qd)
    sim.step_bdf1(K, h=1e-2)
    qb, _ = sim.get_state()
    assert np.array_equal(qa, qb)
    for b in (0, 77, 300, 511):                                          # a sample of the same batch against the oracle
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(q[b], qd[b])
        st = o.step_bdf1(1e-2, K)
        qo, qdo = o.get_state()
        assert _rel(qa[b], qo) <= 1e-8 and _rel(qda[b], qdo) <= 1e-6, (b, _rel(qa[b], qo))
        assert out["newton_iters"][b] == st.newton_iters, (b, out["newton_iters"][b], st.newton_iters)


def test_config4_adjoint_full_size(oracle_lib):
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneAdjointChain
    sc = sceneAdjointChain(16)
    sc.init()
    B, K = 512, 20          # the longest round horizon on which the reference's line-search-free Newton converges on every step
    rng = np.random.default_rng(20240)
    p = 1e-1 * rng.standard_normal((B, sc.nr))
    p[1] = p[0]             # two rollouts with the same parameters
    task = dict(sc.task, t=K * sc.h)
    q0, qd0 = sc.getQ()
    sim = BatchSim(sc, batch=B)
    sim.set_state(q0[None, :], qd0[None, :])
    P, dPdp, info = sim.adjoint_bdf1(K, sc.h, task, p, stats=True)
    assert (info["status"] == 0).all() and np.isfinite(P).all() and np.isfinite(dPdp).all()
    assert P[0] == P[1] and np.array_equal(dPdp[0], dPdp[1])             # no cross-talk between rollouts
    sub = BatchSim(sc, batch=32)
    sub.set_state(q0[None, :], qd0[None, :])
    Ps, dPs, _ = sub.adjoint_bdf1(K, sc.h, task, p[200:232])
    assert np.array_equal(Ps, P[200:232]) and np.array_equal(dPs, dPdp[200:232])      # shard invariance
    for b in (0, 123, 511):                                              # sample vs the oracle's literal restatement
        o = oracle_lib.Oracle(sc.desc())
        Po, dPo, st = o.adjoint_bdf1(sc.h, K, task, p[b])
        assert abs(P[b] - Po) <= 1e-9 * abs(Po) and _rel(dPdp[b], dPo) <= 1e-7, (b, P[b], Po)
        assert info["newton_iters"][b] == st.newton_iters
    # testGrad (driverRedMaxAdjointBDF1.m:46-61) on the device: central differences of P along 8 random directions, one per
    # pair of rollouts of a single launch, against dPdp . direction
    nd, eps = 8, 1e-5
    d = rng.standard_normal((nd, sc.nr))
    pp = np.repeat(p[:1], 2 * nd, axis=0)
    pp[0::2] += eps * d
    pp[1::2] -= eps * d
    fd = BatchSim(sc, batch=2 * nd)
    fd.set_state(q0[None, :], qd0[None, :])
    Pf, _, _ = fd.adjoint_bdf1(K, sc.h, task, pp)
    num = (Pf[0::2] - Pf[1::2]) / (2 * eps)
    ana = d @ dPdp[0]
    assert np.allclose(num, ana, rtol=2e-5, atol=1e-6 * np.abs(ana).max()), (num, ana)


def test_config5_chain_ground_full_size(oracle_lib):
    """BASELINE.json configs[4] at FULL size: 1024 rollouts of the 32-link chain over frictional ground, BDF2, the bench's 100 steps.
    Properties of the whole batch (nothing diverges, shard invariance bit for bit) and 32 rollouts spread over the batch against the
    literal oracle over ALL 100 steps - through touch-down and stick / slip switching, including the rollouts that hold a creeping step
    (the reference's Newton running out its 320 iterations: "did not converge" on both sides): final q to 1e-8, Newton counts within 2 % (a line-search decision at a stick / slip kink may fall the other way),
    the same rollouts flagged."""
    from concurrent.futures import ThreadPoolExecutor
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChainGround, syntheticStates
    sc = sceneChainGround(32)
    sc.init()
    B, K, NS = 1024, 100, 32
    q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1)
    q[0], qd[0] = sc.getQ()
    sim = BatchSim(sc, batch=B)
    sim.set_state(q, qd)
    out = sim.step_bdf2(K, h=sc.h, stats=True)
    qa, qda = sim.get_state()
    assert np.isfinite(qa).all() and np.isfinite(qda).all() and not (out["status"] & 5).any()      # nothing diverged, no NaN
    assert not (out["status"] & 512).any()                                                         # no cooperative group gave up
    assert not (out["status"] & (64 | 256)).any()                                                  # internal hand-over bits stay inside the launch
    sub = BatchSim(sc, batch=64)
    sub.set_state(q[512:576], qd[512:576])
    sub.step_bdf2(K, h=sc.h)
    qs, qds = sub.get_state()
    assert np.array_equal(qs, qa[512:576]) and np.array_equal(qds, qda[512:576])
    # 29 rollouts spread over the batch + three that hold a creeping step (found by tools runs of this very comparison)
    idx = sorted(set([int(i) for i in np.linspace(0, B - 1, NS - 3)] + [205, 682, 886]))

    def run(b):                                  # (ctypes releases the GIL: one oracle per thread)
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(q[b], qd[b])
        st = o.step_bdf2(sc.h, K)
        return o.get_state()[0], st
    with ThreadPoolExecutor(16) as ex:
        res = list(ex.map(run, idx))
    creeping = 0
    for b, (qo, st) in zip(idx, res):
        assert not st.diverged, b
        assert bool(st.not_converged) == bool(out["status"][b] & 2), (b, st.not_converged, out["status"][b])
        creeping += 1 if st.not_converged else 0
        assert _rel(qa[b], qo) <= 1e-8, (b, _rel(qa[b], qo))
        assert abs(int(out["newton_iters"][b]) - st.newton_iters) <= max(3, st.newton_iters // 50), (b, out["newton_iters"][b], st.newton_iters)
    assert creeping >= 1, "the sample holds no rollout with a creeping step: widen it"


def test_trees_of_33_to_64_nodes_match_oracle(oracle_lib):
    """Trees of 33..64 nodes (one row of H per lane: matrix-core Hessian in two column halves, block-column DPP solve with H resident
    in LDS) against the oracle with identical Newton iteration counts, and lu_mode = 1 (always partial pivoting, the row-per-lane
    solve) against the default to roundoff."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChain, sceneTree
    for sc, K in ((sceneTree(64), 10), (sceneChain(48), 6), (sceneTree(40), 8)):
        sc.init()
        B = 6
        q, qd = _tree_states(sc, B)
        res = {}
        for mode in (0, 1):
            sim = BatchSim(sc, batch=B)
            sim.opts.lu_mode = mode
            sim.set_state(q, qd)
            out = sim.step_bdf1(K, h=1e-2, stats=True, history="full")
            res[mode] = (sim.get_state(), out)
            sim.close()
        (q0, qd0), o0 = res[0]
        (q1, qd1), o1 = res[1]
        assert (o0["status"] & 15 == 0).all() and (o1["status"] & 15 == 0).all()
        assert np.allclose(o0["T"], o1["T"], rtol=1e-9, atol=1e-9) and np.allclose(o0["q"], o1["q"], rtol=1e-9, atol=1e-11)
        for b in range(B):
            assert _rel(q1[b], q0[b]) <= 1e-10, (sc.name, b, _rel(q1[b], q0[b]))
        for b in (0, B - 1):
            o = oracle_lib.Oracle(sc.desc())
            o.set_state(q[b], qd[b])
            st = o.step_bdf1(1e-2, K)
            qo, _ = o.get_state()
            assert _rel(q0[b], qo) <= 1e-8, (sc.name, b, _rel(q0[b], qo))
            # long chains in cgs units: |M| ulp(q) reaches tol, the oracle's last iterations of a step wander over the lattice of
            # doubles until |g| < tol is met; the compensated iterate converges without them
            ng, no = int(o0["newton_iters"][b]), st.newton_iters
            assert (no - 3 * K <= ng <= no) if "chain" in sc.name else ng == no, (sc.name, b, ng, no)


def test_tree64_global_constants_kernels(oracle_lib, monkeypatch):
    """Batches of more than two rollouts per CU run the 64-lane step kernels that read the per-node constants from global memory
    instead of LDS (four wavefronts per CU instead of two; rmx_kernels.hip RMX_PART 3).  Same arithmetic on the same values: the
    results must equal the LDS-constants kernels' (RMX_GCONST_MIN moves the threshold, read at model creation), BDF1 and BDF2, and
    the oracle's."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneTree
    sc = sceneTree(64)
    sc.init()
    B, K = 6, 8
    q, qd = _tree_states(sc, B)
    res = {}
    for integ in ("bdf1", "bdf2"):
        for thr in ("100000", "1"):          # never / always
            monkeypatch.setenv("RMX_GCONST_MIN", thr)
            sim = BatchSim(sc, batch=B)
            sim.set_state(q, qd)
            out = (sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2)(K, h=1e-2, stats=True)
            res[(integ, thr)] = (sim.get_state(), out)
            sim.close()
        (qa, qda), oa = res[(integ, "100000")]
        (qb, qdb), ob = res[(integ, "1")]
        assert (ob["status"] & 15 == 0).all() and np.array_equal(oa["newton_iters"], ob["newton_iters"])
        assert _rel(qb, qa) <= 1e-13 and _rel(qdb, qda) <= 1e-11, (integ, _rel(qb, qa))
    monkeypatch.delenv("RMX_GCONST_MIN")
    (qb, _), ob = res[("bdf1", "1")]
    for b in (0, B - 1):
        o = oracle_lib.Oracle(sc.desc())
        o.set_state(q[b], qd[b])
        st = o.step_bdf1(1e-2, K)
        qo, _ = o.get_state()
        assert _rel(qb[b], qo) <= 1e-8 and int(ob["newton_iters"][b]) == st.newton_iters


@pytest.mark.parametrize("nodes", [64, 41, -64])
def test_tree64_two_wave_kernels(monkeypatch, nodes):
    """Batches of at most one rollout per two SIMDs give a 33..64-node tree TWO wavefronts (rmx_kernels.hip RMX_PART 5): the second
    one takes the odd columns of the Hessian tiles and its share of the later column blocks in every phase of the block-column
    elimination.  Every matrix entry sees the same operations on the same values in the same order as in the one-wave kernels:
    bit-identical states, iteration counts and status words (RMX_W2_MAX moves the threshold, read at model creation) - BDF2, pivoting
    throughout (lu_mode 1: wave 0 alone, the helper released at the end), serial chains (-64: the 64-link chain, FULLCHAIN front),
    history on, and states wild enough for line searches and failed steps.  41 nodes: partly filled trees (n at run time).
    The guarded BDF1 steps of a TREE run the loop that evaluates the next step's first point ahead (w2_steps_bdf1): (i) with the
    run-ahead switched off (RMX_W2_RUNAHEAD=0) the results are the same bits - what is run ahead is exactly what would have been run;
    (ii) against the one-wave kernels that loop is a different compilation of the same operations: the compiler contracts a few of
    them differently, the states agree to rounding (1e-11 after 14 steps) with equal Newton counts."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneChain, sceneTree
    sc = sceneTree(nodes) if nodes > 0 else sceneChain(-nodes)
    sc.init()
    B, K = 12, 12
    q, qd = _tree_states(sc, B)
    rng = np.random.default_rng(7)
    wild_q = q + rng.uniform(-1.5, 1.5, q.shape)
    wild_qd = rng.uniform(-40.0, 40.0, qd.shape)

    def run(integ, lu_mode, qq, qqd, tol, w2, ahead="1"):
        monkeypatch.setenv("RMX_W2_MAX", w2)
        monkeypatch.setenv("RMX_W2_RUNAHEAD", ahead)
        sim = BatchSim(sc, batch=B)
        sim.opts.lu_mode = lu_mode
        sim.opts.tol = tol
        sim.set_state(qq, qqd)
        step = sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2
        step(2, h=1e-2)                                          # (BDF2: the second call runs on history)
        out = step(K, h=1e-2, stats=True, history=True)
        res = (sim.get_state(), out)
        sim.close()
        return res

    def same_bits(ra, rb, what):
        ((qa, qda), oa), ((qb, qdb), ob) = ra, rb
        assert np.array_equal(qa, qb, equal_nan=True) and np.array_equal(qda, qdb, equal_nan=True), what
        for k in ("newton_iters", "ls_halvings", "status", "T", "V"):
            assert np.array_equal(oa[k], ob[k], equal_nan=True), (what, k)

    for integ in ("bdf1", "bdf2"):
        for lu_mode, (qq, qqd), tol in ((0, (q, qd), 1e-9), (1, (q, qd), 1e-9), (0, (wild_q, wild_qd), 1e-6)):
            one = run(integ, lu_mode, qq, qqd, tol, "0")
            two = run(integ, lu_mode, qq, qqd, tol, "100000")
            runs_ahead = integ == "bdf1" and lu_mode == 0 and nodes > 0
            if not runs_ahead:
                same_bits(one, two, (integ, lu_mode, tol))
            else:
                same_bits(two, run(integ, lu_mode, qq, qqd, tol, "100000", ahead="0"), (integ, lu_mode, tol, "run-ahead on / off"))
                if tol == 1e-9:
                    ((qa, qda), oa), ((qb, qdb), ob) = one, two
                    assert _rel(qb, qa) <= 1e-11 and _rel(qdb, qda) <= 1e-9, (_rel(qb, qa), _rel(qdb, qda))
                    assert np.array_equal(oa["newton_iters"], ob["newton_iters"]) and np.array_equal(oa["status"], ob["status"])
                    assert np.allclose(oa["T"], ob["T"], rtol=1e-10, atol=0) and np.allclose(oa["V"], ob["V"], rtol=1e-10, atol=0)
            oa = one[1]
            if tol == 1e-9:
                assert (oa["status"] & 15 == 0).all()
            else:
                assert oa["ls_halvings"].sum() > 0 or (oa["status"] & 15).any(), "no line search ran: the wild states are too tame"
    monkeypatch.delenv("RMX_W2_MAX")
    monkeypatch.delenv("RMX_W2_RUNAHEAD")


def test_chain32_two_wave_kernel(monkeypatch):
    """Shards of 128 .. 512 rollouts of the full 32-link chain (the 1024-rollout batch of BASELINE.json configs[1] on two or more GPUs) run
    BDF1 with a second wavefront per rollout that evaluates the point which may end a step's solve while the first one evaluates the
    next step's first point (rmx_kernels.hip RMX_PART 6, w2_steps_bdf1).  The helper runs the full front, every decision is
    newton_rot's: states, Newton counts, status words and histories equal the one-wave headline kernel's bit for bit, with the run-ahead
    on and off, on the bench states and on states wild enough for line searches."""
    from redmax_amd import BatchSim, sceneChain, syntheticStates
    sc = sceneChain(32)
    sc.init()
    B, K = 128, 12
    q, qd = syntheticStates(sc.nr, B)
    wq, wqd = syntheticStates(sc.nr, B, sq=0.6, sv=4.0)

    def run(qq, qqd, tol, w2, ahead="1"):
        monkeypatch.setenv("RMX_W2_MAX", w2)
        monkeypatch.setenv("RMX_W2_RUNAHEAD", ahead)
        sim = BatchSim(sc, batch=B)
        sim.opts.tol = tol
        sim.set_state(qq, qqd)
        sim.step_bdf1(2, h=1e-2)
        out = sim.step_bdf1(K, h=1e-2, stats=True, history=True)
        res = (sim.get_state(), out)
        sim.close()
        return res

    for qq, qqd, tol in ((q, qd, 1e-9), (wq, wqd, 1e-6)):
        one = run(qq, qqd, tol, "0")
        for ahead in ("1", "0"):
            ((qa, qda), oa), ((qb, qdb), ob) = one, run(qq, qqd, tol, "100000", ahead)
            assert np.array_equal(qa, qb, equal_nan=True) and np.array_equal(qda, qdb, equal_nan=True), (tol, ahead)
            for k in ("newton_iters", "ls_halvings", "status", "T", "V"):
                assert np.array_equal(oa[k], ob[k], equal_nan=True), (tol, ahead, k)
        if tol == 1e-6:
            assert one[1]["ls_halvings"].sum() > 0 or (one[1]["status"] & 15).any(), "no line search ran: the wild states are too tame"
    monkeypatch.delenv("RMX_W2_MAX")
    monkeypatch.delenv("RMX_W2_RUNAHEAD")


def test_chain32_pair_kernel(monkeypatch):
    """The headline kernel (rmx_pair32.h, RMX_PART 7): every evaluation of the front carries the next step's first point beside the
    line-search trial, in the half-wave a 32-node tree leaves idle; a trial that ends a step's solve hands the next solve its first
    evaluation.  Every decision and every operation per point is the one-point kernel's (RMX_PAIRC=0): states, Newton counts, halvings,
    status words and per-step energies must be equal bit for bit - on the bench states at the reference's tol, on states wild enough for
    line searches and diverging rollouts, with the pivoting solve (lu_mode 1), on plain doubles (compensated 0),
    and for a single step and a single rollout."""
    from redmax_amd import BatchSim, sceneChain, syntheticStates
    sc = sceneChain(32)
    sc.init()
    monkeypatch.setenv("RMX_W2_MAX", "0")

    def run(pair, B, K, tol=1e-9, wild=False, lu_mode=0, comp=1):
        monkeypatch.setenv("RMX_PAIRC", pair)
        q, qd = syntheticStates(sc.nr, B, sq=0.6, sv=4.0) if wild else syntheticStates(sc.nr, B)
        sim = BatchSim(sc, batch=B)
        sim.opts.tol = tol
        sim.opts.lu_mode = lu_mode
        sim.opts.compensated = comp
        sim.set_state(q, qd)
        sim.step_bdf1(2, h=1e-2)
        out = sim.step_bdf1(K, h=1e-2, stats=True, history="full")      # T, V, q, qdot of every step (Scene.saveHistory)
        res = (sim.get_state(), out)
        sim.close()
        return res

    cases = (dict(B=128, K=12), dict(B=128, K=12, tol=1e-6, wild=True), dict(B=64, K=6, lu_mode=1), dict(B=64, K=8, comp=0),
             dict(B=64, K=6, tol=1e-6, wild=True, lu_mode=1), dict(B=1, K=1), dict(B=3, K=2))
    for kw in cases:
        ((qa, qda), oa), ((qb, qdb), ob) = run("0", **kw), run("1", **kw)
        assert np.array_equal(qa, qb, equal_nan=True) and np.array_equal(qda, qdb, equal_nan=True), kw
        for k in ("newton_iters", "ls_halvings", "status", "T", "V", "q", "qdot"):
            assert np.array_equal(oa[k], ob[k], equal_nan=True), (kw, k)
        if kw.get("wild") and not kw.get("lu_mode"):
            assert oa["ls_halvings"].sum() > 0 or (oa["status"] & 15).any(), "no line search ran: the wild states are too tame"
    monkeypatch.delenv("RMX_PAIRC")
    monkeypatch.delenv("RMX_W2_MAX")


def test_max_valid_amplitude_sample(oracle_lib):
    """The headline workload at the LARGEST initial-state amplitude the reference algorithm survives (q, qdot ~ U(-0.1856, 0.1856):
    found by tools/max_valid_amplitude.py on the literal oracle, profiles/r04_max_valid_amplitude.json): 1024 rollouts x 100 BDF1
    steps on the GPU - no rollout diverges -, and a 64-rollout sample of the same batch against the literal oracle over the whole
    rollout: final q to 1e-8 (SURVEY.md 8(d)'s rollout tolerance)."""
    import os
    from redmax_amd import BatchSim, sceneChain, syntheticStates
    from redmax_amd.scenes import MAX_VALID_INIT_AMPLITUDE as A
    sc = sceneChain(32)
    sc.init()
    B, K, NS = 1024, 100, 64
    q, qd = syntheticStates(sc.nr, B, sq=A, sv=A)
    sim = BatchSim(sc, batch=B)
    sim.set_state(q, qd)
    out = sim.step_bdf1(K, h=sc.h, stats=True)
    qg, qdg = sim.get_state()
    sim.close()
    assert np.isfinite(qg).all() and (out["status"] & 1 == 0).all()            # nothing diverged
    assert (out["status"] & 15 != 0).mean() <= 0.01                              # the compensated iterate converges (nearly) everywhere
    idx = np.linspace(0, B - 1, NS).astype(int)
    qo, qdo = np.ascontiguousarray(q[idx]), np.ascontiguousarray(qd[idx])
    oracle_lib.set_newton()
    cnt = oracle_lib.batch_step_bdf1(sc.desc(), qo, qdo, sc.h, K, nthreads=os.cpu_count(), counters=True)
    assert int(cnt["diverged"].sum()) == 0 and float(cnt["worst_exit_g"].max()) < 1e-6
    eq = np.linalg.norm(qg[idx] - qo, axis=1) / np.linalg.norm(qo, axis=1)
    assert eq.max() <= 1e-8, eq.max()
