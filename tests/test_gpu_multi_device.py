"""Multi-device stepping through the C ABI (ABI 107: rmx_step_*_async, rmx_history_read, rmx_group_*; VERDICT round 3 row *).

BASELINE.json's north_star shards the batch axis across GPUs with MATLAB (one host thread) as the host.  The GPU box has ONE
device, so the shards here share device 0, each on its own stream: that exercises every line of the N-device code path (a device
may be listed more than once) and lets HIP events - which share a clock on one device - prove that the launches overlap.  The
results must not depend on the sharding: trajectories are independent, so every array equals the single-batch result bit for bit.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _single(scene, q, qd, K, integ, record_full=True):
    from redmax_amd import BatchSim
    sim = BatchSim(scene, batch=q.shape[0])
    sim.set_state(q, qd)
    out = (sim.step_bdf1 if integ == 1 else sim.step_bdf2)(K, h=scene.h, stats=True, history="full" if record_full else True)
    out["qf"], out["qdf"] = sim.get_state()
    out["E"] = sim.energy()
    sim.close()
    return out


def test_group_on_two_streams_equals_one_batch_and_overlaps():
    from redmax_amd import GroupSim, sceneChain, syntheticStates
    sc = sceneChain(32)
    sc.init()
    B, K = 2048, 25
    q, qd = syntheticStates(sc.nr, B)
    ref = _single(sc, q, qd, K, 1)
    g = GroupSim(sc, B, devices=(0, 0))
    assert g.nshards == 2 and [s[2] for s in g.shards] == [1024, 1024] and [s[1] for s in g.shards] == [0, 1024]
    g.set_state(q, qd)
    out = g.step(K, integrator=1, h=sc.h, record=3)
    qf, qdf = g.get_state()
    T, V = g.energy()
    for k in ("T", "V", "q", "qdot", "newton_iters", "ls_halvings", "status"):
        assert np.array_equal(out[k], ref[k]), k
    assert np.array_equal(qf, ref["qf"]) and np.array_equal(qdf, ref["qdf"])
    assert np.array_equal(T, ref["E"][0]) and np.array_equal(V, ref["E"][1])
    # the two launches ran at the same time: the second started before the first ended, and the wall clock of the whole step is
    # shorter than the two kernels back to back
    assert out["start_ms"][1] < out["end_ms"][0], out
    assert out["wall_ms"] < out["kernel_ms"].sum(), out
    # ... and on a SECOND recorded step as well: the per-step record lives in buffers that stay on the batch (no hipFree, which
    # synchronises the device, between the two shards' launches; round-4 advice)
    out2 = g.step(K, integrator=1, h=sc.h, record=3)
    assert out2["start_ms"][1] < out2["end_ms"][0], out2
    assert out2["wall_ms"] < out2["kernel_ms"].sum(), out2
    g.close()


def test_async_bookkeeping_fails_loudly():
    """history_read / GroupSim.sync with nothing outstanding, and a synchronous step over a recorded launch still in flight."""
    from redmax_amd import BatchSim, GroupSim, sceneChain, syntheticStates
    from redmax_amd._abi import RedMaxHipError
    sc = sceneChain(8)
    sc.init()
    q, qd = syntheticStates(sc.nr, 4)
    sim = BatchSim(sc, batch=4)
    sim.set_state(q, qd)
    with pytest.raises(RedMaxHipError, match="no step_history_async"):
        sim.history_read()
    sim.step_history_async(3, integrator=1, h=sc.h)
    with pytest.raises(RedMaxHipError, match="pending"):
        sim.step_bdf1(1, h=sc.h, history=True)               # would replace the record of the launch in flight
    sim.step_history_async(3, integrator=1, h=sc.h)          # (the refused call has dropped the python-side shape: record again)
    sim.sync()
    rec = sim.history_read()
    assert rec["q"].shape == (3, 4, sc.nr)
    sim.step_bdf1_async(2, h=sc.h)                           # an unrecorded launch replaces the record ...
    sim.sync()
    with pytest.raises(RedMaxHipError, match="no step_history_async"):
        sim.history_read()                                   # ... so there is nothing to read (no stale shapes, no empty arrays)
    # half a history pair is refused BEFORE the state advances
    q1, _ = sim.get_state()
    import ctypes as C
    from redmax_amd import _abi
    T = np.empty((1, 4))
    rc = sim._L.rmx_step_bdf1(sim._batch, C.byref(sim.opts), 1, None, _abi.dptr(T), None)
    assert rc == -1
    assert np.array_equal(sim.get_state()[0], q1)
    sim.close()
    g = GroupSim(sc, 4, devices=(0, 0))
    with pytest.raises(RedMaxHipError, match="no step_async"):
        g.sync()
    g.close()


def test_uneven_shards_bdf2_with_euler_charts():
    """5 trajectories over 3 shards (2 + 2 + 1) of scene 7 (JointSpherical, BDF2: the chart switches inside the rollout)."""
    from redmax_amd import GroupSim
    from redmax_amd.scenes import scenesRedMax
    sc = scenesRedMax(7)
    sc.init()
    q0, qd0 = sc.getQ()
    B, K = 5, sc.nsteps
    rng = np.random.default_rng(3)
    q = q0[None, :] + 0.05 * rng.standard_normal((B, sc.nr))
    qd = qd0[None, :] + 0.2 * rng.standard_normal((B, sc.nr))
    q[0], qd[0] = q0, qd0
    ref = _single(sc, q, qd, K, 2)
    g = GroupSim(sc, B, devices=(0, 0, 0))
    assert [s[2] for s in g.shards] == [2, 2, 1]
    g.set_state(q, qd)
    out = g.step(K, integrator=2, h=sc.h, record=7)
    for k in ("T", "V", "q", "qdot", "charts", "newton_iters", "status"):
        assert np.array_equal(out[k], ref[k]), k
    H = out["T"][-1, 0] + out["V"][-1, 0] - sc_V0(sc)
    assert abs(H - sc.Hexpected[1]) <= 1e-2          # trajectory 0 is the scene's own state: the reference's golden energy
    g.close()


def sc_V0(sc):
    from redmax_amd import BatchSim
    sim = BatchSim(sc, batch=1)
    q0, qd0 = sc.getQ()
    sim.set_state(q0[None, :], qd0[None, :])
    V0 = sim.energy()[1][0]
    sim.close()
    return V0


def test_group_with_ground_contact_and_split_calls():
    """Scene 11 (Free2D body over a frictional ground) through rmx_group_step_async + rmx_group_sync, in two consecutive calls
    (BDF2 continues across calls on every shard)."""
    from redmax_amd import BatchSim, GroupSim
    from redmax_amd.scenes import scenesRedMax
    sc = scenesRedMax(11)
    sc.init()
    q0, qd0 = sc.getQ()
    B = 6
    rng = np.random.default_rng(5)
    q = q0[None, :] + 0.02 * rng.standard_normal((B, sc.nr))
    qd = qd0[None, :] + 0.1 * rng.standard_normal((B, sc.nr))
    sim = BatchSim(sc, batch=B)
    sim.set_state(q, qd)
    a = sim.step_bdf2(300, h=sc.h, stats=True, history=True)
    b = sim.step_bdf2(300, h=sc.h, stats=True, history=True)
    qr, qdr = sim.get_state()
    sim.close()
    g = GroupSim(sc, B, devices=(0, 0))
    g.set_state(q, qd)
    g.step_async(300, integrator=2, h=sc.h, record=1)
    o1 = g.sync()
    g.step_async(300, integrator=2, h=sc.h, record=1)
    o2 = g.sync()
    qg, qdg = g.get_state()
    g.close()
    assert np.array_equal(o1["T"], a["T"]) and np.array_equal(o2["V"], b["V"])
    assert np.array_equal(o1["newton_iters"], a["newton_iters"]) and np.array_equal(o2["newton_iters"], b["newton_iters"])
    assert np.array_equal(qg, qr) and np.array_equal(qdg, qdr)


def test_async_entries_on_two_batches_of_one_host_thread():
    """rmx_step_history_async / rmx_step_bdf2_async / rmx_history_read on plain batches: two batches in flight at once, driven
    by one thread, each equal to its synchronous twin."""
    from redmax_amd import BatchSim, sceneTree, syntheticStates
    sc = sceneTree(64)
    sc.init()
    qs, _ = sc.getQ()
    B, K = 300, 12
    rng = np.random.default_rng(11)
    q = qs[None, :] + rng.uniform(-0.05, 0.05, (2 * B, sc.nr))
    qd = rng.uniform(-0.1, 0.1, (2 * B, sc.nr))
    refs = [_single(sc, q[i * B:(i + 1) * B], qd[i * B:(i + 1) * B], K, 1 + i) for i in range(2)]
    sims = [BatchSim(sc, batch=B) for _ in range(2)]
    for i, s in enumerate(sims):
        s.set_state(q[i * B:(i + 1) * B], qd[i * B:(i + 1) * B])
        s.opts.h = sc.h
        s.stats_reset()
    sims[0].step_history_async(K, integrator=1, record=3)
    sims[1].step_history_async(K, integrator=2, record=3)
    for i, s in enumerate(sims):
        s.sync()
        rec = s.history_read()
        st = s.stats_read()
        qf, qdf = s.get_state()
        for k in ("T", "V", "q", "qdot"):
            assert np.array_equal(rec[k], refs[i][k]), (i, k)
        assert np.array_equal(st["newton_iters"], refs[i]["newton_iters"])
        assert np.array_equal(qf, refs[i]["qf"]) and np.array_equal(qdf, refs[i]["qdf"])
    # BDF2 without a record, continued asynchronously: equals the synchronous continuation
    sims[1].step_bdf2_async(5)
    sims[1].sync()
    twin = BatchSim(sc, batch=B)
    twin.set_state(q[B:], qd[B:])
    twin.step_bdf2(K, h=sc.h)
    twin.step_bdf2(5, h=sc.h)
    assert np.array_equal(sims[1].get_state()[0], twin.get_state()[0])
    # a part that was not recorded cannot be read
    from redmax_amd._abi import RedMaxHipError
    sims[0].step_history_async(3, integrator=1, record=1)
    sims[0].sync()
    sims[0]._async = (3, 3)
    with pytest.raises(RedMaxHipError, match="not recorded"):
        sims[0].history_read()
    for s in sims + [twin]:
        s.close()


class _DevBuf:
    """[rows][cols] float64 device array through the HIP runtime the library itself is linked against (torch brings its own copy of the
    runtime: once the library has initialised the system one in this process, torch's no longer finds the GPU)."""
    _hip = None

    def __init__(self, rows, cols, fill=np.nan):
        import ctypes as C
        if _DevBuf._hip is None:
            _DevBuf._hip = C.CDLL("libamdhip64.so")
        self.C, self.shape = C, (rows, cols)
        self.ptr = C.c_void_p()
        self.nbytes = rows * cols * 8
        assert self._hip.hipMalloc(C.byref(self.ptr), C.c_size_t(max(self.nbytes, 8))) == 0
        self.fill(fill)

    def fill(self, v):
        h = np.full(self.shape, v)
        assert self._hip.hipMemcpy(self.ptr, h.ctypes.data_as(self.C.c_void_p), self.C.c_size_t(self.nbytes), 1) == 0    # hipMemcpyHostToDevice

    def get(self):
        out = np.empty(self.shape)
        assert self._hip.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), self.ptr, self.C.c_size_t(self.nbytes), 2) == 0  # hipMemcpyDeviceToHost
        return out

    def free(self):
        self._hip.hipFree(self.ptr)


def test_group_gather_device_rccl_and_copy_paths():
    """ABI 111, north_star's "RCCL over xGMI only for the final gather" inside the library: rmx_group_gather_device enqueues the gather
    of (q, qdot) on the shards' own streams behind their step kernels, with device-resident destinations.  On the one-GPU box: a
    ONE-shard group forms a one-rank RCCL communicator (ncclCommInitAll + ncclAllGather / the root form); a group that lists device 0
    twice takes the copy path (RCCL refuses duplicate GPUs), equal and unequal shards; the caller's buffers and the group's own
    (rmx_group_gather) must both hold exactly what rmx_group_get_state returns."""
    from redmax_amd import GroupSim, sceneChain, syntheticStates
    sc = sceneChain(8)
    sc.init()
    for B, devices, want_all, want_root in ((64, (0,), "rccl:allgather", "rccl:sendrecv"), (64, (0, 0), "copy", "copy"), (65, (0, 0, 0), "copy", "copy")):
        q, qd = syntheticStates(sc.nr, B)
        g = GroupSim(sc, B, devices=devices)
        g.set_state(q, qd)
        dq = [_DevBuf(B, sc.nr) for _ in devices]
        dqd = [_DevBuf(B, sc.nr) for _ in devices]
        g.step_async(6, integrator=1, h=sc.h, record=0)      # the gather is enqueued BEHIND the launches that are still in flight
        path = g.gather_device([t.ptr.value for t in dq], [t.ptr.value for t in dqd])
        g.sync()
        qf, qdf = g.get_state()
        assert path == want_all, (path, devices)
        assert not np.array_equal(qf, q)
        for t, u in zip(dq, dqd):
            assert np.array_equal(t.get(), qf) and np.array_equal(u.get(), qdf), devices
        # one root: only its destination is written
        root = len(devices) - 1
        for t in dq + dqd:
            t.fill(np.nan)
        ptr_q = [dq[s].ptr.value if s == root else None for s in range(len(devices))]
        ptr_qd = [dqd[s].ptr.value if s == root else None for s in range(len(devices))]
        assert g.gather_device(ptr_q, ptr_qd, root=root) == want_root
        assert np.array_equal(dq[root].get(), qf) and np.array_equal(dqd[root].get(), qdf)
        assert all(np.isnan(dq[s].get()).all() for s in range(len(devices)) if s != root)
        # the group's own destinations
        assert g.gather() == want_all
        for s in range(len(devices)):
            a, b = g.gathered_read(s)
            assert np.array_equal(a, qf) and np.array_equal(b, qdf)
        for t in dq + dqd:
            t.free()
        g.close()


def test_group_refuses_bad_plans():
    from redmax_amd import GroupSim, sceneChain
    from redmax_amd._abi import RedMaxHipError
    sc = sceneChain(4)
    sc.init()
    with pytest.raises(RedMaxHipError, match="more shards than trajectories"):
        GroupSim(sc, 2, devices=(0, 0, 0))
    with pytest.raises(RedMaxHipError, match="device index out of range"):
        GroupSim(sc, 4, devices=(0, 99))
