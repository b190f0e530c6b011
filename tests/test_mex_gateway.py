"""The MATLAB half of the boundary: mex/redmax_hip_mex.c.

MATLAB is not available, so the gateway is compiled against mex/stub/mex.h (MATLAB's documented C Matrix API signatures),
linked with mex/stub/mex_stub.c (host-memory implementation of the mx*/mex* calls it uses) and libredmax_hip.so, and
mexFunction is driven through ctypes:
  * CPU (-m "not gpu"): it builds warning-free, answers 'version', turns library errors into mexErrMsgIdAndTxt (no HIP
    device here => 'create' must fail loudly, never fall back), and reads exactly the struct fields that
    matlab/+redmax/flattenScene.m writes;
  * GPU (-m gpu): scenes go create -> set -> step(full history) -> get -> destroy through the gateway and must reproduce
    the reference's golden energies and the ctypes path bit for bit.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEX_C = os.path.join(ROOT, "mex", "redmax_hip_mex.c")

mxDOUBLE, mxINT32, mxUINT64 = 6, 12, 15


@pytest.fixture(scope="module")
def gw():
    import __graft_entry__ as ge
    ge.build()                                  # libredmax_hip.so must exist to link against
    so = ge.build_mex_stub(werror=True)
    L = C.CDLL(so)
    vp = C.c_void_p
    for name, res, args in (
            ("mxCreateDoubleMatrix", vp, (C.c_size_t, C.c_size_t, C.c_int)),
            ("mxCreateDoubleScalar", vp, (C.c_double,)),
            ("mxCreateNumericMatrix", vp, (C.c_size_t, C.c_size_t, C.c_int, C.c_int)),
            ("mxCreateNumericArray", vp, (C.c_size_t, C.POINTER(C.c_size_t), C.c_int, C.c_int)),
            ("mxCreateStructMatrix", vp, (C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_char_p))),
            ("mxCreateString", vp, (C.c_char_p,)),
            ("mxDestroyArray", None, (vp,)),
            ("mxGetData", vp, (vp,)),
            ("mxGetNumberOfElements", C.c_size_t, (vp,)),
            ("mxGetField", vp, (vp, C.c_size_t, C.c_char_p)),
            ("mxSetField", None, (vp, C.c_size_t, C.c_char_p, vp)),
            ("rmxstub_call", C.c_int, (C.c_int, C.POINTER(vp), C.c_int, C.POINTER(vp))),
            ("rmxstub_error", C.c_char_p, ()),
            ("rmxstub_ndim", C.c_size_t, (vp,)),
            ("rmxstub_dim", C.c_size_t, (vp, C.c_int)),
            ("rmxstub_class", C.c_int, (vp,))):
        f = getattr(L, name)
        f.restype, f.argtypes = res, list(args)
    return Gateway(L)


class MexError(RuntimeError):
    pass


class Gateway:
    """MATLAB-value marshalling for the stub: numpy arrays are passed in MATLAB's column-major shape."""

    def __init__(self, L):
        self.L = L

    def to_mx(self, v):
        L = self.L
        if isinstance(v, str):
            return L.mxCreateString(v.encode())
        if isinstance(v, dict):
            names = (C.c_char_p * len(v))(*[k.encode() for k in v])
            s = L.mxCreateStructMatrix(1, 1, len(v), names)
            for k, x in v.items():
                L.mxSetField(s, 0, k.encode(), self.to_mx(x))
            return s
        a = np.asarray(v)
        if a.dtype == np.uint64:
            cls = mxUINT64
        elif a.dtype.kind in "iu" and a.dtype == np.int32:
            cls = mxINT32
        else:
            cls, a = mxDOUBLE, a.astype(np.float64)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        dims = (C.c_size_t * a.ndim)(*a.shape)
        m = L.mxCreateNumericArray(a.ndim, dims, cls, 0)
        flat = np.asfortranarray(a).reshape(-1, order="F")
        C.memmove(L.mxGetData(m), flat.ctypes.data, flat.nbytes)
        return m

    def from_mx(self, m):
        L = self.L
        if not m:
            return None
        cls = L.rmxstub_class(m)
        nd = L.rmxstub_ndim(m)
        shape = tuple(L.rmxstub_dim(m, i) for i in range(nd))
        if cls == 2:          # struct: only the fields the gateway's 'info' returns
            return {k: self.from_mx(L.mxGetField(m, 0, k.encode())) for k in ("nr", "nm", "nsph", "batch", "idxR", "nshards", "devices",
                                                                               "shard_first", "shard_count")}
        n = int(np.prod(shape))
        if cls == 4:          # char array (mxCreateString: 16-bit code units) -> str
            buf = (C.c_char * (2 * n)).from_address(L.mxGetData(m)) if n else b""
            return "".join(chr(c) for c in np.frombuffer(bytes(buf), dtype=np.uint16))
        dt = {mxDOUBLE: np.float64, mxINT32: np.int32, mxUINT64: np.uint64}[cls]
        buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(L.mxGetData(m)) if n else b""
        return np.frombuffer(bytes(buf), dtype=dt).reshape(shape, order="F").copy()

    def call(self, nlhs, *args):
        L = self.L
        prhs = [self.to_mx(a) for a in args]
        plhs = (C.c_void_p * max(nlhs, 1))()
        rc = L.rmxstub_call(nlhs, plhs, len(prhs), (C.c_void_p * len(prhs))(*prhs))
        err = L.rmxstub_error().decode()
        out = [self.from_mx(plhs[i]) for i in range(nlhs)] if rc == 0 else None
        for a in prhs:
            L.mxDestroyArray(a)
        for i in range(nlhs):
            if plhs[i]:
                L.mxDestroyArray(plhs[i])
        if rc:
            raise MexError(err)
        return out[0] if nlhs == 1 else out


def flatten(scene):
    """What matlab/+redmax/flattenScene.m produces, from the Python mirror of the scene (same field names and shapes)."""
    d = scene.desc()
    n = d["njoints"]
    out = {
        "njoints": float(n), "parent": d["parent"].astype(np.int32), "type": d["type"].astype(np.int32),
        "axis": d["axis"].T, "E0_pj": d["E0_pj"].reshape(n, 4, 4).transpose(2, 1, 0), "E0_ji": d["E0_ji"].reshape(n, 4, 4).transpose(2, 1, 0),
        "I_i": d["I_i"].T, "qRest": d["qRest"], "tau": d["tau"], "stiffness": d["stiffness"], "damping": d["damping"],
        "qLimL": d["qLimL"], "qLimU": d["qLimU"], "qLimK": d["qLimK"], "qLimD": d["qLimD"], "plane": d["plane"].T,
        "grav": d["grav"].reshape(3, 1), "qRestR": d["qRestR"].reshape(-1, 1),
    }
    if d.get("contact") is not None and np.any(d["contact"]):
        g = d["ground"]
        out.update(contact=d["contact"].astype(np.int32), sides=d["sides"].T, groundE=np.asarray(g["E"], dtype=np.float64).reshape(4, 4),
                   kn=g["kn"], kt=g["kt"], mu=g["mu"], kd=g["kd"])
        gb = d.get("ground_body") or {"E": np.stack([np.asarray(g["E"], dtype=np.float64)] * n), "kn": np.full(n, g["kn"]),
                                      "kt": np.full(n, g["kt"]), "mu": np.full(n, g["mu"]), "kd": np.full(n, g["kd"])}
        out.update(groundE_body=np.stack([np.asarray(E, dtype=np.float64).reshape(4, 4).T.reshape(16) for E in gb["E"]]).T,
                   kn_body=np.asarray(gb["kn"], dtype=np.float64).reshape(1, n), kt_body=np.asarray(gb["kt"], dtype=np.float64).reshape(1, n),
                   mu_body=np.asarray(gb["mu"], dtype=np.float64).reshape(1, n), kd_body=np.asarray(gb["kd"], dtype=np.float64).reshape(1, n))
    return out


def test_gateway_builds_warning_free_and_answers_version(gw):
    assert gw.call(1, "version") == 111


def test_gateway_reports_errors_the_matlab_way(gw):
    with pytest.raises(MexError, match="unknown command"):
        gw.call(0, "frobnicate")
    with pytest.raises(MexError, match="handle"):
        gw.call(1, "get", np.array([[12345]], dtype=np.uint64))
    with pytest.raises(MexError, match="desc.parent is required"):
        gw.call(1, "create", {"njoints": 3.0}, 1.0)


def test_create_without_a_device_fails_loudly(gw):
    if gw.call(1, "devices") > 0:          # (what the library itself sees: torch may not see a device the HIP runtime does)
        pytest.skip("a HIP device is present")
    from redmax_amd.scenes import scenesRedMax
    sc = scenesRedMax(0)
    sc.init()
    with pytest.raises(MexError, match="no HIP device"):
        gw.call(1, "create", flatten(sc), 1.0)


def test_flattenScene_writes_the_fields_the_gateway_reads():
    src = open(MEX_C).read()
    create = src[src.index("static void cmd_create"):src.index("static void cmd_destroy")]
    read = set(re.findall(r'(?:f64|i32|scalar_field|field)\(s, "(\w+)"', create))
    m = open(os.path.join(ROOT, "matlab", "+redmax", "flattenScene.m")).read()
    written = set(re.findall(r"desc\.(\w+)\b", m))
    assert read == written, (sorted(read - written), sorted(written - read))
    # and the Python stand-in used by the GPU tests below produces the same set for a contact scene
    from redmax_amd.scenes import scenesRedMax
    sc = scenesRedMax(11)
    sc.init()
    assert set(flatten(sc)) == written


def test_matlab_shims_do_not_copy_reference_files():
    """The MATLAB side adds files to the reference (package functions, a wrapper class, drivers); it must not carry any of
    the reference's classes (Scene.m, Joint.m ...), which stay the reference's own."""
    names = set(os.listdir(os.path.join(ROOT, "matlab", "+redmax")))
    assert names == {"flattenScene.m", "HipSim.m", "simLoopHip.m", "simLoopBatchHip.m", "runDriverHip.m"}


def test_command_table_matches_the_matlab_callers():
    """Every command string matlab/ passes to redmax_hip_mex exists in the gateway's dispatch, and every command of the gateway
    except the two process-level queries is reachable from redmax.HipSim (the MATLAB class a user holds)."""
    src = open(MEX_C).read()
    table = set(re.findall(r'!strcmp\(cmd, "(\w+)"\)', src[src.index("void mexFunction"):]))
    used, hipsim = set(), set()
    for root, _, files in os.walk(os.path.join(ROOT, "matlab")):
        for f in files:
            if f.endswith(".m"):
                cmds = set(re.findall(r"redmax_hip_mex\('(\w+)'", open(os.path.join(root, f)).read()))
                used |= cmds
                if f == "HipSim.m":
                    hipsim = cmds
    assert used <= table, sorted(used - table)
    assert table - hipsim == {"version", "devices"}, sorted(table - hipsim)
    assert {"step_async", "sync", "timing"} <= hipsim          # the multi-device / asynchronous commands (ABI 107)


def test_flattenScene_lists_further_force_objects_as_fixed_massless_children():
    """A floor and a wall on one cuboid (the reference's forces are a list, Force.m:26-56): up to round 4 flattenScene.m raised, as
    redmax.py did; now both list the second force as a fixed, massless child of the body's joint (one force object per listing entry
    is what the library takes).  The .m file cannot be executed here: the block is checked to be there, to add a JointFixed entry
    with zero inertia and the body's own transform, and to sit BEFORE the writes of the force's own fields."""
    m = open(os.path.join(ROOT, "matlab", "+redmax", "flattenScene.m")).read()
    guard = m.index("if desc.contact(hit)")
    block = m[guard:m.index("desc.contact(hit) = 1;")]
    assert "error(" not in block
    for line in ("desc.njoints = m;", "desc.parent(m) = hit - 1;", "desc.type(m) = 0;", "desc.E0_ji(:,:,m) = f.cuboid.E0_ji;",
                 "desc.I_i(:,m) = zeros(6,1);", "hit = m;"):
        assert line in block, line
    # the Python mirror does the same, and names the literal listing for checkers
    from redmax_amd.scenes import sceneChainFloorAndWall
    sc = sceneChainFloorAndWall(3)
    sc.init()
    d = sc.desc()
    assert d["njoints"] == 6 and list(d["type"][3:]) == [0, 0, 0] and list(d["parent"][3:]) == [0, 1, 2] and not d["I_i"][3:].any()
    assert sc.desc_literal()["njoints"] == 3 and len(sc.desc_literal()["extra_forces"]) == 3


@pytest.mark.gpu
@pytest.mark.parametrize("sid,itype", [(0, 1), (2, 1), (3, 2), (6, 2), (7, 2), (11, 2)])
def test_scene_through_the_gateway_meets_the_golden(gw, sid, itype):
    """driverRedMaxBDF1/2(sceneID,true) as matlab/+redmax/simLoopHip.m runs it: flatten, create, set, step with the full
    history, get, destroy; H(end) against Hexpected (Scene.plotEnergies: |dH| <= 1e-2) and against the ctypes path."""
    from redmax_amd import BatchSim
    from redmax_amd.scenes import scenesRedMax
    sc = scenesRedMax(sid)
    sc.init()
    q0, qd0 = sc.getQ()
    h = gw.call(1, "create", flatten(sc), 1.0, 0.0)
    info = gw.call(1, "info", h)
    assert int(info["nr"][0, 0]) == sc.nr and int(info["nm"][0, 0]) == sc.nm
    gw.call(0, "set", h, q0.reshape(-1, 1), qd0.reshape(-1, 1))
    T0, V0 = gw.call(2, "energy", h)
    T, V, st, Q, Qd, Ch = gw.call(6, "step", h, float(itype), sc.h, float(sc.nsteps))
    q, qd = gw.call(2, "get", h)
    charts = gw.call(1, "getcharts", h)
    gw.call(0, "destroy", h)
    assert T.shape == (1, sc.nsteps) and Q.shape == (sc.nr, 1, sc.nsteps) and st.shape == (1, 3)
    H = T[0, -1] + V[0, -1] - V0[0, 0]
    assert abs(H - sc.Hexpected[itype - 1]) <= 1e-2, (sid, H, sc.Hexpected[itype - 1])
    assert np.array_equal(Q[:, 0, -1], q[:, 0]) and np.array_equal(Qd[:, 0, -1], qd[:, 0])
    sim = BatchSim(sc, batch=1)
    sim.set_state(q0[None, :], qd0[None, :])
    out = (sim.step_bdf1 if itype == 1 else sim.step_bdf2)(sc.nsteps, h=sc.h, stats=True, history="full")
    qc, qdc = sim.get_state()
    assert np.array_equal(qc[0], q[:, 0]) and np.array_equal(qdc[0], qd[:, 0])          # same library, same bits
    assert np.array_equal(out["T"][:, 0], T[0]) and st[0, 0] == out["newton_iters"][0]
    if sim.nsph:
        assert np.array_equal(sim.charts()[0], charts[:, 0])
        assert Ch.shape == (sim.nsph, 1, sc.nsteps) and np.array_equal(Ch[:, 0, :].T, out["charts"][:, 0, :])
    sim.close()


@pytest.mark.gpu
def test_batched_eval_and_state_layout_through_the_gateway(gw, oracle_lib):
    """nr x B column-major MATLAB matrices are the ABI's [B][nr] arrays: a batch of 3 different states must come back per
    column, and eval's H (nr x nr x B) must be the oracle's H per trajectory."""
    from redmax_amd.scenes import scenesRedMax, syntheticStates
    sc = scenesRedMax(2)
    sc.init()
    B = 3
    q, qd = syntheticStates(sc.nr, B, first=5)
    h = gw.call(1, "create", flatten(sc), float(B))
    gw.call(0, "set", h, q.T, qd.T)
    q2, qd2 = gw.call(2, "get", h)
    assert np.array_equal(q2, q.T) and np.array_equal(qd2, qd.T)
    g, H = gw.call(2, "eval", h, q.T, q.T, (q + sc.h * qd).T, sc.h)
    v = 1e-2 * np.random.default_rng(8).standard_normal((B, sc.nr))
    M, f, K, D, dMv = gw.call(5, "values", h, q.T, qd.T, v.T)          # computeValues' full output through the gateway
    gw.call(0, "destroy", h)
    for b in range(B):
        o = oracle_lib.Oracle(sc.desc())
        go, Ho = o.eval_bdf1(q[b], q[b], qd[b], sc.h)
        assert np.linalg.norm(g[:, b] - go) <= 1e-11 * np.linalg.norm(go)
        assert np.linalg.norm(H[:, :, b] - Ho) <= 1e-11 * np.linalg.norm(Ho)
        o.set_state(q[b], qd[b])
        Mo, fo, dMo, Ko, Do = o.compute_values(deriv=True)
        scale = np.linalg.norm(Mo) + np.linalg.norm(Do) + np.linalg.norm(Ko)
        assert np.linalg.norm(M[:, :, b] - Mo) <= 1e-11 * scale and np.linalg.norm(f[:, b] - fo) <= 1e-11 * np.linalg.norm(fo)
        assert np.linalg.norm(K[:, :, b] - Ko) <= 1e-10 * scale and np.linalg.norm(D[:, :, b] - Do) <= 1e-11 * scale
        assert np.linalg.norm(dMv[:, :, b] - np.einsum("rci,c->ri", dMo, v[b])) <= 1e-10 * scale


@pytest.mark.gpu
def test_two_shards_through_the_gateway_async_and_sync(gw):
    """ABI 107, the MATLAB half: 'create' with a device VECTOR shards the batch (here two shards on device 0), 'step' runs them
    concurrently and gathers, 'step_async' + 'sync' is the same launch split in two calls.  Results must equal the one-shard
    handle bit for bit, in every output (T, V, stats, Q, Qdot), and 'timing' must show the two launches overlapping."""
    from redmax_amd.scenes import sceneChain, syntheticStates
    sc = sceneChain(32)
    sc.init()
    B, K = 2048, 20
    q, qd = syntheticStates(sc.nr, B)
    d = flatten(sc)
    outs = []
    for devs, use_async in ((np.array([0.0]), False), (np.array([0.0, 0.0]), False), (np.array([0.0, 0.0]), True)):
        h = gw.call(1, "create", d, float(B), devs)
        info = gw.call(1, "info", h)
        assert int(info["nshards"][0, 0]) == devs.size and info["shard_count"].sum() == B
        gw.call(0, "set", h, q.T, qd.T)
        if use_async:
            gw.call(0, "step_async", h, 1.0, 1e-2, float(K), {}, 3.0)
            with pytest.raises(MexError, match="in flight"):
                gw.call(1, "step", h, 1.0, 1e-2, 1.0)
            # every command that touches the state, the scratch buffers or the counters waits for 'sync' too (round-4 advice);
            # 'info' and 'timing' stay available
            for cmd in (("get",), ("set", q.T, qd.T), ("gather",), ("energy",), ("ticks",), ("eval", q.T, q.T, q.T, 1e-2), ("values", q.T, qd.T)):
                with pytest.raises(MexError, match="in flight"):
                    gw.call(1, cmd[0], h, *cmd[1:])
            gw.call(1, "info", h)
            gw.call(1, "timing", h)
            T, V, st, Q, Qd = gw.call(5, "sync", h)
        else:
            T, V, st, Q, Qd = gw.call(5, "step", h, 1.0, 1e-2, float(K))
        q1, qd1 = gw.call(2, "get", h)
        # 'gather' (ABI 111): the final gather with device-resident destinations - a one-rank RCCL communicator for the one-shard handle,
        # device-to-device copies for the two shards that share device 0 (RCCL refuses a clique that names a GPU twice) - every shard's
        # device, and shard 0's alone, must hold exactly what 'get' returns
        for root in ((), (0.0,)):
            gq, gqd, path = gw.call(3, "gather", h, *root)
            assert np.array_equal(gq, q1) and np.array_equal(gqd, qd1), (devs, root)
            assert path == ("copy" if devs.size == 2 else ("rccl:allgather" if not root else "rccl:sendrecv")), (path, devs, root)
        wall, k, t0, t1 = gw.call(4, "timing", h)
        gw.call(0, "destroy", h)
        outs.append((T, V, st, Q, Qd, q1, qd1))
        if devs.size == 2:
            assert t0[0, 1] < t1[0, 0], ("the second shard's launch did not start before the first one ended", t0, t1)
            assert wall[0, 0] < k.sum(), (wall, k)
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b)
