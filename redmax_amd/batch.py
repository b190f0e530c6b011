"""Batched rollouts of one scene on one MI355X: the host-side handle around the C ABI.

``BatchSim`` is what replaces the body of ``simLoop`` (matlab-diff/driverRedMaxBDF1.m:57-91):
the scene is flattened once (``Scene.desc()``), B independent (q, qdot) states live in HBM and
``step_bdf1(nsteps)`` advances all of them in one kernel launch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi


class BatchSim:
    def __init__(self, scene_or_desc, batch=1, device=0):
        d = scene_or_desc.desc() if hasattr(scene_or_desc, "desc") else scene_or_desc
        self._L = _abi.lib()
        self._async = None                                 # (nsteps, record) of a step_history_async whose record has not been replaced
        self._desc, self._keep = _abi.make_desc(d)
        self._model = C.c_void_p()
        self._batch = C.c_void_p()
        _abi.check(self._L.rmx_model_create(C.byref(self._desc), int(device), C.byref(self._model)), "rmx_model_create")
        self.nr = self._L.rmx_model_nr(self._model)
        self.nm = self._L.rmx_model_nm(self._model)
        gc = _abi.make_ground_contact(d, self._keep)      # scene.forces: ForceGroundCuboid
        if gc is not None:
            _abi.check(self._L.rmx_model_set_ground_contact(self._model, C.byref(gc)), "rmx_model_set_ground_contact")
        self.nsph = self._L.rmx_model_nsph(self._model)
        self.B = int(batch)
        self.device = int(device)
        _abi.check(self._L.rmx_batch_create(self._model, self.B, C.byref(self._batch)), "rmx_batch_create")
        self.opts = _abi.Opts()
        self._L.rmx_opts_default(C.byref(self.opts))

    def close(self):
        if getattr(self, "_batch", None):
            self._L.rmx_batch_destroy(self._batch)
            self._batch = None
        if getattr(self, "_model", None):
            self._L.rmx_model_destroy(self._model)
            self._model = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Joint.setQ / getQ for the batch, reference (leaf-to-root) DOF order ----
    def _arr(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != (self.B, self.nr):
            a = np.ascontiguousarray(np.broadcast_to(a, (self.B, self.nr)))
        return a

    def charts(self):
        """JointSpherical.chart of every spherical joint and trajectory, [B][nsph] (reference numbering 1..12)."""
        c = np.zeros((self.B, max(self.nsph, 1)), dtype=np.int32)
        if self.nsph:
            c = np.zeros((self.B, self.nsph), dtype=np.int32)
            _abi.check(self._L.rmx_get_charts(self._batch, _abi.iptr(c)), "rmx_get_charts")
            return c
        return c[:, :0]

    def set_charts(self, charts):
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(charts, dtype=np.int32), (self.B, self.nsph)))
        _abi.check(self._L.rmx_set_charts(self._batch, _abi.iptr(c)), "rmx_set_charts")

    def idxR(self):
        idx = np.zeros(self._desc.njoints, dtype=np.int32)
        _abi.check(self._L.rmx_model_idxR(self._model, _abi.iptr(idx)), "rmx_model_idxR")
        return idx

    def set_state(self, q, qdot):
        q, qdot = self._arr(q), self._arr(qdot)
        _abi.check(self._L.rmx_set_state(self._batch, _abi.dptr(q), _abi.dptr(qdot)), "rmx_set_state")

    def get_state(self):
        q = np.empty((self.B, self.nr))
        qd = np.empty((self.B, self.nr))
        _abi.check(self._L.rmx_get_state(self._batch, _abi.dptr(q), _abi.dptr(qd)), "rmx_get_state")
        return q, qd

    def get_state_device(self, q_ptr, qdot_ptr):
        """Copy the state into caller-owned DEVICE buffers (e.g. torch tensors' data_ptr())."""
        _abi.check(self._L.rmx_get_state_device(self._batch, C.c_void_p(q_ptr), C.c_void_p(qdot_ptr)), "rmx_get_state_device")

    def set_state_device(self, q_ptr, qdot_ptr):
        _abi.check(self._L.rmx_set_state_device(self._batch, C.c_void_p(q_ptr), C.c_void_p(qdot_ptr)), "rmx_set_state_device")

    # ---- evalBDF1 & friends: g (and H) for every trajectory ----
    def eval_residual(self, q, qA, qB, eta, want_H=True):
        q, qA, qB = self._arr(q), self._arr(qA), self._arr(qB)
        g = np.empty((self.B, self.nr))
        H = np.empty((self.B, self.nr * self.nr)) if want_H else None
        _abi.check(self._L.rmx_eval(self._batch, _abi.dptr(q), _abi.dptr(qA), _abi.dptr(qB), float(eta), _abi.dptr(g), _abi.dptr(H)), "rmx_eval")
        if want_H:
            return g, H.reshape(self.B, self.nr, self.nr).transpose(0, 2, 1)   # column-major -> [b][row][col]
        return g

    def eval_bdf1(self, q1, q0, qdot0, h, want_H=True):
        q0 = self._arr(q0)
        return self.eval_residual(q1, q0, q0 + h * self._arr(qdot0), h, want_H)

    def eval_mfd(self, q, qdot):
        """computeValues (driverRedMaxBDF1.m:190-243) at (q, qdot): M [B][nr][nr], f [B][nr], D = df/dqdot [B][nr][nr]."""
        q, qdot = self._arr(q), self._arr(qdot)
        M = np.empty((self.B, self.nr * self.nr))
        D = np.empty((self.B, self.nr * self.nr))
        f = np.empty((self.B, self.nr))
        _abi.check(self._L.rmx_eval_mfd(self._batch, _abi.dptr(q), _abi.dptr(qdot), _abi.dptr(M), _abi.dptr(f), _abi.dptr(D)), "rmx_eval_mfd")
        sh = (self.B, self.nr, self.nr)
        return M.reshape(sh).transpose(0, 2, 1), f, D.reshape(sh).transpose(0, 2, 1)      # column-major -> [b][row][col]

    def compute_values(self, q, qdot, v=None, tensor=False):
        """computeValues' full output (driverRedMaxBDF1.m:188-243) at (q, qdot), any tree size: dict with M, D, K [B][nr][nr], f [B][nr],
        dMv (with v: column i = dMdq(:,:,i) v) and, with tensor=True, dMdq [B][nr][nr][nr] indexed [b][r][c][i]."""
        q, qdot = self._arr(q), self._arr(qdot)
        nr, B = self.nr, self.B
        out = {k: np.empty((B, nr * nr)) for k in ("M", "D", "K")}
        out["f"] = np.empty((B, nr))
        vv = None
        if v is not None:
            vv = self._arr(v)
            out["dMv"] = np.empty((B, nr * nr))
        if tensor:
            out["dMdq"] = np.empty((B, nr * nr * nr))
        _abi.check(self._L.rmx_compute_values(self._batch, _abi.dptr(q), _abi.dptr(qdot), _abi.dptr(vv) if vv is not None else None,
                                              _abi.dptr(out["M"]), _abi.dptr(out["f"]), _abi.dptr(out["D"]), _abi.dptr(out["K"]),
                                              _abi.dptr(out["dMv"]) if vv is not None else None,
                                              _abi.dptr(out["dMdq"]) if tensor else None), "rmx_compute_values")
        sh = (B, nr, nr)
        for k in ("M", "D", "K", "dMv"):
            if k in out:
                out[k] = out[k].reshape(sh).transpose(0, 2, 1)          # column-major -> [b][row][col]
        if tensor:
            out["dMdq"] = out["dMdq"].reshape((B, nr, nr, nr)).transpose(0, 3, 2, 1)    # (i, c, r) storage order -> [b][r][c][i]
        return out

    # ---- stepping ----
    def _step(self, fn, nsteps, h, stats, history):
        if h is not None:
            self.opts.h = float(h)
        self._async = None                                 # any step call replaces the record of an earlier step_history_async
        st = None
        out = {}
        if stats:
            out["newton_iters"] = np.zeros(self.B, dtype=np.int32)
            out["ls_halvings"] = np.zeros(self.B, dtype=np.int32)
            out["status"] = np.zeros(self.B, dtype=np.int32)
            st = _abi.Stats(_abi.iptr(out["newton_iters"]), _abi.iptr(out["ls_halvings"]), _abi.iptr(out["status"]))
        T = V = None
        if history:
            T = np.empty((nsteps, self.B))
            V = np.empty((nsteps, self.B))
            out["T"], out["V"] = T, V
        stp = C.byref(st) if st is not None else None
        if history == "full":                     # Scene.saveHistory: q, qdot of every step as well (Scene.m:134-161)
            out["q"] = np.empty((nsteps, self.B, self.nr))
            out["qdot"] = np.empty((nsteps, self.B, self.nr))
            out["charts"] = np.full((nsteps, self.B, self.nsph), 7, dtype=np.int32)     # JointSpherical.chart after every step
            hist = _abi.History(_abi.dptr(T), _abi.dptr(V), _abi.dptr(out["q"]), _abi.dptr(out["qdot"]),
                                _abi.iptr(out["charts"]) if self.nsph else None)
            _abi.check(self._L.rmx_step_history(self._batch, C.byref(self.opts), int(nsteps), 1 if fn == "bdf1" else 2, stp,
                                                C.byref(hist)), "rmx_step_history")
        else:
            f = self._L.rmx_step_bdf1 if fn == "bdf1" else self._L.rmx_step_bdf2
            _abi.check(f(self._batch, C.byref(self.opts), int(nsteps), stp, _abi.dptr(T), _abi.dptr(V)), "rmx_step")
        out["ms"] = self._L.rmx_last_step_ms(self._batch)
        return out

    def step_bdf1(self, nsteps, h=None, stats=False, history=False):
        """history: False | True (T, V per step) | "full" (T, V, q, qdot per step: Scene.saveHistory)."""
        return self._step("bdf1", nsteps, h, stats, history)

    def step_bdf2(self, nsteps, h=None, stats=False, history=False):
        return self._step("bdf2", nsteps, h, stats, history)

    def step_euler(self, nsteps, h, history=False):
        """euler() of matlab-simple/testRedMax.m:67-109 (linearly-implicit Euler, config 1)."""
        out = {}
        T = V = None
        if history:
            T = np.empty((nsteps, self.B))
            V = np.empty((nsteps, self.B))
            out["T"], out["V"] = T, V
        _abi.check(self._L.rmx_step_euler(self._batch, float(h), int(nsteps), _abi.dptr(T), _abi.dptr(V)), "rmx_step_euler")
        out["ms"] = self._L.rmx_last_step_ms(self._batch)
        return out

    def adjoint_bdf2(self, nsteps, h, task, p, stats=False):
        """taskObjective of driverRedMaxAdjointBDF2.m:38-62 (TaskBDF2PointPos): SDIRK2 start step + BDF2 forward, TaskBDF2.calcFinal
        backward.  Arguments and results as adjoint_bdf1."""
        return self.adjoint_bdf1(nsteps, h, task, p, stats, _fn="rmx_adjoint_bdf2")

    def adjoint_bdf1(self, nsteps, h, task, p, stats=False, _fn="rmx_adjoint_bdf1"):
        """taskObjective (driverRedMaxAdjointBDF1.m:39-62) for every trajectory: forward rollout from the current state
        under torques pscale*p, then the backward sweep.  task: dict(body, xlocal, xtarget, t | step, pscale, wreg, wpos);
        p: [B][nr].  Returns (P[B], dPdp[B][nr], info)."""
        tk = _abi.TaskPointPos()
        tk.body = int(task["body"])
        for i in range(3):
            tk.xlocal[i] = float(task["xlocal"][i])
            tk.xtarget[i] = float(task["xtarget"][i])
        tk.step = int(task["step"]) if "step" in task else int(round(float(task["t"]) / float(h)))
        tk.pscale, tk.wreg, tk.wpos = float(task["pscale"]), float(task["wreg"]), float(task["wpos"])
        opts = _abi.Opts()
        C.memmove(C.byref(opts), C.byref(self.opts), C.sizeof(opts))
        opts.h = float(h)
        opts.iterMaxPerDof = 5                      # driverRedMaxAdjointBDF1.m:108
        p = self._arr(p)
        P = np.empty(self.B)
        dPdp = np.empty((self.B, self.nr))
        info = {}
        st = None
        if stats:
            info["newton_iters"] = np.zeros(self.B, dtype=np.int32)
            info["status"] = np.zeros(self.B, dtype=np.int32)
            st = _abi.Stats(_abi.iptr(info["newton_iters"]), None, _abi.iptr(info["status"]))
        _abi.check(getattr(self._L, _fn)(self._batch, C.byref(opts), int(nsteps), C.byref(tk), _abi.dptr(p), _abi.dptr(P),
                                         _abi.dptr(dPdp), C.byref(st) if st is not None else None), _fn)
        info["ms"] = self._L.rmx_last_step_ms(self._batch)
        return P, dPdp, info

    def adjoint_bdf1_device(self, nsteps, h, task, p_ptr, P_ptr, dPdp_ptr, stats=False, _fn="rmx_adjoint_bdf1_device"):
        """adjoint_bdf1 with DEVICE pointers (integers, e.g. torch.Tensor.data_ptr()) for p [B][nr], P [B] and dPdp [B][nr]: nothing
        crosses the host boundary but the optional counters.  Returns info."""
        tk = _abi.TaskPointPos()
        tk.body = int(task["body"])
        for i in range(3):
            tk.xlocal[i] = float(task["xlocal"][i])
            tk.xtarget[i] = float(task["xtarget"][i])
        tk.step = int(task["step"]) if "step" in task else int(round(float(task["t"]) / float(h)))
        tk.pscale, tk.wreg, tk.wpos = float(task["pscale"]), float(task["wreg"]), float(task["wpos"])
        opts = _abi.Opts()
        C.memmove(C.byref(opts), C.byref(self.opts), C.sizeof(opts))
        opts.h = float(h)
        opts.iterMaxPerDof = 5                      # driverRedMaxAdjointBDF1.m:108
        info = {}
        st = None
        if stats:
            info["newton_iters"] = np.zeros(self.B, dtype=np.int32)
            info["status"] = np.zeros(self.B, dtype=np.int32)
            st = _abi.Stats(_abi.iptr(info["newton_iters"]), None, _abi.iptr(info["status"]))
        _abi.check(getattr(self._L, _fn)(self._batch, C.byref(opts), int(nsteps), C.byref(tk), C.c_void_p(p_ptr), C.c_void_p(P_ptr),
                                         C.c_void_p(dPdp_ptr), C.byref(st) if st is not None else None), _fn)
        info["ms"] = self._L.rmx_last_step_ms(self._batch)
        return info

    def adjoint_bdf2_device(self, nsteps, h, task, p_ptr, P_ptr, dPdp_ptr, stats=False):
        return self.adjoint_bdf1_device(nsteps, h, task, p_ptr, P_ptr, dPdp_ptr, stats=stats, _fn="rmx_adjoint_bdf2_device")

    def last_step_kernel(self):
        """Label of the step kernel the last step call launched (rmx_last_step_kernel): which size / batch / environment dependent
        variant the library chose."""
        return self._L.rmx_last_step_kernel(self._batch).decode()

    def step_ticks(self):
        """Shader-clock ticks each rollout's wavefront spent in the kernel(s) of the last step call (rmx_step_ticks): [B] uint64."""
        t = np.zeros(self.B, dtype=np.uint64)
        _abi.check(self._L.rmx_step_ticks(self._batch, t.ctypes.data_as(C.POINTER(C.c_ulonglong))), "rmx_step_ticks")
        return t

    def step_bdf1_async(self, nsteps, h=None):
        if h is not None:
            self.opts.h = float(h)
        self._async = None
        _abi.check(self._L.rmx_step_bdf1_async(self._batch, C.byref(self.opts), int(nsteps)), "rmx_step_bdf1_async")

    def step_bdf2_async(self, nsteps, h=None):
        if h is not None:
            self.opts.h = float(h)
        self._async = None
        _abi.check(self._L.rmx_step_bdf2_async(self._batch, C.byref(self.opts), int(nsteps)), "rmx_step_bdf2_async")

    def step_history_async(self, nsteps, integrator=1, record=_abi.REC_ENERGY | _abi.REC_STATE, h=None):
        """simLoop enqueued, nothing waited for; `record` (REC_ENERGY | REC_STATE | REC_CHARTS) stays on the device until
        history_read() (after sync())."""
        if h is not None:
            self.opts.h = float(h)
        self._async = (int(nsteps), int(record))
        _abi.check(self._L.rmx_step_history_async(self._batch, C.byref(self.opts), int(nsteps), int(integrator), int(record)), "rmx_step_history_async")

    def history_read(self):
        if self._async is None:
            raise _abi.RedMaxHipError("history_read: no step_history_async record is outstanding on this batch "
                                      "(the last step call was synchronous, unrecorded, or has replaced it)")
        nsteps, record = self._async
        out = {}
        hist = _abi.History()
        if record & _abi.REC_ENERGY:
            out["T"], out["V"] = np.empty((nsteps, self.B)), np.empty((nsteps, self.B))
            hist.T, hist.V = _abi.dptr(out["T"]), _abi.dptr(out["V"])
        if record & _abi.REC_STATE:
            out["q"], out["qdot"] = np.empty((nsteps, self.B, self.nr)), np.empty((nsteps, self.B, self.nr))
            hist.q, hist.qdot = _abi.dptr(out["q"]), _abi.dptr(out["qdot"])
        if record & _abi.REC_CHARTS and self.nsph:
            out["charts"] = np.full((nsteps, self.B, self.nsph), 7, dtype=np.int32)
            hist.charts = _abi.iptr(out["charts"])
        _abi.check(self._L.rmx_history_read(self._batch, C.byref(hist)), "rmx_history_read")
        return out

    def sync(self):
        _abi.check(self._L.rmx_sync(self._batch), "rmx_sync")
        return self._L.rmx_last_step_ms(self._batch)

    def profile_phases(self, reps=20, h=1e-2):
        """Mean cycles per wavefront of (g-eval, g+H-eval, LU solve, 2 reductions) at the current state."""
        c = np.zeros(16)
        _abi.check(self._L.rmx_profile_phases(self._batch, int(reps), float(h), _abi.dptr(c)), "rmx_profile_phases")
        d = dict(zip(("eval_g", "eval_gH", "lu", "reductions"), c[:4]))
        d["gH_stamps"] = dict(zip(("joint_T", "jump_E", "screw_phi", "xi_beta", "inertia_w", "lds_write", "suffix_scan",
                                   "subtree_read", "residual", "H_vectors", "col_write", "H_columns"), c[4:]))
        return d

    def stats_reset(self):
        _abi.check(self._L.rmx_stats_reset(self._batch), "rmx_stats_reset")

    def stats_read(self):
        out = {k: np.zeros(self.B, dtype=np.int32) for k in ("newton_iters", "ls_halvings", "status")}
        st = _abi.Stats(_abi.iptr(out["newton_iters"]), _abi.iptr(out["ls_halvings"]), _abi.iptr(out["status"]))
        _abi.check(self._L.rmx_stats_read(self._batch, C.byref(st)), "rmx_stats_read")
        return out

    def energy(self):
        T = np.empty(self.B)
        V = np.empty(self.B)
        _abi.check(self._L.rmx_energy(self._batch, _abi.dptr(T), _abi.dptr(V)), "rmx_energy")
        return T, V


class GroupSim:
    """The whole batch over a LIST of devices (rmx_group_*, include/redmax_hip.h): one model + batch per listed device, contiguous
    shards, every array the whole batch.  ``step`` launches all shards before it waits for the first - the multi-device simLoop a
    single host thread (MATLAB: matlab/+redmax/HipSim.m with a device vector) can drive; a device may be listed more than once."""

    def __init__(self, scene_or_desc, batch, devices=(0,)):
        d = scene_or_desc.desc() if hasattr(scene_or_desc, "desc") else scene_or_desc
        self._L = _abi.lib()
        self._desc, self._keep = _abi.make_desc(d)
        gc = _abi.make_ground_contact(d, self._keep)
        dev = np.ascontiguousarray(list(devices), dtype=np.int32)
        self._g = C.c_void_p()
        _abi.check(self._L.rmx_group_create(C.byref(self._desc), C.byref(gc) if gc is not None else None, int(batch), _abi.iptr(dev),
                                            len(dev), C.byref(self._g)), "rmx_group_create")
        self.B = int(batch)
        self.nshards = self._L.rmx_group_nshards(self._g)
        m0 = self._L.rmx_group_shard_model(self._g, 0)
        self.nr, self.nsph = self._L.rmx_model_nr(m0), self._L.rmx_model_nsph(m0)
        self.shards = []
        for s in range(self.nshards):
            dv, f, c = C.c_int(), C.c_int(), C.c_int()
            _abi.check(self._L.rmx_group_shard(self._g, s, C.byref(dv), C.byref(f), C.byref(c)), "rmx_group_shard")
            self.shards.append((dv.value, f.value, c.value))
        self.opts = _abi.Opts()
        self._L.rmx_opts_default(C.byref(self.opts))
        self._async = None

    def close(self):
        if getattr(self, "_g", None):
            self._L.rmx_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _arr(self, a):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (self.B, self.nr)))

    def set_state(self, q, qdot):
        q, qdot = self._arr(q), self._arr(qdot)
        _abi.check(self._L.rmx_group_set_state(self._g, _abi.dptr(q), _abi.dptr(qdot)), "rmx_group_set_state")

    def get_state(self):
        q, qd = np.empty((self.B, self.nr)), np.empty((self.B, self.nr))
        _abi.check(self._L.rmx_group_get_state(self._g, _abi.dptr(q), _abi.dptr(qd)), "rmx_group_get_state")
        return q, qd

    GATHER_ALL = -1      # RMX_GATHER_ALL

    def gather_device(self, d_q, d_qdot, root=GATHER_ALL):
        """The final gather with device-resident destinations (rmx_group_gather_device): d_q[s], d_qdot[s] = raw device pointers
        (ints; e.g. torch tensor.data_ptr()) of [batch][nr] float64 arrays on shard s's device, None for a shard that receives nothing.
        root: a shard index, or GATHER_ALL.  RCCL (single-process clique over the group's devices) when the devices are pairwise
        distinct, device-to-device copies when a device is listed twice.  Returns how the gather travelled (rmx_group_gather_path)."""
        P = C.c_void_p * self.nshards
        pq = P(*[C.c_void_p(int(p)) if p else None for p in d_q])
        pqd = P(*[C.c_void_p(int(p)) if p else None for p in d_qdot])
        _abi.check(self._L.rmx_group_gather_device(self._g, pq, pqd, int(root)), "rmx_group_gather_device")
        return self._L.rmx_group_gather_path(self._g).decode()

    def gather(self, root=GATHER_ALL):
        """The same gather into destinations the group owns on the receiving shards' devices (rmx_group_gather)."""
        _abi.check(self._L.rmx_group_gather(self._g, int(root)), "rmx_group_gather")
        return self._L.rmx_group_gather_path(self._g).decode()

    def gathered_read(self, shard=0):
        """Host copy of shard `shard`'s gathered (q, qdot) ([batch][nr]); gathered_ptrs: its device pointers."""
        q, qd = np.empty((self.B, self.nr)), np.empty((self.B, self.nr))
        _abi.check(self._L.rmx_group_gathered_read(self._g, int(shard), _abi.dptr(q), _abi.dptr(qd)), "rmx_group_gathered_read")
        return q, qd

    def gathered_ptrs(self, shard=0):
        a, b = C.c_void_p(), C.c_void_p()
        _abi.check(self._L.rmx_group_gathered(self._g, int(shard), C.byref(a), C.byref(b)), "rmx_group_gathered")
        return a.value, b.value

    def _outputs(self, nsteps, record):
        out = {k: np.zeros(self.B, dtype=np.int32) for k in ("newton_iters", "ls_halvings", "status")}
        st = _abi.Stats(_abi.iptr(out["newton_iters"]), _abi.iptr(out["ls_halvings"]), _abi.iptr(out["status"]))
        hist = _abi.History()
        if record & _abi.REC_ENERGY:
            out["T"], out["V"] = np.empty((nsteps, self.B)), np.empty((nsteps, self.B))
            hist.T, hist.V = _abi.dptr(out["T"]), _abi.dptr(out["V"])
        if record & _abi.REC_STATE:
            out["q"], out["qdot"] = np.empty((nsteps, self.B, self.nr)), np.empty((nsteps, self.B, self.nr))
            hist.q, hist.qdot = _abi.dptr(out["q"]), _abi.dptr(out["qdot"])
        if record & _abi.REC_CHARTS and self.nsph:
            out["charts"] = np.full((nsteps, self.B, self.nsph), 7, dtype=np.int32)
            hist.charts = _abi.iptr(out["charts"])
        return out, st, hist

    def step(self, nsteps, integrator=1, h=None, record=0):
        """simLoop of the whole batch (rmx_group_step); returns counters + the recorded per-step arrays + timing."""
        if h is not None:
            self.opts.h = float(h)
        out, st, hist = self._outputs(int(nsteps), int(record))
        _abi.check(self._L.rmx_group_step(self._g, C.byref(self.opts), int(nsteps), int(integrator), C.byref(st), C.byref(hist)), "rmx_group_step")
        out.update(self.timing())
        return out

    def step_async(self, nsteps, integrator=1, h=None, record=0):
        if h is not None:
            self.opts.h = float(h)
        self._async = (int(nsteps), int(record))
        _abi.check(self._L.rmx_group_step_async(self._g, C.byref(self.opts), int(nsteps), int(integrator), int(record)), "rmx_group_step_async")

    def sync(self):
        if self._async is None:
            raise _abi.RedMaxHipError("GroupSim.sync: no step_async launch is outstanding on this group")
        nsteps, record = self._async
        self._async = None
        out, st, hist = self._outputs(nsteps, record)
        _abi.check(self._L.rmx_group_sync(self._g, C.byref(st), C.byref(hist)), "rmx_group_sync")
        out.update(self.timing())
        return out

    def energy(self):
        T, V = np.empty(self.B), np.empty(self.B)
        _abi.check(self._L.rmx_group_energy(self._g, _abi.dptr(T), _abi.dptr(V)), "rmx_group_energy")
        return T, V

    def timing(self):
        """wall_ms of the last step and, per shard, kernel ms and the start / end of its launch relative to the first shard on the
        same device (rmx_group_timing)."""
        w = C.c_double()
        k, t0, t1 = np.zeros(self.nshards), np.zeros(self.nshards), np.zeros(self.nshards)
        _abi.check(self._L.rmx_group_timing(self._g, C.cast(C.byref(w), _abi._dp), _abi.dptr(k), _abi.dptr(t0), _abi.dptr(t1)), "rmx_group_timing")
        return {"wall_ms": w.value, "kernel_ms": k, "start_ms": t0, "end_ms": t1}
