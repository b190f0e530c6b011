"""ctypes wrapper around oracle/libredmax_oracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from redmax_amd/ (the product).  See redmax_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libredmax_oracle.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class _Desc(C.Structure):
    _fields_ = [
        ("njoints", C.c_int),
        ("parent", _ip), ("type", _ip),
        ("axis", _dp), ("E0_pj", _dp), ("E0_ji", _dp), ("I_i", _dp),
        ("q", _dp), ("qdot", _dp), ("tau", _dp), ("stiffness", _dp), ("damping", _dp),
        ("qLimL", _dp), ("qLimU", _dp), ("qLimK", _dp), ("qLimD", _dp),
        ("grav", C.c_double * 3),
        ("normalize_axis", C.c_int),
    ]


class Stats(C.Structure):
    _fields_ = [("newton_iters", C.c_int), ("ls_halvings", C.c_int), ("residual_evals", C.c_int),
                ("hessian_evals", C.c_int), ("diverged", C.c_int), ("not_converged", C.c_int), ("chart_switches", C.c_int),
                ("worst_exit_g", C.c_double)]


class TaskPointPos(C.Structure):
    _fields_ = [("body", C.c_int), ("xlocal", C.c_double * 3), ("xtarget", C.c_double * 3), ("t", C.c_double),
                ("pscale", C.c_double), ("wreg", C.c_double), ("wpos", C.c_double)]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("redmax_oracle.c", "redmax_tensorfree.c", "redmax_oracle.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(_Desc)]
        L.orc_destroy.argtypes = [C.c_void_p]
        for f in ("orc_nr", "orc_nm"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        L.orc_idxR.argtypes = [C.c_void_p, _ip]
        L.orc_reset.argtypes = [C.c_void_p]
        L.orc_set_idxR.argtypes = [C.c_void_p, _ip]
        L.orc_set_spherical.argtypes = [C.c_void_p, C.c_int, _ip]
        L.orc_get_charts.argtypes = [C.c_void_p, _ip]
        L.orc_set_charts.argtypes = [C.c_void_p, _ip]
        L.orc_euler.argtypes = [C.c_int, _dp, _dp, _dp]
        L.orc_euler.restype = C.c_double
        L.orc_euler_inv.argtypes = [C.c_int, _dp, _dp]
        L.orc_get_state.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_set_state.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_set_qrest.argtypes = [C.c_void_p, _dp]
        L.orc_energy.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_jacobian.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.orc_compute_values.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        L.orc_eval_residual.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double, _dp, _dp]
        L.orc_step_bdf1.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(Stats), _dp, _dp]
        L.orc_step_bdf2.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.POINTER(Stats), _dp, _dp]
        L.orc_step_euler_simple.argtypes = [C.c_void_p, C.c_double, C.c_int, _dp, _dp]
        L.orc_batch_step_bdf1.argtypes = [C.POINTER(_Desc), C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_int]
        L.orc_batch_step_bdf1.restype = C.c_long
        L.orc_batch_step_bdf1_ex.argtypes = [C.POINTER(_Desc), C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_int, _ip, _ip, _ip]
        L.orc_batch_step_bdf1_ex.restype = C.c_long
        L.orc_batch_step_bdf1_ex2.argtypes = [C.POINTER(_Desc), C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_int, _ip, _ip, _ip, _ip, _dp]
        L.orc_batch_step_bdf1_ex2.restype = C.c_long
        L.otf_nr.argtypes = [C.POINTER(_Desc)]
        L.otf_eval.argtypes = [C.POINTER(_Desc), _dp, _dp, _dp, C.c_double, _dp, _dp]
        L.otf_eval_lo.argtypes = [C.POINTER(_Desc), _dp, _dp, _dp, _dp, C.c_double, _dp, _dp]
        L.otf_batch_step_bdf1.argtypes = [C.POINTER(_Desc), C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double,
                                          C.c_int, C.c_int, C.c_int, _ip, _ip, _ip]
        L.otf_batch_step_bdf1.restype = C.c_long
        L.orc_set_newton.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_set_ls_fail_limit.argtypes = [C.c_int]
        L.orc_set_trace.argtypes = [_dp, C.c_int]
        L.orc_trace_count.restype = C.c_int
        L.orc_set_ground_contact.argtypes = [C.c_void_p, _ip, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_set_ground_contact_body.argtypes = [C.c_void_p, _ip, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int]
        L.orc_add_ground_contact.argtypes = [C.c_void_p, C.c_int, _dp, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_add_ground_contact.restype = C.c_int
        L.orc_adjoint_bdf1.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(TaskPointPos), _dp, _dp, C.POINTER(Stats)]
        L.orc_adjoint_bdf1.restype = C.c_double
        L.orc_adjoint_bdf2.argtypes = L.orc_adjoint_bdf1.argtypes
        L.orc_adjoint_bdf2.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def make_desc(d, normalize_axis=1):
    """d: the dict produced by redmax_amd.redmax.Scene.desc(). Returns (struct, keepalive)."""
    keep = {}

    def f64(k):
        keep[k] = np.ascontiguousarray(d[k], dtype=np.float64)
        return keep[k].ctypes.data_as(_dp)

    def i32(k):
        keep[k] = np.ascontiguousarray(d[k], dtype=np.int32)
        return keep[k].ctypes.data_as(_ip)

    s = _Desc()
    s.njoints = int(d["njoints"])
    s.parent, s.type = i32("parent"), i32("type")
    for k in ("axis", "E0_pj", "E0_ji", "I_i", "q", "qdot", "tau", "stiffness", "damping", "qLimL", "qLimU", "qLimK", "qLimD"):
        setattr(s, k, f64(k))
    for i in range(3):
        s.grav[i] = float(d["grav"][i])
    s.normalize_axis = normalize_axis
    return s, keep


# sub-joints (type, axis) of the reference's multi-DOF joints, parent first; type codes as in Scene.desc()
_E = np.eye(3)


def _composite_subjoints(jtype, plane):
    if jtype == 3:      # JointPlanar.m:24-31   Q(1:3,4) = B*q, S = [0; B]
        return [(2, plane[:3]), (2, plane[3:])]
    if jtype == 4:      # JointTranslational.m:22-26   Q(1:3,4) = q
        return [(2, _E[0]), (2, _E[1]), (2, _E[2])]
    if jtype == 5:      # JointUniversal.m:71-74   R = X1(q1)*Y2(q2)
        return [(1, _E[0]), (1, _E[1])]
    if jtype == 6:      # JointFree2D.m:20-33   Q = [Rz(q3) [q1;q2;0]]
        return [(2, _E[0]), (2, _E[1]), (1, _E[2])]
    if jtype == 7:      # JointSpherical.m:33, 247-262   chart XYZ at construction: R = X1*Y2*Z3 (chart switches: orc_set_spherical)
        return [(1, _E[0]), (1, _E[1]), (1, _E[2])]
    if jtype == 8:      # JointFree3D.m:16-34   JointTranslational then JointSpherical
        return [(2, _E[0]), (2, _E[1]), (2, _E[2]), (1, _E[0]), (1, _E[1]), (1, _E[2])]
    raise ValueError("joint type %d" % jtype)


def lower_composite(d):
    """Restates the reference's multi-DOF joints whose Q(q) is a product of one-parameter motions (JointPlanar,
    JointTranslational, JointUniversal, JointFree2D) as chains of prismatic/revolute joints with massless intermediate links:
    same world transforms, same generalised coordinates and velocities, hence the same M, f, K, D.  The chain's DOFs keep the
    reference's reduced indices idxR = nr + (1:ndof) (Joint.m:152) through orc_set_idxR.  Pinned by Hexpected of scenes
    4, 5, 6, 8 and 11 (scenesRedMax.m:147-148, 166-167, 189-190, 230-231, 292-293).  JointSpherical / JointFree3D (Euler charts,
    types 7 / 8) lower the same way; "sph_first" lists the first revolute node of every spherical group for orc_set_spherical,
    which adds the chart switching of reparam_ (pinned by scenes 7 and 9, :206-207, 250-251)."""
    typ = np.asarray(d["type"])
    n = int(d["njoints"])
    if not np.any(typ > 2):
        if d.get("extra_forces"):                    # (listing entry = node)
            d = dict(d, extra_forces=[dict(f, node=int(f["body"])) for f in d["extra_forces"]])
        return d
    ndof = [0 if t == 0 else 1 if t <= 2 else len(_composite_subjoints(t, np.zeros(6))) for t in typ]
    base = [0] * n
    nr = 0
    for L in range(n - 1, -1, -1):
        base[L] = nr
        nr += ndof[L]
    qR = np.asarray(d["qR"], float)
    qdR = np.asarray(d["qdotR"], float)
    qrR = np.asarray(d.get("qRestR", qR), float)
    out = {k: [] for k in ("parent", "type", "axis", "E0_pj", "E0_ji", "I_i", "q", "qdot", "qRest", "tau", "stiffness", "damping",
                           "qLimL", "qLimU", "qLimK", "qLimD", "idx", "contact", "sides")}
    eye16 = np.eye(4).reshape(16)
    last = [0] * n
    sph_first = []
    has_contact = d.get("contact") is not None
    for L in range(n):
        if typ[L] <= 2:
            subs = [(int(typ[L]), np.asarray(d["axis"][L], float))]
        else:
            subs = _composite_subjoints(int(typ[L]), np.asarray(d["plane"][L], float))
        if typ[L] in (7, 8):
            sph_first.append(len(out["type"]) + (3 if typ[L] == 8 else 0))
        for k, (t, ax) in enumerate(subs):
            final = k == len(subs) - 1
            out["parent"].append((last[d["parent"][L]] if d["parent"][L] >= 0 else -1) if k == 0 else len(out["type"]) - 1)
            out["type"].append(t)
            out["axis"].append(ax)
            out["E0_pj"].append(np.asarray(d["E0_pj"][L]) if k == 0 else eye16)
            out["E0_ji"].append(np.asarray(d["E0_ji"][L]) if final else eye16)
            out["I_i"].append(np.asarray(d["I_i"][L]) if final else np.zeros(6))
            i = base[L] + k
            out["idx"].append(i if t != 0 else -1)
            out["q"].append(qR[i] if t != 0 else 0.0)
            out["qdot"].append(qdR[i] if t != 0 else 0.0)
            out["qRest"].append(qrR[i] if t != 0 else 0.0)
            for key in ("tau", "stiffness", "damping", "qLimL", "qLimU", "qLimK", "qLimD"):
                out[key].append(d[key][L])
            out["contact"].append(int(d["contact"][L]) if (has_contact and final) else 0)
            out["sides"].append(np.asarray(d["sides"][L]) if (has_contact and final) else np.zeros(3))
            out.setdefault("gL", []).append(L)        # listing entry of this node (per-body ground frames follow the body)
        last[L] = len(out["type"]) - 1
    low = {"njoints": len(out["type"]), "grav": d["grav"], "idx": np.array(out["idx"], dtype=np.int32), "last_of_listing": last,
           "sph_first": np.array(sph_first, dtype=np.int32)}
    for k in ("parent", "type"):
        low[k] = np.array(out[k], dtype=np.int32)
    for k in ("axis", "E0_pj", "E0_ji", "I_i"):
        low[k] = np.ascontiguousarray(np.stack(out[k]), dtype=np.float64)
    for k in ("q", "qdot", "qRest", "tau", "stiffness", "damping", "qLimL", "qLimU", "qLimK", "qLimD"):
        low[k] = np.array(out[k], dtype=np.float64)
    if has_contact:
        low["contact"] = np.array(out["contact"], dtype=np.int32)
        low["sides"] = np.ascontiguousarray(np.stack(out["sides"]), dtype=np.float64)
        low["ground"] = d["ground"]
        if d.get("ground_body") is not None:      # one frame / set of constants per ForceGroundCuboid object, listing order -> node order
            gb = d["ground_body"]
            low["ground_body"] = {k: np.ascontiguousarray(np.asarray(gb[k], dtype=np.float64)[out["gL"]]) for k in ("E", "kn", "kt", "mu", "kd")}
        if d.get("extra_forces"):                    # further force objects: the body of listing entry L is carried by node last[L]
            low["extra_forces"] = [dict(f, node=int(last[int(f["body"])])) for f in d["extra_forces"]]
    return low


class Oracle:
    """One reference scene on the CPU oracle."""

    def __init__(self, desc_dict, normalize_axis=1):
        self._L = lib()
        desc_dict = lower_composite(desc_dict)
        self._d, self._keep = make_desc(desc_dict, normalize_axis)
        self._h = C.c_void_p(self._L.orc_create(C.byref(self._d)))
        self.nr = self._L.orc_nr(self._h)
        self.nm = self._L.orc_nm(self._h)
        if "idx" in desc_dict:
            self._idx = np.ascontiguousarray(desc_dict["idx"], dtype=np.int32)
            if self._L.orc_set_idxR(self._h, self._idx.ctypes.data_as(_ip)) != 0:
                raise ValueError("idx is not a permutation of the reduced DOFs")
        self.nsph = 0
        if len(desc_dict.get("sph_first", ())):
            self._sph = np.ascontiguousarray(desc_dict["sph_first"], dtype=np.int32)
            self.nsph = len(self._sph)
            if self._L.orc_set_spherical(self._h, self.nsph, self._sph.ctypes.data_as(_ip)) != 0:
                raise ValueError("bad spherical groups")
        if "qRest" in desc_dict:
            self.set_qrest_joint_order(desc_dict["qRest"])
        if desc_dict.get("contact") is not None and np.any(desc_dict["contact"]):
            g = desc_dict["ground"]
            self._cflags = np.ascontiguousarray(desc_dict["contact"], dtype=np.int32)
            self._csides = np.ascontiguousarray(desc_dict["sides"], dtype=np.float64)
            self._cE = np.ascontiguousarray(np.asarray(g["E"], dtype=np.float64).reshape(4, 4).T.reshape(16))
            gb = desc_dict.get("ground_body")
            if gb is None:
                self._L.orc_set_ground_contact(self._h, self._cflags.ctypes.data_as(_ip), _p(self._csides), _p(self._cE),
                                               float(g["kn"]), float(g["kt"]), float(g["mu"]), float(g["kd"]))
            else:                                     # [n][4][4] row-major -> [n][16] column-major
                self._cEb = np.ascontiguousarray(np.asarray(gb["E"], dtype=np.float64).reshape(-1, 4, 4).transpose(0, 2, 1).reshape(-1, 16))
                self._ck = [np.ascontiguousarray(gb[k], dtype=np.float64) for k in ("kn", "kt", "mu", "kd")]
                self._L.orc_set_ground_contact_body(self._h, self._cflags.ctypes.data_as(_ip), _p(self._csides), _p(self._cEb), 1,
                                                    _p(self._ck[0]), _p(self._ck[1]), _p(self._ck[2]), _p(self._ck[3]), 1)
            # the reference keeps its force objects in a list (Force.m:26-56): any number of ForceGroundCuboid per body
            for f in desc_dict.get("extra_forces") or ():
                E = np.ascontiguousarray(np.asarray(f["E"], dtype=np.float64).reshape(4, 4).T.reshape(16))
                if self._L.orc_add_ground_contact(self._h, int(f["node"]), _p(E), float(f["kn"]), float(f["kt"]), float(f["mu"]), float(f["kd"])) != 0:
                    raise ValueError("extra force on a body that carries no ForceGroundCuboid")

    def __del__(self):
        try:
            if self._h:
                self._L.orc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def idxR(self):
        a = np.zeros(self._d.njoints, dtype=np.int32)
        self._L.orc_idxR(self._h, a.ctypes.data_as(_ip))
        return a

    def set_qrest_joint_order(self, qrest_per_joint):
        idx = self.idxR()
        r = np.zeros(max(self.nr, 1))
        for i, k in enumerate(idx):
            if k >= 0:
                r[k] = qrest_per_joint[i]
        self._L.orc_set_qrest(self._h, _p(r))

    def reset(self):
        self._L.orc_reset(self._h)

    def charts(self):
        """Current Euler chart (reference numbering 1..12) of every JointSpherical / JointFree3D, in listing order."""
        c = np.zeros(max(self.nsph, 1), dtype=np.int32)
        self._L.orc_get_charts(self._h, c.ctypes.data_as(_ip))
        return c[:self.nsph]

    def set_charts(self, charts):
        c = np.ascontiguousarray(charts, dtype=np.int32)
        if len(c) != self.nsph or self._L.orc_set_charts(self._h, c.ctypes.data_as(_ip)) != 0:
            raise ValueError("charts must be %d values in 1..12" % self.nsph)

    def get_state(self):
        q = np.zeros(self.nr)
        qd = np.zeros(self.nr)
        self._L.orc_get_state(self._h, _p(q), _p(qd))
        return q, qd

    def set_state(self, q, qdot):
        q = np.ascontiguousarray(q, dtype=np.float64)
        qdot = np.ascontiguousarray(qdot, dtype=np.float64)
        self._L.orc_set_state(self._h, _p(q), _p(qdot))

    def energy(self):
        T = C.c_double()
        V = C.c_double()
        self._L.orc_energy(self._h, C.byref(T), C.byref(V))
        return T.value, V.value

    def jacobian(self, deriv=False):
        nm, nr = self.nm, self.nr
        J = np.zeros(nm * nr)
        Jd = np.zeros(nm * nr)
        if deriv:
            dJ = np.zeros(nm * nr * nr)
            dJd = np.zeros(nm * nr * nr)
            self._L.orc_jacobian(self._h, _p(J), _p(Jd), _p(dJ), _p(dJd))
            return (J.reshape((nm, nr), order="F"), Jd.reshape((nm, nr), order="F"),
                    dJ.reshape((nm, nr, nr), order="F"), dJd.reshape((nm, nr, nr), order="F"))
        self._L.orc_jacobian(self._h, _p(J), _p(Jd), None, None)
        return J.reshape((nm, nr), order="F"), Jd.reshape((nm, nr), order="F")

    def compute_values(self, deriv=True):
        nr = self.nr
        M = np.zeros(nr * nr)
        f = np.zeros(nr)
        if not deriv:
            self._L.orc_compute_values(self._h, _p(M), _p(f), None, None, None)
            return M.reshape((nr, nr), order="F"), f
        dM = np.zeros(nr * nr * nr)
        K = np.zeros(nr * nr)
        D = np.zeros(nr * nr)
        self._L.orc_compute_values(self._h, _p(M), _p(f), _p(dM), _p(K), _p(D))
        return (M.reshape((nr, nr), order="F"), f, dM.reshape((nr, nr, nr), order="F"),
                K.reshape((nr, nr), order="F"), D.reshape((nr, nr), order="F"))

    def eval_residual(self, q, qA, qB, eta, want_H=True):
        nr = self.nr
        q, qA, qB = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, qA, qB))
        g = np.zeros(nr)
        if want_H:
            H = np.zeros(nr * nr)
            self._L.orc_eval_residual(self._h, _p(q), _p(qA), _p(qB), float(eta), _p(g), _p(H))
            return g, H.reshape((nr, nr), order="F")
        self._L.orc_eval_residual(self._h, _p(q), _p(qA), _p(qB), float(eta), _p(g), None)
        return g

    def eval_bdf1(self, q1, q0, qdot0, h, want_H=True):
        """evalBDF1 (driverRedMaxBDF1.m:160-187)."""
        q0 = np.asarray(q0, dtype=np.float64)
        return self.eval_residual(q1, q0, q0 + h * np.asarray(qdot0, dtype=np.float64), h, want_H)

    def step_bdf1(self, h, nsteps, history=False):
        st = Stats()
        if history:
            T = np.zeros(nsteps)
            V = np.zeros(nsteps)
            self._L.orc_step_bdf1(self._h, float(h), int(nsteps), C.byref(st), _p(T), _p(V))
            return st, T, V
        self._L.orc_step_bdf1(self._h, float(h), int(nsteps), C.byref(st), None, None)
        return st

    def step_bdf2(self, h, nsteps, step0=0, history=False):
        st = Stats()
        if history:
            T = np.zeros(nsteps)
            V = np.zeros(nsteps)
            self._L.orc_step_bdf2(self._h, float(h), int(step0), int(nsteps), C.byref(st), _p(T), _p(V))
            return st, T, V
        self._L.orc_step_bdf2(self._h, float(h), int(step0), int(nsteps), C.byref(st), None, None)
        return st

    def adjoint_bdf2(self, h, nsteps, task, p):
        """taskObjective of driverRedMaxAdjointBDF2.m:38-62 (TaskBDF2PointPos, scene 101).  Returns (P, dPdp, stats)."""
        return self.adjoint_bdf1(h, nsteps, task, p, _fn="orc_adjoint_bdf2")

    def adjoint_bdf1(self, h, nsteps, task, p, _fn="orc_adjoint_bdf1"):
        """taskObjective (driverRedMaxAdjointBDF1.m:39-62). task: dict(body, xlocal, xtarget, t, pscale, wreg, wpos).
        Returns (P, dPdp, stats)."""
        tk = TaskPointPos()
        tk.body = int(task["body"])
        for i in range(3):
            tk.xlocal[i] = float(task["xlocal"][i])
            tk.xtarget[i] = float(task["xtarget"][i])
        tk.t, tk.pscale, tk.wreg, tk.wpos = (float(task[k]) for k in ("t", "pscale", "wreg", "wpos"))
        p = np.ascontiguousarray(p, dtype=np.float64)
        dPdp = np.zeros(self.nr)
        st = Stats()
        P = getattr(self._L, _fn)(self._h, float(h), int(nsteps), C.byref(tk), _p(p), _p(dPdp), C.byref(st))
        return float(P), dPdp, st

    def step_euler_simple(self, h, nsteps):
        T = np.zeros(nsteps)
        V = np.zeros(nsteps)
        self._L.orc_step_euler_simple(self._h, float(h), int(nsteps), _p(T), _p(V))
        return T, V


def set_newton(tol=1e-9, dxMax=1e3, iterMaxPerDof=10, iterLsMax=20):
    """Newton constants for every subsequent step call (defaults = driverRedMaxBDF1.m:95-98)."""
    lib().orc_set_newton(float(tol), float(dxMax), int(iterMaxPerDof), int(iterLsMax))


class newton_trace:
    """with newton_trace() as t: o.step_bdf1(h, 1) -> t.rows = [[|g| at the start of the iteration, |g| after its line search, trials], ...]
    of every Newton iteration run inside the block (diagnostic of the oracle's newton(); single-threaded calls only)."""

    def __init__(self, cap=4096):
        self._buf = np.zeros((cap, 3))
        self.rows = None

    def __enter__(self):
        lib().orc_set_trace(_p(self._buf), self._buf.shape[0])
        return self

    def __exit__(self, *exc):
        n = lib().orc_trace_count()
        lib().orc_set_trace(None, 0)
        self.rows = self._buf[:n].copy()
        return False


def set_ls_fail_limit(n=0):
    """rmx_opts.ls_fail_limit restated (0 = off = the reference's newton)."""
    lib().orc_set_ls_fail_limit(int(n))


def batch_step_bdf1(desc_dict, q, qdot, h, nsteps, nthreads=0, counters=False):
    """B independent rollouts on the host cores (OpenMP over trajectories): the cpu_baseline leg.
    q, qdot: [B][nr] arrays, updated in place.  Returns total Newton iterations; with counters=True a dict of per-rollout
    arrays (newton_iters, ls_halvings, bad = steps that diverged or did not converge) instead."""
    L = lib()
    d, keep = make_desc(desc_dict)
    assert q.flags.c_contiguous and qdot.flags.c_contiguous and q.dtype == np.float64
    # qRest: the batch helper takes it from the descriptor's q (model constant)
    if not counters:
        return int(L.orc_batch_step_bdf1(C.byref(d), int(q.shape[0]), _p(q), _p(qdot), float(h), int(nsteps), int(nthreads)))
    out = {k: np.zeros(q.shape[0], dtype=np.int32) for k in ("newton_iters", "ls_halvings", "bad", "diverged")}
    out["worst_exit_g"] = np.zeros(q.shape[0])
    L.orc_batch_step_bdf1_ex2(C.byref(d), int(q.shape[0]), _p(q), _p(qdot), float(h), int(nsteps), int(nthreads),
                              *[out[k].ctypes.data_as(_ip) for k in ("newton_iters", "ls_halvings", "bad", "diverged")], _p(out["worst_exit_g"]))
    return out


# ---- redmax_tensorfree.c: "Baseline B", the tensor-free CPU implementation (same algorithm as the HIP kernels) ----
def tensorfree_eval(desc_dict, q, qA, qB, eta, want_H=True):
    """g (and H [row][col]) of the generic implicit residual from the tensor-free CPU code."""
    L = lib()
    d, keep = make_desc(desc_dict)
    nr = L.otf_nr(C.byref(d))
    q, qA, qB = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, qA, qB))
    g = np.zeros(nr)
    H = np.zeros(nr * nr) if want_H else None
    L.otf_eval(C.byref(d), _p(q), _p(qA), _p(qB), float(eta), _p(g), _p(H))
    return (g, H.reshape(nr, nr).T.copy()) if want_H else g


def tensorfree_eval_lo(desc_dict, q, qlo, qA, qB, eta, want_H=True):
    """tensorfree_eval at the compensated iterate q + qlo (qlo enters v = q - qB and qdot = (q - qA)/eta only)."""
    L = lib()
    d, keep = make_desc(desc_dict)
    nr = L.otf_nr(C.byref(d))
    q, qlo, qA, qB = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, qlo, qA, qB))
    g = np.zeros(nr)
    H = np.zeros(nr * nr) if want_H else None
    L.otf_eval_lo(C.byref(d), _p(q), _p(qlo), _p(qA), _p(qB), float(eta), _p(g), _p(H))
    return (g, H.reshape(nr, nr).T.copy()) if want_H else g


def tensorfree_batch_step_bdf1(desc_dict, q, qdot, h, nsteps, nthreads=0, tol=1e-9, dxMax=1e3, iterMaxPerDof=10, iterLsMax=20,
                               compensated=True):
    """simLoop for B rollouts (q, qdot [B][nr], in place), OpenMP over rollouts.  Returns per-rollout counters.
    compensated: the Newton iterate as the unevaluated sum x + xlo, as the HIP kernels carry it by default (rmx_opts.compensated)."""
    L = lib()
    d, keep = make_desc(desc_dict)
    assert q.flags.c_contiguous and qdot.flags.c_contiguous and q.dtype == np.float64
    out = {k: np.zeros(q.shape[0], dtype=np.int32) for k in ("newton_iters", "ls_halvings", "status")}
    L.otf_batch_step_bdf1(C.byref(d), int(q.shape[0]), _p(q), _p(qdot), float(h), int(nsteps), int(nthreads), float(tol), float(dxMax),
                          int(iterMaxPerDof), int(iterLsMax), int(bool(compensated)), *[out[k].ctypes.data_as(_ip) for k in ("newton_iters", "ls_halvings", "status")])
    return out


def euler(chart, q):
    """redmax.JointSpherical.getEuler(chart, q, 0): (R, T, detT)."""
    R = np.zeros(9)
    T = np.zeros(9)
    d = lib().orc_euler(int(chart), _p(np.ascontiguousarray(q, dtype=np.float64)), _p(R), _p(T))
    return R.reshape(3, 3), T.reshape(3, 3), d


def euler_inv(chart, R):
    """redmax.JointSpherical.getEulerInv(chart, R)."""
    q = np.zeros(3)
    lib().orc_euler_inv(int(chart), _p(np.ascontiguousarray(R, dtype=np.float64).reshape(9)), _p(q))
    return q
