"""ctypes binding of the C ABI in include/redmax_hip.h (libredmax_hip.so, built in-tree).

This is the only place the product touches native code.  There is deliberately no fallback:
if the shared library is missing or no HIP device is visible, calls raise RedMaxHipError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libredmax_hip.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# every symbol include/redmax_hip.h and include/redmax_hip_profile.h (the measurement hooks) declare (tests/test_abi_symbols.py checks the .so exports them all)
SYMBOLS = (
    "rmx_last_error", "rmx_version", "rmx_device_count", "rmx_opts_default",
    "rmx_model_create", "rmx_model_destroy", "rmx_model_nr", "rmx_model_nm", "rmx_model_idxR",
    "rmx_model_set_ground_contact", "rmx_model_nsph", "rmx_get_charts", "rmx_set_charts",
    "rmx_batch_create", "rmx_batch_destroy", "rmx_batch_size",
    "rmx_set_state", "rmx_get_state", "rmx_set_state_device", "rmx_get_state_device",
    "rmx_eval", "rmx_eval_mfd", "rmx_compute_values", "rmx_step_bdf1", "rmx_step_bdf2", "rmx_step_history", "rmx_step_euler", "rmx_adjoint_bdf1", "rmx_adjoint_bdf2", "rmx_adjoint_bdf1_device", "rmx_adjoint_bdf2_device", "rmx_energy",
    "rmx_last_step_ms", "rmx_last_step_kernel", "rmx_batch_stream", "rmx_step_bdf1_async", "rmx_step_bdf2_async", "rmx_step_history_async", "rmx_sync",
    "rmx_history_read", "rmx_stats_reset", "rmx_stats_read", "rmx_profile_phases", "rmx_step_ticks",
    "rmx_group_create", "rmx_group_destroy", "rmx_group_batch_size", "rmx_group_nshards", "rmx_group_shard", "rmx_group_shard_batch",
    "rmx_group_shard_model", "rmx_group_set_state", "rmx_group_get_state", "rmx_group_step", "rmx_group_step_async", "rmx_group_sync",
    "rmx_group_energy", "rmx_group_timing", "rmx_group_gather_device", "rmx_group_gather_path", "rmx_group_gather", "rmx_group_gathered",
    "rmx_group_gathered_read",
)
REC_ENERGY, REC_STATE, REC_CHARTS = 1, 2, 4      # RMX_REC_*


class RedMaxHipError(RuntimeError):
    pass


class ModelDesc(C.Structure):
    _fields_ = [
        ("njoints", C.c_int),
        ("parent", _ip), ("type", _ip),
        ("axis", _dp), ("E0_pj", _dp), ("E0_ji", _dp), ("I_i", _dp),
        ("qRest", _dp), ("tau", _dp), ("stiffness", _dp), ("damping", _dp),
        ("qLimL", _dp), ("qLimU", _dp), ("qLimK", _dp), ("qLimD", _dp),
        ("grav", C.c_double * 3),
        ("plane", _dp), ("qRestR", _dp),
    ]


class Opts(C.Structure):
    _fields_ = [("h", C.c_double), ("tol", C.c_double), ("dxMax", C.c_double),
                ("iterMaxPerDof", C.c_int), ("iterLsMax", C.c_int), ("lu_mode", C.c_int), ("compensated", C.c_int), ("ls_fail_limit", C.c_int)]


class TaskPointPos(C.Structure):
    _fields_ = [("body", C.c_int), ("xlocal", C.c_double * 3), ("xtarget", C.c_double * 3), ("step", C.c_int),
                ("pscale", C.c_double), ("wreg", C.c_double), ("wpos", C.c_double)]


class GroundContact(C.Structure):
    _fields_ = [("flags", _ip), ("sides", _dp), ("E", C.c_double * 16), ("kn", C.c_double), ("kt", C.c_double),
                ("mu", C.c_double), ("kd", C.c_double), ("E_body", _dp), ("kn_body", _dp), ("kt_body", _dp), ("mu_body", _dp),
                ("kd_body", _dp)]


class History(C.Structure):
    _fields_ = [("T", _dp), ("V", _dp), ("q", _dp), ("qdot", _dp), ("charts", _ip)]


class Stats(C.Structure):
    _fields_ = [("newton_iters", _ip), ("ls_halvings", _ip), ("status", _ip)]


_lib = None


def lib():
    """Load libredmax_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RedMaxHipError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rmx_last_error.restype = C.c_char_p
    L.rmx_version.restype = C.c_int
    L.rmx_device_count.restype = C.c_int
    L.rmx_opts_default.argtypes = [C.POINTER(Opts)]
    L.rmx_model_create.argtypes = [C.POINTER(ModelDesc), C.c_int, C.POINTER(vp)]
    L.rmx_model_destroy.argtypes = [vp]
    L.rmx_model_nr.argtypes = [vp]
    L.rmx_model_nm.argtypes = [vp]
    L.rmx_model_idxR.argtypes = [vp, _ip]
    L.rmx_model_set_ground_contact.argtypes = [vp, C.POINTER(GroundContact)]
    L.rmx_model_nsph.argtypes = [vp]
    L.rmx_get_charts.argtypes = [vp, _ip]
    L.rmx_set_charts.argtypes = [vp, _ip]
    L.rmx_batch_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.rmx_batch_destroy.argtypes = [vp]
    L.rmx_batch_size.argtypes = [vp]
    L.rmx_set_state.argtypes = [vp, _dp, _dp]
    L.rmx_get_state.argtypes = [vp, _dp, _dp]
    L.rmx_set_state_device.argtypes = [vp, vp, vp]
    L.rmx_get_state_device.argtypes = [vp, vp, vp]
    L.rmx_eval.argtypes = [vp, _dp, _dp, _dp, C.c_double, _dp, _dp]
    L.rmx_eval_mfd.argtypes = [vp, _dp, _dp, _dp, _dp, _dp]
    L.rmx_compute_values.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.rmx_step_bdf1.argtypes = [vp, C.POINTER(Opts), C.c_int, C.POINTER(Stats), _dp, _dp]
    L.rmx_step_bdf2.argtypes = [vp, C.POINTER(Opts), C.c_int, C.POINTER(Stats), _dp, _dp]
    L.rmx_step_history.argtypes = [vp, C.POINTER(Opts), C.c_int, C.c_int, C.POINTER(Stats), C.POINTER(History)]
    L.rmx_step_euler.argtypes = [vp, C.c_double, C.c_int, _dp, _dp]
    L.rmx_adjoint_bdf1.argtypes = [vp, C.POINTER(Opts), C.c_int, C.POINTER(TaskPointPos), _dp, _dp, _dp, C.POINTER(Stats)]
    L.rmx_adjoint_bdf2.argtypes = L.rmx_adjoint_bdf1.argtypes
    L.rmx_adjoint_bdf1_device.argtypes = [vp, C.POINTER(Opts), C.c_int, C.POINTER(TaskPointPos), vp, vp, vp, C.POINTER(Stats)]
    L.rmx_adjoint_bdf2_device.argtypes = L.rmx_adjoint_bdf1_device.argtypes
    L.rmx_step_ticks.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.rmx_energy.argtypes = [vp, _dp, _dp]
    L.rmx_last_step_ms.argtypes = [vp]
    L.rmx_last_step_ms.restype = C.c_double
    L.rmx_last_step_kernel.argtypes = [vp]
    L.rmx_last_step_kernel.restype = C.c_char_p
    L.rmx_batch_stream.argtypes = [vp]
    L.rmx_batch_stream.restype = vp
    L.rmx_step_bdf1_async.argtypes = [vp, C.POINTER(Opts), C.c_int]
    L.rmx_sync.argtypes = [vp]
    L.rmx_step_bdf2_async.argtypes = [vp, C.POINTER(Opts), C.c_int]
    L.rmx_step_history_async.argtypes = [vp, C.POINTER(Opts), C.c_int, C.c_int, C.c_int]
    L.rmx_history_read.argtypes = [vp, C.POINTER(History)]
    L.rmx_group_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(GroundContact), C.c_int, _ip, C.c_int, C.POINTER(vp)]
    L.rmx_group_destroy.argtypes = [vp]
    L.rmx_group_batch_size.argtypes = [vp]
    L.rmx_group_nshards.argtypes = [vp]
    L.rmx_group_shard.argtypes = [vp, C.c_int, _ip, _ip, _ip]
    L.rmx_group_shard_batch.argtypes = [vp, C.c_int]
    L.rmx_group_shard_batch.restype = vp
    L.rmx_group_shard_model.argtypes = [vp, C.c_int]
    L.rmx_group_shard_model.restype = vp
    L.rmx_group_set_state.argtypes = [vp, _dp, _dp]
    L.rmx_group_get_state.argtypes = [vp, _dp, _dp]
    L.rmx_group_step.argtypes = [vp, C.POINTER(Opts), C.c_int, C.c_int, C.POINTER(Stats), C.POINTER(History)]
    L.rmx_group_step_async.argtypes = [vp, C.POINTER(Opts), C.c_int, C.c_int, C.c_int]
    L.rmx_group_sync.argtypes = [vp, C.POINTER(Stats), C.POINTER(History)]
    L.rmx_group_energy.argtypes = [vp, _dp, _dp]
    L.rmx_group_gather_device.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int]
    L.rmx_group_gather_path.argtypes = [vp]
    L.rmx_group_gather_path.restype = C.c_char_p
    L.rmx_group_gather.argtypes = [vp, C.c_int]
    L.rmx_group_gathered.argtypes = [vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.rmx_group_gathered_read.argtypes = [vp, C.c_int, _dp, _dp]
    L.rmx_group_timing.argtypes = [vp, _dp, _dp, _dp, _dp]
    L.rmx_stats_reset.argtypes = [vp]
    L.rmx_profile_phases.argtypes = [vp, C.c_int, C.c_double, _dp]
    L.rmx_stats_read.argtypes = [vp, C.POINTER(Stats)]
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().rmx_last_error()
        raise RedMaxHipError("%s failed (%d): %s" % (what or "redmax_hip call", rc, msg.decode() if msg else "?"))


def dptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def iptr(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def make_ground_contact(d, keep):
    """scene.forces (ForceGroundCuboid objects) of a Scene.desc() dict -> GroundContact, or None when the scene has none."""
    if d.get("contact") is None or not np.any(d["contact"]):
        return None
    g = d["ground"]
    gc = GroundContact()
    keep["contact"] = np.ascontiguousarray(d["contact"], dtype=np.int32)
    keep["sides"] = np.ascontiguousarray(d["sides"], dtype=np.float64)
    gc.flags, gc.sides = iptr(keep["contact"]), dptr(keep["sides"])
    gc.E[:] = list(np.asarray(g["E"], dtype=np.float64).reshape(4, 4).T.reshape(16))
    gc.kn, gc.kt, gc.mu, gc.kd = float(g["kn"]), float(g["kt"]), float(g["mu"]), float(g["kd"])
    gb = d.get("ground_body")
    if gb is not None:        # force objects with their own frames / constants, listing order; [n][4][4] row-major -> column-major
        keep["gE"] = np.ascontiguousarray(np.asarray(gb["E"], dtype=np.float64).reshape(-1, 4, 4).transpose(0, 2, 1).reshape(-1, 16))
        gc.E_body = dptr(keep["gE"])
        for k in ("kn", "kt", "mu", "kd"):
            keep["g" + k] = np.ascontiguousarray(gb[k], dtype=np.float64)
            setattr(gc, k + "_body", dptr(keep["g" + k]))
    return gc


def make_desc(d):
    """dict from redmax.Scene.desc() -> (ModelDesc, keepalive)."""
    keep = {}

    def f64(k):
        keep[k] = np.ascontiguousarray(d[k], dtype=np.float64)
        return keep[k].ctypes.data_as(_dp)

    def i32(k):
        keep[k] = np.ascontiguousarray(d[k], dtype=np.int32)
        return keep[k].ctypes.data_as(_ip)

    s = ModelDesc()
    s.njoints = int(d["njoints"])
    s.parent, s.type = i32("parent"), i32("type")
    for k in ("axis", "E0_pj", "E0_ji", "I_i", "qRest", "tau", "stiffness", "damping", "qLimL", "qLimU", "qLimK", "qLimD"):
        setattr(s, k, f64(k))
    for i in range(3):
        s.grav[i] = float(d["grav"][i])
    s.plane = f64("plane") if d.get("plane") is not None else None
    s.qRestR = f64("qRestR") if d.get("qRestR") is not None and len(d["qRestR"]) else None
    return s, keep
