"""The C-ABI library loads on a CPU-only box and exports every symbol include/redmax_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("redmax_hip.h", "redmax_hip_profile.h")      # the host-facing ABI; the measurement hooks bench.py / tools / 'ticks' bind


def _declared(headers=HEADERS):
    out = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(rmx_[A-Za-z0-9_]+)\s*\(", src))
    return sorted(out)


def test_measurement_hooks_are_not_in_the_host_facing_header():
    """Profiling / timing entries live in include/redmax_hip_profile.h: what a MEX or ctypes host binds to simulate has none of them."""
    host = _declared(("redmax_hip.h",))
    prof = _declared(("redmax_hip_profile.h",))
    assert prof == ["rmx_last_step_kernel", "rmx_last_step_ms", "rmx_profile_phases", "rmx_step_ticks"]
    assert not set(host) & set(prof)
    txt = open(os.path.join(ROOT, "include", "redmax_hip.h")).read()
    assert "s_memtime" not in txt and "RMX_STAMP" not in txt


def test_header_and_binding_agree():
    from redmax_amd import _abi
    assert sorted(_abi.SYMBOLS) == _declared()


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from redmax_amd import _abi
    L = ctypes.CDLL(_abi.LIB_PATH)
    for name in _declared():
        assert hasattr(L, name), name
    assert _abi.lib().rmx_version() == 111


def test_no_device_fails_loudly():
    """Without a HIP device model creation must fail with RMX_E_NODEVICE - there is no CPU fallback."""
    from redmax_amd import _abi, scenesRedMax, BatchSim
    if _abi.lib().rmx_device_count() > 0:
        return   # on a GPU box this property is not observable
    sc = scenesRedMax(0)
    sc.init()
    try:
        BatchSim(sc, batch=1)
    except _abi.RedMaxHipError as e:
        assert "no HIP device" in str(e)
    else:
        raise AssertionError("BatchSim must not silently succeed without a GPU")


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (tests/, smoke and bench's cpu_baseline only)."""
    pkg = os.path.join(ROOT, "redmax_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                bad = re.findall(r"^\s*(?:from|import)\s+oracle\b|libredmax_oracle|redmax_oracle\.h|orc_[a-z_]+\(", txt, flags=re.M)
                assert not bad, (os.path.join(dp, f), bad)
