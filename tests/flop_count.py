"""Flops the ALGORITHM needs (SURVEY.md 8(d) counting: 1 per add / multiply, FMA = 2), from the scalar CPU twin of the kernels.

    python tests/flop_count.py [out.json]          (default: profiles/algorithm_flops.json)

oracle/redmax_tensorfree.c is the world-frame O(n^2) algorithm the HIP kernels execute, in scalar C (test infrastructure:
checked against the literal oracle in tests/test_oracle_tensorfree.py).  Compiled as C++ with a counting `double`
(tests/flopcount/counted.h) it yields, per workload, the flops of one residual-only evaluation, of the Hessian part of one (g, H)
evaluation, of one solve and of Newton's own bookkeeping.  bench.py multiplies them with the evaluation / iteration counts MEASURED in
the timed launch: `roofline.useful_frac` = those flops / kernel time / fp64 peak - the lane-honest companion of `frac`, which
counts every wave-wide instruction with all 64 lanes (idle lanes of a 32-node tree, the four-fold replicated pivot columns of
the DPP solve and masked MFMA tiles included).  Divisions, square roots and sin / cos are listed but not counted as flops.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build():
    so = os.path.join(ROOT, "build", "flopcount", "libtwin_counted.so")
    src = os.path.join(ROOT, "tests", "flopcount", "twin_counted.cpp")
    deps = [src, os.path.join(ROOT, "tests", "flopcount", "counted.h"), os.path.join(ROOT, "oracle", "redmax_tensorfree.c")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fpermissive", "-w", "-I", os.path.join(ROOT, "oracle"),
                               "-o", so, src])
    return C.CDLL(so)


def counts(L, scene, q, qd, h):
    from oracle import oracle as orc
    desc, keep = orc.make_desc(scene.desc())
    out = np.zeros((4, 4), dtype=np.int64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    qd = np.ascontiguousarray(qd, dtype=np.float64)
    L.fc_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    L.fc_counts(C.byref(desc), q.ctypes.data, qd.ctypes.data, float(h), out.ctypes.data)
    del keep
    g, gh, lu, nw = out
    flops = lambda c: int(c[0] + c[1])
    return {
        "front": {"flops": flops(g), "add": int(g[0]), "mul": int(g[1]), "div": int(g[2]), "sincos_sqrt": int(g[3])},
        "hessian_beyond_front": {"flops": flops(gh) - flops(g), "add": int(gh[0] - g[0]), "mul": int(gh[1] - g[1]), "div": int(gh[2] - g[2])},
        "solve": {"flops": flops(lu), "add": int(lu[0]), "mul": int(lu[1]), "div": int(lu[2])},
        "newton_bookkeeping": {"flops": flops(nw)},
        # what bench.py uses: per front evaluation / per Newton iteration beyond its front (Hessian + solve + bookkeeping)
        "per_front": flops(g), "per_newton": flops(gh) - flops(g) + flops(lu) + flops(nw),
    }


def main():
    from redmax_amd import sceneChain, sceneTree, syntheticStates
    from redmax_amd.scenes import sceneAdjointChain
    L = build()
    out = {"rule": "1 per fp64 add / subtract / multiply (FMA = 2), divisions / sqrt / sin / cos listed apart; scalar algorithm = "
                   "oracle/redmax_tensorfree.c (world-frame O(n^2) recursion, dense partial-pivot LU), counted by tests/flop_count.py",
           "workloads": {}}
    for name, scene in (("chain", sceneChain(32)), ("tree64", sceneTree(64)), ("chain72", sceneChain(72)), ("chain128", sceneChain(128)),
                        ("chain256", sceneChain(256)), ("adjoint16", sceneAdjointChain(16))):
        scene.init()
        q, qd = syntheticStates(scene.nr, 1, first=3)
        out["workloads"][name] = dict(counts(L, scene, q[0], qd[0], 1e-2), n=int(scene.nr))
        print(name, json.dumps(out["workloads"][name]))
    # the ground workload's evaluation without its contact terms is the 32-chain's: the twin has no ForceGroundCuboid, so the figure is a
    # LOWER bound of the useful work there (bench.py says so)
    out["workloads"]["ground"] = dict(out["workloads"]["chain"], note="contact terms not in the scalar twin: lower bound")
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "algorithm_flops.json")
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
