"""Host-side logic: the +redmax scene mirror (index layout, ordering checks, descriptor) and se3 helpers."""
import math

import numpy as np
import pytest

from redmax_amd import se3
from redmax_amd.redmax import BodyCuboid, JointFixed, JointRevolute, Scene
from redmax_amd.scenes import sceneChain, scenesRedMax, sceneTree, syntheticStates


def test_se3_identities():
    rng = np.random.default_rng(0)
    R = se3.aaToMat(rng.normal(size=3), 0.9)
    E = se3.transform(R=R, p=rng.normal(size=3))
    assert np.allclose(se3.inv(E) @ E, np.eye(4), atol=1e-14)
    assert np.allclose(se3.Ad(se3.inv(E)) @ se3.Ad(E), np.eye(6), atol=1e-13)
    a, b = rng.normal(size=6), rng.normal(size=6)
    assert np.allclose(se3.ad(a) @ b, -se3.ad(b) @ a, atol=1e-14)           # Lie bracket antisymmetry
    assert np.allclose(se3.Ad(E) @ se3.ad(a) @ se3.Ad(se3.inv(E)), se3.ad(se3.Ad(E) @ a), atol=1e-12)


def test_aaToMat_axis_aligned_has_exact_zeros():
    """se3.m:124-156: axis-aligned rotations carry exact zeros/ones."""
    R = se3.aaToMat([0, 1, 0], 0.3)
    assert R[0, 1] == 0.0 and R[1, 0] == 0.0 and R[1, 1] == 1.0 and R[0, 0] == math.cos(0.3) and R[0, 2] == math.sin(0.3)
    Rn = se3.aaToMat([0, 0, -2.0], 0.3)
    assert np.allclose(Rn, se3.aaToMat([0, 0, 1], -0.3))


def test_inertia_cuboid():
    m = se3.inertiaCuboid([10, 1, 1], 1.0)
    assert np.allclose(m, [10 / 12 * 2, 10 / 12 * 101, 10 / 12 * 101, 10, 10, 10])


def test_scene_index_layout_leaf_to_root():
    sc = scenesRedMax(0)
    sc.init()
    assert (sc.nr, sc.nm) == (3, 30)
    assert [j.idxR for j in sc.joints] == [[2], [], [1], [], [0]]
    assert sc.joints[0].body.idxM == list(range(24, 30)) and sc.joints[-1].body.idxM == list(range(0, 6))
    d = sc.desc()
    assert list(d["parent"]) == [-1, 0, 1, 2, 3] and list(d["type"]) == [1, 0, 1, 0, 1]
    assert d["E0_pj"].shape == (5, 16) and d["E0_pj"][1][12] == 10.0       # column-major: translation x at [12]
    assert sc.nsteps == 100


def test_scene_rejects_bad_ordering():
    sc = Scene()
    b = [BodyCuboid(1.0, [1, 1, 1]) for _ in range(3)]
    j0 = JointRevolute(None, b[0], [0, 1, 0])
    j1 = JointRevolute(j0, b[1], [0, 1, 0])
    j2 = JointFixed(j0, b[2])
    for j in (j0, j1, j2):
        j.setJointTransform(np.eye(4))
    sc.bodies = [b[0], b[2], b[1]]
    sc.joints = [j0, j2, j1]            # not depth-first order of the tree (children of j0 are j1 then j2)
    with pytest.raises(ValueError):
        sc.init()


def test_qrest_is_initial_q():
    sc = scenesRedMax(14)
    sc.init()
    assert [j.qRest for j in sc.joints] == [0.0, -math.pi / 6, -math.pi / 6]


def test_synthetic_states_are_shard_invariant():
    qa, qda = syntheticStates(32, 8, first=0)
    qb, qdb = syntheticStates(32, 4, first=4)
    assert np.array_equal(qa[4:], qb) and np.array_equal(qda[4:], qdb)
    assert np.all(qa[0] == 0.1) and np.all(qda[0] == 0.0)
    assert np.abs(qa[1:]).max() <= 0.1


def test_benchmark_scenes_shapes():
    c = sceneChain(32)
    c.init()
    assert (c.nr, c.nm) == (32, 192)
    t = sceneTree(64)
    t.init()
    assert len(t.joints) == 64 and t.nr == 64
    types = {j.jtype for j in t.joints}
    assert types == {1, 2}


def test_roofline_calibration_matches_the_built_kernel():
    """bench.py's rooflines are built from SQ counter calibrations (profiles/roofline_calibration.json): counter totals of the default
    launch of every workload and, where the per-stage instruction counts are static, a two-parameter model.  They belong to ONE build of
    the kernels: every entry stores the opcode hashes of the kernels it was measured on, __graft_entry__.build() writes those of the
    library it linked (tools/kernel_fingerprints.py, read back from the .so), and a kernel change without a fresh calibration fails
    here (and leaves that workload's roofline without achieved / frac)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import __graft_entry__ as ge
    ge.build()
    fp = json.load(open(bench.FINGERPRINT_FILE))
    assert any("k_step_bdf1<32, false, false, true" in v["name"] for v in fp.values())
    for key in ("chain", "tree64", "tree64x", "ground", "adjoint", "chain128"):
        cal, stale = bench.load_calibration(key)
        assert cal is not None and stale is None, (key, stale)
        assert cal["launch"]["flops"] > 0 and cal["launch"]["SQ_INSTS_VALU"] > 0 and cal["newton_iters"] > 0
        assert bench.algorithm_flops(key) is not None
    cal, _ = bench.load_calibration("chain")
    # the headline workload is calibrated on the kernel that runs it: the two-point kernel of rmx_pair32.h
    assert len(cal["kernels"]) == 1 and cal["kernels"][0].startswith("k_step_bdf1_pair32") and cal.get("per_wave_basis") == "fronts_beyond_iters"
    assert any("k_step_bdf1_pair32" in v["name"] and v["scratch_bytes"] == 0 for v in fp.values())
    for k in ("flops", "SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MFMA_MOPS_F64"):
        assert cal["per_wave"][k]["front"] >= 0 and cal["per_wave"][k]["newton"] > 0
    assert abs(cal["per_wave"]["SQ_INSTS_VALU_MFMA_MOPS_F64"]["newton"] - 60.0) < 0.5      # 15 v_mfma_f64_16x16x4_f64 per Newton iteration
    # the headline kernel holds no scratch (kernel resources are part of the fingerprint file)
    head = [v for v in fp.values() if "k_step_bdf1<32, false, false, true" in v["name"]][0]
    assert head["scratch_bytes"] == 0 and head["vgpr"] <= 512


def test_design_kernel_table_is_the_built_library():
    """DESIGN.md quotes registers / scratch / instruction counts of the kernels in ONE table, generated by tools/kernel_table.py from
    the fingerprints of the library __graft_entry__.build() linked.  A kernel change without regenerating the table fails here (the
    round-5 review found prose figures that contradicted the traces)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    ge.build()
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_table.py")], capture_output=True, text=True, check=True).stdout
    rows = [ln for ln in out.splitlines() if ln.startswith("| `k_")]
    assert len(rows) >= 10
    design = open(os.path.join(root, "DESIGN.md")).read()
    missing = [r for r in rows if r not in design]
    assert not missing, "DESIGN.md's kernel table is stale (python tools/kernel_table.py):\n" + "\n".join(missing)
