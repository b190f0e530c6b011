"""bench.py keeps the driver's contract: ONE JSON line with the metric of BASELINE.json, whole-job value, the roofline and
cpu_baseline objects; the extra workloads print the same shape without them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_default_workload_line():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    d = _run("--steps", "20", "--warmup", "5", "--cpu-steps", "4", "--cpu-traj", "16")
    assert d["metric"] == base["metric"] and d["unit"] == "rollout-steps/s"
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 20, 5)
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1024 * 20 / (d["ms_per_step"] * 20 / 1e3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "TFLOP/s" and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["kernel_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert d["q_l2_relerr_vs_oracle_max"] < 1e-8
    assert d["config"]["all_finite"] and d["config"]["not_converged_trajectories"] == 0


@pytest.mark.parametrize("wl", ["tree64", "ground"])
def test_extra_workload_lines(wl):
    d = _run("--workload", wl, "--steps", "10", "--warmup", "2", "--batch", "64")
    assert d["value"] > 0 and d["roofline"] is None and "cpu_baseline" not in d
    assert d["config"]["batch_per_gpu"] == 64 and d["config"]["all_finite"]
