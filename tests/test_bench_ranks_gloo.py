"""bench.py's own rank code (shard plans, input generation from global indices, the timed region with its barriers, the final
gather, max-over-ranks timing, JSON assembly) driven at world_size 2 on CPU: torch.distributed backend gloo, and an
oracle-backed stepper standing in for the HIP one (same methods as bench.GpuStepper; there is no GPU here).  Covers the weak
line, the strong-scaling line with UNEVEN shards (3 rollouts over 2 ranks: the padded gather), and the side measurements."""
import json

import pytest
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleStepper:
    """bench.GpuStepper's interface on the CPU oracle (test stand-in for the kernel)."""

    def __init__(self, scene, batch, device, integ="bdf1"):
        self.scene, self.B, self.nr = scene, batch, scene.nr
        self.h, self.tol = scene.h, 1e-9
        self.q = np.zeros((batch, scene.nr))
        self.qd = np.zeros((batch, scene.nr))
        self._acc = None
        self.stats_reset()

    def set_opts(self, h, tol, compensated=1):
        self.h, self.tol = h, tol           # (the literal oracle has plain doubles only)

    def set_state(self, q, qd):
        self.q, self.qd = np.ascontiguousarray(q, dtype=np.float64).copy(), np.ascontiguousarray(qd, dtype=np.float64).copy()

    def get_state(self):
        return self.q.copy(), self.qd.copy()

    def _step(self, K):
        from oracle import oracle as orc
        orc.set_newton(tol=self.tol)
        c = orc.batch_step_bdf1(self.scene.desc(), self.q, self.qd, self.h, K, nthreads=1, counters=True)
        orc.set_newton()
        return c

    def warmup(self, W):
        if W > 0:
            self._step(W)

    def stats_reset(self):
        self._acc = {"newton_iters": np.zeros(self.B, dtype=np.int32), "ls_halvings": np.zeros(self.B, dtype=np.int32),
                     "status": np.zeros(self.B, dtype=np.int32)}

    def sync_device(self):
        pass

    def launch(self, K):
        c = self._step(K)
        self._acc["newton_iters"] += c["newton_iters"]
        self._acc["ls_halvings"] += c["ls_halvings"]
        self._acc["status"] |= np.where(c["bad"] > 0, 2, 0).astype(np.int32)

    def wait(self):
        return 1.0

    def stats(self):
        return self._acc

    def state_tensors(self, torch, on_device):
        assert not on_device
        return torch.from_numpy(self.q.copy()), torch.from_numpy(self.qd.copy())

    def close(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    args = bench.parse_args(["--gpus", str(world), "--links", "4", "--batch", "3", "--steps", "3", "--warmup", "1", "--repeats", "1", "--ref-steps", "4",
                             "--json-out", out])
    assert bench.rank_main(args, make_stepper=OracleStepper, backend="gloo") == 0


def test_bench_rank_code_world_size_2(tmp_path):
    out = str(tmp_path / "line.json")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    d = json.load(open(out))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["global_batch"] == 6 and d["config"]["gathered_rows"] == 6      # weak: 3 per rank, all 6 rows gathered
    assert abs(d["value"] - 6 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-3 * d["value"]
    assert d["config"]["all_finite"] and d["config"]["not_converged_trajectories"] == 0
    assert d["repeat"]["launches"] == 2
    s = d["strong_scaling"]
    assert s["global_batch"] == 3 and s["value"] > 0                                   # 2 + 1 rollouts: uneven shards
    assert d["value_strong"] == s["value"]                                             # the metric-conformant figure, top level
    k = d["kernel_ms_per_rank"]
    assert len(k["weak"]) == 2 and len(k["strong"]) == 2 and min(k["weak"] + k["strong"]) > 0
    assert d["config"]["newton_tol"] == 1e-9 == d["config"]["reference_newton_tol"]      # the headline runs the reference's constant
    r = d["value_plain_iterate"]
    assert r["newton_tol"] == 1e-9 and r["steps"] == 4 and r["value"] > 0 and r["all_finite"]
    assert d["value_at_survey_init"]["value"] > 0 and d["value_at_survey_init"]["steps"] == 4
    assert d["value_at_tol_1e-8"]["steps"] == 3
    assert d["roofline"] is None and "cpu_baseline" not in d                            # GPU-only objects


def test_self_launch_command_line(monkeypatch):
    """`python bench.py --gpus N` without a torchrun environment re-executes under torch.distributed.run with N ranks."""
    import bench
    seen = {}
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main(["--gpus", "4", "--steps", "7"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
