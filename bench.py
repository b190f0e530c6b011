#!/usr/bin/env python
"""bench.py -- sim steps/s of the batched RedMax BDF1 step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 32-link serial revolute chain (scenesRedMax.m:52-79 pattern), BDF1, fp64, 1024
independent rollouts.  A "step" is one BDF1 step of a rank's batch; `value` = rollout-steps per second summed over all
ranks, inputs resident in HBM before the timed region, the K steps of a rank are ONE kernel launch, and the only
collective is the final all-gather of (q, qdot) (redmax_amd/sharding.py; RCCL when every rank has its own GPU).

`value` / `scaling`: weak scaling - 1024 rollouts PER GPU (the batch axis shards, per-GPU work fixed).  The same run also
measures the strong-scaling reading of BASELINE.json's north_star (1024 rollouts IN TOTAL, 1024/N per GPU) and reports it
as `strong_scaling`; at N = 1 the two coincide.

Objects on the JSON line besides the driver's contract:
  roofline                 executed fp64 work of the dominant kernel against the fp64 peak (frac <= 1), its issue-bound
                           ceiling, the SURVEY §8(d) "algorithmic" figure for reference, HBM traffic per launch
  repeat                   the K-step launch repeated from the same state: median / min / max kernel time
  value_plain_iterate      the same launch with the Newton iterate in plain doubles (rmx_opts.compensated = 0), 100 steps
  value_at_survey_init     the same launch from SURVEY 8(d)'s wide initial-state ranges, 100 steps
  value_at_max_valid_init  the same launch at the largest initial-state amplitude the reference algorithm survives (measured: 0.1856)
  value_at_tol_1e-8        the tolerance rounds 1 and 2 ran the headline at
  strong_scaling           1024 rollouts in total over the N ranks; its figure is also the top-level `value_strong` (N > 1)
  kernel_ms_per_rank       (N > 1) HIP-event kernel time of every rank's timed launch, weak and strong plan, in rank order
  cpu_baseline             the literal CPU restatement of the reference (oracle) on the host cores, bounded sample
  cpu_baseline_tensor_free the tensor-free CPU implementation (the algorithm the GPU executes), same sample protocol
  newton_count_agreement   per-rollout Newton iteration counts, GPU vs oracle, on the in-run sample
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

METRIC = "sim steps/sec (whole node), 1024-batch 32-DOF chain BDF1; q L2 err vs ref"     # BASELINE.json, verbatim

# SURVEY.md §8(d) "algorithmic" flop counts for the n=32 serial chain (1 per add/mul, FMA = 2): what a sparsity-exploiting
# evaluation of the reference's J / dJdq formulation would cost.  The kernels execute ~10x less (O(n^2) world-frame
# recursion), so this figure is reported as `algorithmic_equiv_tflops`, NOT as the roofline fraction.
F_G, F_H, F_LU = 363712, 1418432, 23893

# EXECUTED work per workload: profiles/roofline_calibration.json, produced by tools/roofline_from_pmc.py from the SQ instruction counters
# of the profiled bench commands (separate rocprofv3 --pmc passes).  Per workload it holds (i) the counter TOTALS of the timed launch of
# the default bench command - the workloads are deterministic, so a run with the same signature (steps, warm-up, batch, tol) repeats the
# same Newton iterations and the totals ARE its executed work - and (ii) where the instruction counts per stage are static (chain,
# tree64) the two-parameter model counts(launch) = front_evals * FRONT + newton_iters * NEWTON fitted on two launches with different
# iterations-per-step mixes, valid for any --steps.  flops = 64 lanes x (ADD_F64 + MUL_F64 + 2 FMA_F64) + 512 x MFMA_MOPS_F64: what the
# SIMD spent, idle lanes included (`frac`).  The lane-honest companion `useful_frac` prices the same launch with the flops the ALGORITHM
# needs (profiles/algorithm_flops.json: the scalar CPU twin compiled with a counting double, tests/flop_count.py).  Each entry also
# holds the opcode hashes of the kernels it was measured on; __graft_entry__.build() writes those of the library it has just linked
# (tools/kernel_fingerprints.py) next to it.  If they differ the calibration is stale: the roofline object then says so and carries no
# achieved / frac (tests/test_host_logic.py and tests/test_gpu_bench_contract.py fail on it).
CALIBRATION_FILE = os.path.join(ROOT, "profiles", "roofline_calibration.json")
FINGERPRINT_FILE = os.path.join(ROOT, "redmax_amd", "kernel_fingerprint.json")
ALGORITHM_FLOPS_FILE = os.path.join(ROOT, "profiles", "algorithm_flops.json")


def workload_key(wl, n, B_local):
    """key of a bench workload in the calibration / algorithm-flops files"""
    if wl == "chain":
        return "chain" if n == 32 else "chain%d" % n
    if wl == "tree64":
        return "tree64x" if B_local > 512 else "tree64"      # more than two rollouts per CU: the kernels with the constants in global memory
    return wl


def load_calibration(key="chain"):
    """(calibration entry of the workload or None, stale reason or None)"""
    try:
        cal = json.load(open(CALIBRATION_FILE))
    except (OSError, ValueError) as e:
        return None, "no calibration file (%s)" % e
    ent = (cal.get("workloads") or {}).get(key)
    if ent is None:
        return None, "profiles/roofline_calibration.json has no entry for workload %r: run tools/gpu_session.sh <tag> pmc and tools/roofline_from_pmc.py" % key
    try:
        fp = json.load(open(FINGERPRINT_FILE))
    except (OSError, ValueError) as e:
        return ent, "the library carries no kernel fingerprints (%s): run __graft_entry__.build()" % e
    if not ent.get("fingerprints"):
        return ent, "the calibration entry names no kernel fingerprints"
    for sym, sha in ent["fingerprints"].items():
        have = (fp.get(sym) or {}).get("opcode_sha16")
        if have != sha:
            return ent, ("calibrated on %s = %s, the built kernel is %s: re-run tools/gpu_session.sh <tag> pmc and "
                         "tools/roofline_from_pmc.py" % (sym[:48], sha, have))
    return ent, None


def algorithm_flops(key):
    """(flops per front evaluation, flops per Newton iteration beyond its front, note) of the scalar algorithm, or None"""
    try:
        w = json.load(open(ALGORITHM_FLOPS_FILE))["workloads"]
    except (OSError, ValueError, KeyError):
        return None
    e = w.get({"tree64x": "tree64", "adjoint": "adjoint16"}.get(key, key))
    return (e["per_front"], e["per_newton"], e.get("note"), e.get("n")) if e else None


# The ceiling of newton_count_agreement.vs_oracle at the reference's tol: the literal oracle against ITSELF from initial states one unit in
# the last place apart (tools/newton_count_dither.py, profiles/r06_newton_count_dither.txt: 64 rollouts x 23 steps of this workload).  At
# tol 1e-9 the last iterations of a step test |g| against tol at the resolution of doubles, so the count of a step is decided by
# rounding: no implementation of the same mathematics can agree with the oracle more often than the oracle agrees with itself.
ORACLE_SELF_AGREEMENT = {"frac": None, "frac_at_tol_1e-8": None, "source": "profiles/r06_newton_count_dither.txt"}
try:
    for _ln in open(os.path.join(ROOT, "profiles", "r06_newton_count_dither.txt")):
        if _ln.startswith("{"):
            _d = json.loads(_ln)["tols"]
            ORACLE_SELF_AGREEMENT["frac"] = _d["1e-09"]["frac"]
            ORACLE_SELF_AGREEMENT["frac_at_tol_1e-8"] = _d["1e-08"]["frac"]
except (OSError, ValueError, KeyError):
    pass

FP64_PEAK_TFLOPS = 78.6   # MI355X datasheet: FP64 vector = FP64 matrix = 78.6 TFLOP/s (the microarch guide has no fp64 row)
SHADER_CLOCK_GHZ = 2.4    # max clock (guide); the effective clock under load is lower, so cycle counts below are upper bounds
N_SIMD = 1024             # 256 CUs x 4 SIMDs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="steps in the timed region (default: 100; adjoint: a 20-step horizon, the longest "
                                                          "round horizon over which the reference's line-search-free Newton converges on every step)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="rollouts per GPU (default: 1024; 512 for tree64 and adjoint)")
    ap.add_argument("--workload", choices=("chain", "tree64", "ground", "adjoint"), default="chain",
                    help="chain: BASELINE.json configs[1], the headline metric (default).  tree64: configs[2], 64-joint "
                         "revolute/prismatic tree, BDF1.  ground: configs[4], 32-link chain over frictional ground, BDF2.  adjoint: "
                         "configs[3], 16-DOF chain, forward + backward adjoint sweep (HBM roofline).  Only chain carries cpu baselines")
    ap.add_argument("--links", type=int, default=32)
    ap.add_argument("--tol", type=float, default=1e-9, help="Newton |g| tolerance: the reference's hard-coded 1e-9 (driverRedMaxBDF1.m:95)")
    ap.add_argument("--plain-iterate", action="store_true", help="rmx_opts.compensated = 0 for the headline line (plain doubles)")
    ap.add_argument("--repeats", type=int, default=5, help="extra timed launches of the same K steps from the same state")
    ap.add_argument("--burn-in", type=float, default=60.0, help="milliseconds of untimed launches of the same K steps, each from the initial state, BEFORE "
                                                                "the W warm-up steps (brings the GPU out of its idle clock state); 0: none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-tol", "--no-side-legs", dest="no_reference_tol", action="store_true",
                    help="skip the side measurements (plain iterate, SURVEY init ranges, tol 1e-8)")
    ap.add_argument("--ref-steps", type=int, default=100, help="steps of the side measurements: the reference's rollout length "
                                                               "(tEnd = 1 at h = 1e-2, Scene.m:117), whatever --steps is")
    ap.add_argument("--no-strong", action="store_true")
    ap.add_argument("--cpu-traj", type=int, default=0, help="rollouts in the CPU sample (default: one per host thread, <= 256)")
    ap.add_argument("--cpu-steps", type=int, default=40)
    ap.add_argument("--json-out", default="", help="also write the JSON line to this file (tests)")
    args = ap.parse_args(argv)
    if args.steps <= 0:
        args.steps = 20 if args.workload == "adjoint" else 100
    return args


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: run the N ranks under torch.distributed.run."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------- the stepper a rank drives
class GpuStepper:
    """One rank's batch on one MI355X (redmax_amd.BatchSim = the C ABI).  tests/test_bench_ranks_gloo.py substitutes an
    oracle-backed object with the same methods to drive this file's rank code on CPU."""

    def __init__(self, scene, batch, device, integ="bdf1"):
        from redmax_amd import BatchSim
        self.sim = BatchSim(scene, batch=batch, device=device)
        self.integ = integ
        self.device = device
        self.B, self.nr = batch, scene.nr
        self._out = None

    def set_opts(self, h, tol, compensated=1, ls_fail_limit=0):
        self.sim.opts.h = h
        self.sim.opts.tol = tol
        self.sim.opts.compensated = int(compensated)
        self.sim.opts.ls_fail_limit = int(ls_fail_limit)

    def set_state(self, q, qd):
        self.sim.set_state(q, qd)

    def get_state(self):
        return self.sim.get_state()

    def warmup(self, W):
        if W > 0:
            (self.sim.step_bdf2 if self.integ == "bdf2" else self.sim.step_bdf1)(W)

    def stats_reset(self):
        self.sim.stats_reset()
        self.sim.sync()

    def sync_device(self):
        self.sim.sync()

    def launch(self, K):
        # all K steps of all rollouts: one call, one kernel launch (configs[4]: rollouts and cooperative groups in one launch);
        # the counters stay on the device until stats()
        (self.sim.step_bdf2_async if self.integ == "bdf2" else self.sim.step_bdf1_async)(K)
        self._out = None

    def wait(self):
        """kernel milliseconds of the launch (HIP events on the kernel's own stream)."""
        return self._out["ms"] if self._out is not None else self.sim.sync()

    def stats(self):
        return self._out if self._out is not None else self.sim.stats_read()

    def rollout_ticks(self):
        """s_memtime ticks every rollout's wavefront spent in the last launch (the launch ends with its slowest rollout)."""
        return self.sim.step_ticks()

    def step_kernel(self):
        """label of the step kernel the library chose for the last launch (rmx_last_step_kernel)"""
        return self.sim.last_step_kernel()

    def state_tensors(self, torch, on_device):
        """(q, qdot) of this rank as torch tensors for the gather: device tensors (RCCL) or host tensors (gloo)."""
        if on_device:
            dev = torch.device("cuda", self.device)
            q = torch.empty((self.B, self.nr), dtype=torch.float64, device=dev)
            qd = torch.empty_like(q)
            self.sim.get_state_device(q.data_ptr(), qd.data_ptr())
            return q, qd
        q, qd = self.sim.get_state()
        return torch.from_numpy(q), torch.from_numpy(qd)

    def close(self):
        self.sim.close()


def build_workload(wl, links):
    from redmax_amd import sceneChain, sceneChainGround, sceneTree, syntheticStates
    if wl == "chain":
        scene = sceneChain(links)
        scene.init()
        return scene, 1e-2, "bdf1", (lambda first, count: syntheticStates(scene.nr, count, first=first))
    if wl == "tree64":
        scene = sceneTree(64)
        scene.init()
        qs, _ = scene.getQ()

        def gen(first, count):
            q0, qd0 = np.empty((count, scene.nr)), np.empty((count, scene.nr))
            for i in range(count):
                rng = np.random.default_rng(20240 + first + i)
                q0[i] = qs + rng.uniform(-0.05, 0.05, scene.nr)
                qd0[i] = rng.uniform(-0.1, 0.1, scene.nr)
            return q0, qd0
        return scene, 1e-2, "bdf1", gen
    scene = sceneChainGround(32)
    scene.init()

    def geng(first, count):
        q0, qd0 = syntheticStates(scene.nr, count, first=first, sq=5e-4, sv=0.1)     # every chain starts above the ground
        if first == 0 and count:
            q0[0], qd0[0] = scene.getQ()
        return q0, qd0
    return scene, scene.h, "bdf2", geng


class RankContext:
    """Process-group plumbing of one rank: barrier, gather, max-reduction, on RCCL (one GPU per rank) or gloo."""

    def __init__(self, rank, world, torch, dist, on_device, device, sync_cuda):
        self.rank, self.world, self.torch, self.dist, self.device = rank, world, torch, dist, device
        self.on_device = on_device      # collectives run on device tensors (RCCL); False: host tensors (gloo)
        self.sync_cuda = sync_cuda      # the barrier also waits for the device

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.sync_cuda:
            self.torch.cuda.synchronize()

    def gather(self, stepper, shard):
        if self.dist is None:
            return None
        from redmax_amd import sharding
        q, qd = stepper.state_tensors(self.torch, self.on_device)
        return sharding.gather_states(q, qd, shard)

    def max(self, x):
        from redmax_amd import sharding
        dev = self.torch.device("cuda", self.device) if (self.on_device and self.dist is not None) else None
        return sharding.max_over_ranks(x, dev) if self.dist is not None else float(x)

    def per_rank(self, x):
        """x of every rank, in rank order (what makes a multi-GPU line readable at a glance: which rank was the slow one)."""
        if self.dist is None:
            return [float(x)]
        dev = self.torch.device("cuda", self.device) if self.on_device else "cpu"
        mine = self.torch.tensor([float(x)], dtype=self.torch.float64, device=dev)
        allx = self.torch.empty(self.world, dtype=self.torch.float64, device=dev)
        self.dist.all_gather_into_tensor(allx, mine)
        return [float(v) for v in allx.cpu().tolist()]

    def sum_int(self, x):
        if self.dist is None:
            return int(x)
        t = self.torch.tensor([int(x)], dtype=self.torch.int64, device=self.torch.device("cuda", self.device) if self.on_device else "cpu")
        self.dist.all_reduce(t)
        return int(t.item())


def measure(ctx, make_stepper, scene, gen, shard, h, tol, integ, K, W, repeats, compensated=1, burn_in=0, ls_fail_limit=0):
    """The contract's timed region for one shard plan: W untimed warm-up steps, then EXACTLY K steps bracketed by barrier +
    device sync on both sides, MAX over ranks; then `repeats` more launches of the same K steps from the same (post-warm-up)
    state for the spread.  Returns a dict (identical on every rank where it matters)."""
    if shard.count < 1:
        raise SystemExit("rank %d owns no rollout (batch %d over %d ranks)" % (shard.rank, shard.global_batch, shard.world))
    if integ == "bdf2":
        repeats = 0                    # a restored state restarts BDF2 with its SDIRK2 step: not the same work as the timed launch
    st = make_stepper(scene, shard.count, ctx.device, integ)
    if ls_fail_limit:
        st.set_opts(h, tol, compensated, ls_fail_limit)
    else:
        st.set_opts(h, tol, compensated)
    q0, qd0 = gen(shard.first, shard.count)
    st.set_state(q0, qd0)
    ctx.gather(st, shard)              # warm the collective too (before the warm-up steps: nothing but the barrier + device
    burned, burn_ms = 0, 0.0           # sync the contract asks for lies between the warm-up steps and the timed launch)
    while burn_ms < burn_in and burned < 200:
        # burn-in: untimed launches of the very kernel that will be timed, each from the initial state, until the GPU has been busy
        # for `burn_in` milliseconds: an idle MI355X runs its first ~20 ms of work about 5 % below its sustained clock (s_memtime
        # ticks per rollout stay the same, the launch takes longer: tools/first_launch_probe.py).  The state is put back afterwards.
        st.set_state(q0, qd0)
        st.launch(K)
        burn_ms += st.wait()
        burned += 1
    if burned:
        # ... and ONE untimed rehearsal of the timed sequence itself (counter reset, the synchronous W warm-up steps, device sync,
        # barrier, the K-step launch, wait, gather, barrier): the first launch behind the process's first synchronous step call pays
        # 13 - 30 us of one-time host / runtime set-up (tools: profiles/r6n_timed_region_probe.txt), 2 - 4 % of a 0.8 ms launch
        st.set_state(q0, qd0)
        st.stats_reset()
        st.warmup(W)
        st.sync_device()
        ctx.barrier()
        st.launch(K)
        st.wait()
        ctx.gather(st, shard)
        ctx.barrier()
        st.set_state(q0, qd0)
    st.stats_reset()
    st.warmup(W)
    st.sync_device()
    ctx.barrier()
    t0 = time.perf_counter()
    st.launch(K)
    kernel_ms = st.wait()
    gathered = ctx.gather(st, shard)   # the single collective of the path: final gather of (q, qdot)
    ctx.barrier()
    elapsed = ctx.max(time.perf_counter() - t0)
    s = st.stats()
    qf, qdf = st.get_state()
    out = {
        "elapsed": elapsed, "kernel_ms": ctx.max(kernel_ms), "kernel_ms_per_rank": [round(v, 4) for v in ctx.per_rank(kernel_ms)],
        "iters": ctx.sum_int(s["newton_iters"].sum()), "halvings": ctx.sum_int(s["ls_halvings"].sum()),
        "bad": ctx.sum_int(((s["status"] & 15) != 0).sum()), "pivoted": ctx.sum_int(((s["status"] & 16) != 0).sum()),
        "finite": bool(np.isfinite(qf).all() and np.isfinite(qdf).all()), "rollouts": shard.global_batch,
        "gathered_rows": int(gathered[0].shape[0]) if gathered is not None else shard.count,
        "local_iters": s["newton_iters"].copy(),
        "burned": burned,
        "step_kernel": st.step_kernel() if hasattr(st, "step_kernel") else None,
    }
    if hasattr(st, "rollout_ticks"):      # how the launch time is spread over this rank's rollouts (all run concurrently, one wavefront each)
        tk = st.rollout_ticks().astype(np.float64)
        if tk.max() > 0:
            ms = kernel_ms * tk / tk.max()
            out["rollout_ms"] = {"p50": round(float(np.percentile(ms, 50)), 4), "p90": round(float(np.percentile(ms, 90)), 4),
                                 "p99": round(float(np.percentile(ms, 99)), 4), "max": round(float(ms.max()), 4),
                                 "second_slowest": round(float(np.sort(ms)[-2]), 4) if ms.size > 1 else None, "slowest_rollout": int(np.argmax(ms)) + shard.first,
                                 "note": "per-rollout share of the K-step launch (rmx_step_ticks, scaled so that the slowest rollout = the "
                                         "kernel time): every rollout has its own wavefront, the launch ends with the slowest one"}
    rep_k, rep_w = [], []
    if repeats > 0:                    # the state every timed launch starts from: the same warm-up once more (deterministic)
        st.set_state(q0, qd0)
        st.warmup(W)
        qw, qdw = st.get_state()
    for _ in range(max(repeats, 0)):
        st.set_state(qw, qdw)
        ctx.barrier()
        t0 = time.perf_counter()
        st.launch(K)
        rep_k.append(ctx.max(st.wait()))
        ctx.gather(st, shard)
        ctx.barrier()
        rep_w.append(ctx.max(time.perf_counter() - t0))
    if rep_k:
        allk = sorted(rep_k + [out["kernel_ms"]])
        allw = sorted(rep_w + [elapsed])
        out["repeat"] = {"launches": len(allk), "kernel_ms_median": round(float(np.median(allk)), 4), "kernel_ms_min": round(allk[0], 4),
                         "kernel_ms_max": round(allk[-1], 4), "wall_ms_median": round(1e3 * float(np.median(allw)), 4),
                         "value_median": round(shard.global_batch * K / float(np.median(allw)), 1),
                         "note": "the timed K-step launch repeated from the same post-warm-up state (same work every time); "
                                 "`value` is the first launch, as the contract defines the timed region"}
    st.close()
    return out


def roofline(kernel_ms, iters_rank, halv_rank, steps_rank, K, W, B_local, n, wl, tol, local_iters=None, fronts=None, extra_useful=0.0,
             step_kernel=None):
    """Executed-work roofline of one rank's timed launch (rank 0's counters stand for all: identical work distribution by
    construction).  bound = "valu-issue": every step kernel of this library runs one wavefront per SIMD (or per rollout) through
    a sequential Newton chain, so what bounds it is the issue rate of a lone wavefront against the fp64 peak - not HBM (the state
    is read and written once per launch) and not a pipe.
      achieved / frac   executed fp64 flops (64 lanes per wave-wide instruction) / kernel time (HIP events) / 78.6 TF
      useful_frac       the flops the scalar algorithm needs for the same evaluations and iterations / the same time / peak"""
    if kernel_ms <= 0:
        return None
    key = workload_key(wl, n, B_local)
    ent, stale = load_calibration(key)
    alg_fronts = fronts
    if fronts is None:
        alg_fronts = fronts = steps_rank + iters_rank + halv_rank          # one per step (initial guess) + one per line-search trial
        if step_kernel == "k_step_bdf1_pair32":
            # the two-point kernel (rmx_pair32.h): the EXECUTED evaluations of the front are the line-search trials plus the first evaluation
            # of each rollout's launch - every later step's first point rides in the idle half-wave of the trial that ends the step before
            # it (the rare solves that end on a stall / a diverged update re-evaluate: not counted).  The algorithm's evaluations
            # (useful_frac) stay one per step + one per trial: the riding point IS that evaluation.
            fronts = iters_rank + halv_rank + B_local
    sec = kernel_ms * 1e-3
    out = {"bound": "valu-issue", "achieved": None, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "useful_frac": None, "traffic": None,
           "kernel": ", ".join(k.split("(")[0].replace("void ", "") for k in ent["kernels"]) if ent else None, "kernel_ms": round(kernel_ms, 4),
           "newton_iters_per_step": round(iters_rank / max(steps_rank, 1), 3), "ls_halvings_per_step": round(halv_rank / max(steps_rank, 1), 4),
           "front_evals": fronts, "newton_iters": iters_rank}
    if step_kernel:
        out["kernel"] = step_kernel      # what the library says it launched (rmx_last_step_kernel), not a guess from batch thresholds
        if ent and not any(step_kernel.split("<")[0] in k for k in ent["kernels"]):
            stale = "calibrated on %s, the library launched %s" % (", ".join(ent["kernels"]), step_kernel)
    if wl == "chain" and n == 32:
        alg = (iters_rank * (F_H + F_LU) + (iters_rank + halv_rank) * F_G) / sec / 1e12
        out["algorithmic_equiv_tflops"] = round(alg, 2)
        if fronts != alg_fronts:
            out["front_evals_note"] = ("front_evals = EXECUTED evaluations (each carries two points: the trial and the next step's first point); the "
                                       "algorithm's %d evaluations (one per step + one per trial) price useful_frac" % int(alg_fronts))
    af = algorithm_flops(key)
    if af is not None:
        useful = alg_fronts * af[0] + iters_rank * af[1] + extra_useful
        out["useful_tflops"] = round(useful / sec / 1e12, 3)
        out["useful_frac"] = round(useful / sec / 1e12 / FP64_PEAK_TFLOPS, 4)
        out["useful_note"] = ("flops the ALGORITHM needs (scalar CPU twin compiled with a counting double, tests/flop_count.py -> "
                              "profiles/algorithm_flops.json: %d per residual evaluation, %d per Newton iteration beyond it = Hessian + solve) x the "
                              "evaluation / iteration counts of this launch / kernel time / peak: no idle lanes of a partly filled wave, no replicated "
                              "pivot columns, no masked MFMA tiles counted%s" % (af[0], af[1], ("; " + af[2]) if af[2] else ""))
    if ent is None or stale:
        out["calibration_stale"] = stale
        return out
    sig = {"steps": K, "warmup": W, "batch": B_local, "tol": tol, "links": n}
    same = all(ent["signature"].get(k) == v for k, v in sig.items() if ent["signature"].get(k) is not None) and \
        ent.get("newton_iters") == iters_rank and ent.get("front_evals") == fronts
    ex = ent.get("per_wave")
    if same:
        flops, valu, how = ent["launch"]["flops"], ent["launch"]["SQ_INSTS_VALU"], "counter totals of this very launch (same signature, same Newton iteration count: the workload is deterministic)"
    elif ex:
        # (two-point kernel: NEWTON is a whole iteration including its one front, FRONT prices the evaluations beyond one per iteration)
        xf = fronts - iters_rank if ent.get("per_wave_basis") == "fronts_beyond_iters" else fronts
        flops = xf * ex["flops"]["front"] + iters_rank * ex["flops"]["newton"]
        valu = xf * ex["SQ_INSTS_VALU"]["front"] + iters_rank * ex["SQ_INSTS_VALU"]["newton"]
        how = "per-stage counts (front evaluation / Newton iteration, fitted on two counter passes) x the counts of this launch"
    else:
        scale = iters_rank / max(ent.get("newton_iters") or 1, 1)
        flops, valu = ent["launch"]["flops"] * scale, ent["launch"]["SQ_INSTS_VALU"] * scale
        how = "counter totals of the calibrated launch (%s) scaled by the Newton-iteration ratio %.4f: an estimate" % (json.dumps(ent["signature"]), scale)
    ach = flops / sec / 1e12
    # issue-bound ceiling: the fp64 pipe takes one wave-wide instruction per 4 cycles (16 lanes / clk / SIMD)
    waves = B_local * (4 if key.startswith("chain") and n > 64 else 1)      # trees of more than 64 nodes: four wavefronts per rollout
    valu_wave = valu / waves
    slowest = (float(local_iters.max()) / max(float(local_iters.mean()), 1.0)) if local_iters is not None else 1.0
    cycles = sec * SHADER_CLOCK_GHZ * 1e9
    hb = ent.get("hbm_kb_per_launch", {})
    out.update({
        "achieved": round(ach, 3), "frac": round(ach / FP64_PEAK_TFLOPS, 4), "executed_flops_from": how,
        "traffic": int((hb["fetch"] + hb["write"]) * 1024) if ("fetch" in hb and "write" in hb) else None,
        "traffic_note": ("HBM bytes of the calibrated launch (%s) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes): %.1f KB + %.1f KB; "
                         "reported at face value (the guide's x2 FETCH correction is calibrated for 16 B/lane streams)"
                         % (json.dumps(ent["signature"]), hb.get("fetch", 0.0), hb.get("write", 0.0))) if hb else None,
        "calibration": CALIBRATION_FILE.replace(ROOT + os.sep, ""), "kernel_opcode_sha16": sorted(ent["fingerprints"].values()),
        "issue_bound": {"valu_insts_per_wave": round(valu_wave, 1), "cycles_at_4_per_inst": round(4.0 * valu_wave * slowest, 1),
                        "kernel_cycles_at_%.1fGHz" % SHADER_CLOCK_GHZ: round(cycles, 1),
                        "frac": round(4.0 * valu_wave * slowest / cycles, 4),
                        "note": "one wavefront per SIMD: a lone wavefront issues one VALU instruction per ~6 cycles (tools/ubench2.hip: 6.0 ticks "
                                "per independent v_fma_f64 alone, 4.0 with two waves per SIMD), the fp64 pipe could take one per 4; frac = (VALU "
                                "instructions of the slowest wave x 4 cycles) / kernel cycles"},
        "note": "achieved = EXECUTED fp64 flops (SQ_INSTS_VALU_*_F64 / MFMA_MOPS_F64 counters, 64 lanes per wave-wide instruction) / kernel time "
                "measured with HIP events on the kernel's stream; peak = 78.6 TF (fp64 vector = fp64 matrix on MI355X); bound = issue rate of a lone "
                "wavefront, not a pipe.  useful_frac prices the same launch with the scalar algorithm's flops",
    })
    if out.get("useful_frac") is not None and out["useful_frac"] > out["frac"]:
        # the scalar twin behind `useful` runs a DENSE partial-pivot LU (oracle/redmax_tensorfree.c); on a branching tree the kernels
        # eliminate along the tree (tree_solve64: no fill outside the root paths) and execute fewer flops than that algorithm needs -
        # what they execute is all useful then, and the twin's count is kept beside it
        out["useful_tflops_of_the_dense_twin"] = out["useful_tflops"]
        out["useful_tflops"], out["useful_frac"] = out["achieved"], out["frac"]
        out["useful_note"] += ("; CAPPED at the executed work: the twin's dense LU is not what runs on this tree (the solve follows the tree's "
                               "sparsity), so the kernels execute fewer flops than the twin's algorithm needs")
    if ex and ent.get("per_wave_basis") == "fronts_beyond_iters":
        out.update({"executed_flops_per_newton_iter_incl_front": round(ex["flops"]["newton"], 1),
                    "valu_insts_per_newton_iter_incl_front": round(ex["SQ_INSTS_VALU"]["newton"], 1),
                    "valu_insts_per_extra_front_eval": round(ex["SQ_INSTS_VALU"]["front"], 1)})
    elif ex:
        out.update({"executed_flops_per_front_eval": round(ex["flops"]["front"], 1), "executed_flops_per_newton_iter": round(ex["flops"]["newton"], 1),
                    "valu_insts_per_front_eval": round(ex["SQ_INSTS_VALU"]["front"], 1),
                    "valu_insts_per_newton_iter_beyond_its_front": round(ex["SQ_INSTS_VALU"]["newton"], 1)})
    sq = ent.get("launch_sq")
    if sq and sq.get("SQ_ACTIVE_INST_VALU"):
        out["lane_occupancy_counter"] = {
            "SQ_THREAD_CYCLES_VALU": sq.get("SQ_THREAD_CYCLES_VALU"), "SQ_ACTIVE_INST_VALU": sq.get("SQ_ACTIVE_INST_VALU"),
            "active_lanes_per_valu_cycle": round(sq.get("SQ_THREAD_CYCLES_VALU", 0.0) / sq["SQ_ACTIVE_INST_VALU"], 2),
            "note": "EXEC-mask occupancy of the calibrated launch: lanes enabled per cycle with a VALU instruction active, of 64.  Idle lanes of "
                    "a partly filled wave mostly run UNMASKED on an identity / zero column (rmx_device.h), so this counter does not see them: "
                    "useful_frac, not this ratio, is the lane-honest figure"}
    return out


def rank_main(args, make_stepper=None, backend=None):
    import torch
    from redmax_amd import sharding
    rank, local_rank, world = sharding.env_rank()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    on_gpu = make_stepper is None
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the measured path)")
        make_stepper = GpuStepper
        ndev = torch.cuda.device_count()
        device = local_rank % ndev
        shared = world > ndev            # several ranks on one device: RCCL refuses duplicate GPUs, the gather goes through gloo
        torch.cuda.set_device(device)
        backend = backend or ("gloo" if shared else "nccl")
    else:
        device, shared, backend = 0, False, backend or "gloo"
    # RMX_BENCH_FORCE_DIST=1: form the process group at world size 1 as well - how the RCCL code path (device-tensor all-gather, device
    # max-reduction, barrier) is exercised on a 1-GPU box (tests/test_gpu_bench_contract.py); nothing else changes
    force = os.environ.get("RMX_BENCH_FORCE_DIST") == "1"
    dist = sharding.init_process_group(rank, world, backend, device if backend == "nccl" else None) if (world > 1 or force) else None
    ctx = RankContext(rank, world, torch, dist, on_device=on_gpu and backend == "nccl", device=device, sync_cuda=on_gpu)

    wl = args.workload
    if wl == "adjoint":
        return adjoint_main(args, ctx)
    if args.batch <= 0:
        args.batch = 512 if wl == "tree64" else 1024
    B, K, W = args.batch, args.steps, args.warmup
    scene, h, integ, gen = build_workload(wl, args.links)
    n = scene.nr
    weak = sharding.plan(rank, world, B, "weak")
    comp = 0 if args.plain_iterate else 1
    burn = args.burn_in if on_gpu else 0.0         # a clock ramp is a GPU matter: the CPU stand-ins of the tests skip it
    m = measure(ctx, make_stepper, scene, gen, weak, h, args.tol, integ, K, W, args.repeats, comp, burn)
    plain = strong = wide = soft = cutleg = maxv = None
    KR = args.ref_steps
    if wl == "ground" and on_gpu and not args.no_reference_tol:
        # the opt-in straggler policy (rmx_opts.ls_fail_limit, NOT reference behaviour): same launch, Newton loop of a step cut at its
        # second failed line search.  A side figure; `value` is the reference's loop.
        cutleg = measure(ctx, make_stepper, scene, gen, weak, h, args.tol, integ, K, W, 0, comp, burn, 2)
    if wl == "chain" and not args.no_reference_tol:
        # side measurements, always over the reference's own rollout length: the lattice of doubles binds (and the wide initial
        # states fail) late in a rollout, a short --steps window would hide it
        if comp:
            plain = measure(ctx, make_stepper, scene, gen, weak, h, args.tol, integ, KR, W, 0, 0, burn)
        from redmax_amd import syntheticStates
        wide = measure(ctx, make_stepper, scene, lambda first, count: syntheticStates(scene.nr, count, first=first, sq=np.pi / 4, sv=1.0),
                       weak, h, args.tol, integ, KR, W, 0, comp, burn)
        from redmax_amd.scenes import MAX_VALID_INIT_AMPLITUDE as AMAX
        maxv = measure(ctx, make_stepper, scene, lambda first, count: syntheticStates(scene.nr, count, first=first, sq=AMAX, sv=AMAX),
                       weak, h, args.tol, integ, KR, W, 0, comp, burn)
        if args.tol != 1e-8:
            soft = measure(ctx, make_stepper, scene, gen, weak, h, 1e-8, integ, K, W, 0, comp, burn)
    if world > 1 and not args.no_strong:
        strong = measure(ctx, make_stepper, scene, gen, sharding.plan(rank, world, B, "strong"), h, args.tol, integ, K, W, args.repeats, comp, burn)

    if rank == 0:
        value = m["rollouts"] * K / m["elapsed"]
        workload = {"chain": "%d-link serial revolute chain, BDF1 fp64, batch=%d per GPU (BASELINE.json configs[1])" % (n, B),
                    "tree64": "64-joint revolute/prismatic branching tree, BDF1 fp64, batch=%d per GPU (BASELINE.json configs[2])" % B,
                    "ground": "32-link chain over frictional ground (ForceGroundCuboid on every body), BDF2 fp64, h=5e-4, "
                              "batch=%d per GPU (BASELINE.json configs[4])" % B}[wl]
        out = {
            "metric": METRIC, "value": round(value, 1), "unit": "rollout-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * m["elapsed"] / K, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "batch_per_gpu": B, "global_batch": m["rollouts"], "links": n, "h": h,
                       "newton_tol": args.tol, "reference_newton_tol": 1e-9,
                       "newton_iterate": "compensated: x + xlo, xlo in v = x - qB and qdot only (rmx_opts.compensated = 1, the library default)" if comp
                                         else "plain doubles (rmx_opts.compensated = 0)",
                       "init": {"chain": "q,qdot~U(-0.1,0.1), rng(20240+global_index); traj 0: q=0.1,qdot=0",
                                "tree64": "scene state + U(-0.05,0.05), qdot~U(-0.1,0.1), rng(20240+global_index)",
                                "ground": "q~U(-5e-4,5e-4), qdot~U(-0.1,0.1), rng(20240+global_index); traj 0: the scene's state"}[wl],
                       "parallelism": "batch-sharded x%d (redmax_amd.sharding), one %s all-gather of the final (q,qdot)%s" % (
                           world, "RCCL" if backend == "nccl" else backend,
                           "; ranks share a device, so the gather runs on gloo with host tensors" if (on_gpu and shared) else ""),
                       "steps_per_launch": K, "untimed_burn_in": {"ms": burn, "launches": m.get("burned", 0), "rehearsal": "one untimed run of the whole timed sequence (W warm-up steps, sync, K-step launch, wait, gather) before the real one" if m.get("burned", 0) else None}, "not_converged_trajectories": m["bad"], "trajectories_with_pivoted_fallback": m["pivoted"],
                       "all_finite": m["finite"], "gathered_rows": m["gathered_rows"]},
            "roofline": roofline(m["kernel_ms"], m["iters"] / world, m["halvings"] / world, B * K, K, W, B, n, wl, args.tol, m["local_iters"],
                                 step_kernel=m.get("step_kernel")) if on_gpu else None,
        }
        if "repeat" in m:
            out["repeat"] = m["repeat"]
        if "rollout_ms" in m:
            out["rollout_ms"] = m["rollout_ms"]
            r = m["rollout_ms"]
            if wl == "chain" and world == 1 and r.get("second_slowest") and r.get("slowest_rollout") == 0:
                # the launch ends with its slowest wavefront, and that is rollout 0 (the deterministic q = 0.1, qdot = 0 state every round has
                # kept first): the same launch without it would end with the second slowest.  NOT the headline - the workload keeps rollout 0.
                t = m["elapsed"] - (r["max"] - r["second_slowest"]) * 1e-3
                out["value_without_rollout_0"] = {
                    "value": round((m["rollouts"] - 1) * K / t, 1), "unit": "rollout-steps/s", "kernel_ms": r["second_slowest"],
                    "note": "the timed launch if rollout 0 were not in the batch: wall time minus (slowest - second slowest wavefront), %d rollouts.  "
                            "Rollout 0 needs ~19 %% more Newton iterations than the 99th percentile; it stays in `value`" % (m["rollouts"] - 1)}
        if wl != "chain":
            out["metric"] = "sim steps/sec (whole node), " + {"tree64": "64-joint tree BDF1", "ground": "32-link chain + ground contact BDF2"}[wl]
            out["config"]["newton_iters_per_step"] = round(m["iters"] / (m["rollouts"] * K), 3)
            out["config"]["kernel_ms"] = round(m["kernel_ms"], 4)
        if plain is not None:
            out["value_plain_iterate"] = {
                "newton_tol": args.tol, "steps": KR, "value": round(plain["rollouts"] * KR / plain["elapsed"], 1), "unit": "rollout-steps/s",
                "ms_per_step": round(1e3 * plain["elapsed"] / KR, 5), "kernel_ms": round(plain["kernel_ms"], 4),
                "newton_iters_per_step": round(plain["iters"] / (plain["rollouts"] * KR), 3),
                "ls_halvings_per_step": round(plain["halvings"] / (plain["rollouts"] * KR), 3),
                "not_converged_trajectories": plain["bad"], "all_finite": plain["finite"],
                "note": "the same launch with rmx_opts.compensated = 0: the Newton iterate in plain doubles, the reference's arithmetic "
                        "decision for decision, over the reference's %d steps.  g depends on x at the resolution of one ulp through M (x - qB), "
                        "and |M| ulp(q) ~ 1e-9 = tol on this 320 cm cgs chain: on the lattice of doubles |g| < tol holds at lucky points only.  "
                        "The reference finds them because its own evaluation noise dithers the Newton update (literal oracle: 0.3 %% of the "
                        "trajectory-steps fail); the world-frame evaluation has a smoother error and sticks (10-13 %% fail, each after a full "
                        "20-halving line search), on the GPU as in its CPU twin (tests/test_gpu_reference_tol.py).  The headline carries the "
                        "iterate as x + xlo instead and converges on every step (DESIGN.md 5)" % KR}
        if wide is not None:
            out["value_at_survey_init"] = {
                "init": "q~U(-pi/4,pi/4), qdot~U(-1,1), rng(20240+global_index) (SURVEY.md 8(d)); traj 0: q=0.1,qdot=0", "newton_tol": args.tol,
                "steps": KR, "value": round(wide["rollouts"] * KR / wide["elapsed"], 1), "unit": "rollout-steps/s", "kernel_ms": round(wide["kernel_ms"], 4),
                "newton_iters_per_step": round(wide["iters"] / (wide["rollouts"] * KR), 3),
                "ls_halvings_per_step": round(wide["halvings"] / (wide["rollouts"] * KR), 3),
                "not_converged_or_diverged_trajectories": wide["bad"], "all_finite": wide["finite"],
                "note": "the initial-state ranges SURVEY.md 8(d) proposed, over the reference's %d steps.  The folded 3.2 m chain whips; the "
                        "reference algorithm itself (oracle) prints 'Newton diverged' within a few steps on 1-2 %% of these rollouts per step, "
                        "after which a rollout is not a valid simulation (DESIGN.md 5), so the headline uses U(-0.1,0.1); this is the same "
                        "launch on the wide states, failures counted, not hidden" % KR}
        if maxv is not None:
            from redmax_amd.scenes import MAX_VALID_INIT_AMPLITUDE as AMAX
            out["value_at_max_valid_init"] = {
                "init": "q,qdot~U(-%g,%g), rng(20240+global_index); traj 0: q=0.1,qdot=0" % (AMAX, AMAX), "newton_tol": args.tol, "steps": KR,
                "value": round(maxv["rollouts"] * KR / maxv["elapsed"], 1), "unit": "rollout-steps/s", "kernel_ms": round(maxv["kernel_ms"], 4),
                "newton_iters_per_step": round(maxv["iters"] / (maxv["rollouts"] * KR), 3),
                "ls_halvings_per_step": round(maxv["halvings"] / (maxv["rollouts"] * KR), 3),
                "not_converged_or_diverged_trajectories": maxv["bad"], "all_finite": maxv["finite"],
                "note": "the largest initial-state amplitude at which the REFERENCE ALGORITHM (literal oracle) survives all 1024 x 100 "
                        "trajectory-steps - found, not chosen: tools/max_valid_amplitude.py bisects [0.1, pi/4] on the full batch "
                        "(profiles/r04_max_valid_amplitude.json; 0.1963 already loses rollouts to 'Newton diverged' within 5 steps).  "
                        "GPU vs oracle at this amplitude: tests/test_gpu_full_size.py::test_max_valid_amplitude_sample"}
        if soft is not None:
            out["value_at_tol_1e-8"] = {
                "newton_tol": 1e-8, "steps": K, "value": round(soft["rollouts"] * K / soft["elapsed"], 1), "unit": "rollout-steps/s",
                "kernel_ms": round(soft["kernel_ms"], 4), "newton_iters_per_step": round(soft["iters"] / (soft["rollouts"] * K), 3),
                "not_converged_trajectories": soft["bad"],
                "note": "rounds 1 and 2 ran the headline at tol = 1e-8 (above the lattice spacing of g); kept for comparison with BENCH_r01/r02"}
        if cutleg is not None:
            out["value_with_ls_fail_limit_2"] = {
                "value": round(cutleg["rollouts"] * K / cutleg["elapsed"], 1), "unit": "rollout-steps/s", "kernel_ms": round(cutleg["kernel_ms"], 4),
                "newton_iters_per_step": round(cutleg["iters"] / (cutleg["rollouts"] * K), 3), "not_converged_trajectories": cutleg["bad"],
                "rollout_ms": cutleg.get("rollout_ms"),
                "note": "rmx_opts.ls_fail_limit = 2, the opt-in straggler policy (off by default, NOT the reference's loop): a step stops "
                        "iterating at its second failed line search instead of creeping on to iterMax = 320 iterations; rollouts it "
                        "does not touch are bit-identical (tests/test_gpu_straggler_policy.py)"}
        if strong is not None:
            out["note"] = ("METRIC-CONFORMANT FIGURE AT N > 1: strong_scaling.value (BASELINE.json's metric is a 1024-rollout batch: 1024 rollouts "
                           "in TOTAL over the %d ranks).  `value` is the weak-scaling figure the bench contract asks for: %d rollouts per GPU, a "
                           "%d-rollout job" % (world, B, m["rollouts"]))
            # the metric-conformant figure beside `value`, at the top level (round-4 review): BASELINE.json's 1024-rollout batch in total
            out["value_strong"] = round(strong["rollouts"] * K / strong["elapsed"], 1)
            out["kernel_ms_per_rank"] = {"weak": m["kernel_ms_per_rank"], "strong": strong["kernel_ms_per_rank"]}
            out["strong_scaling"] = {
                "global_batch": strong["rollouts"], "batch_per_gpu": strong["rollouts"] / world,
                "value": round(strong["rollouts"] * K / strong["elapsed"], 1), "unit": "rollout-steps/s",
                "ms_per_step": round(1e3 * strong["elapsed"] / K, 5), "kernel_ms": round(strong["kernel_ms"], 4),
                "repeat": strong.get("repeat"),
                "note": "BASELINE.json north_star reading: %d rollouts in TOTAL over the %d ranks.  One rollout is one wavefront and a "
                        "Newton step is a sequential chain, so below 1024 rollouts per GPU the kernel time does not shrink with the "
                        "batch (SIMDs idle): expect a flat curve (SURVEY.md §8(e))" % (strong["rollouts"], world)}
        if on_gpu and world == 1 and not args.no_cpu_baseline and wl == "chain":
            out.update(cpu_baselines(scene, args, h))
            # the numbers that decide SURVEY.md 8(d)'s Newton-count statement, at the top level of the line (round-5 review)
            nca = out.get("newton_count_agreement") or {}
            out["newton_count_agreement_frac"] = {k: (nca.get(k) or {}).get("frac") for k in ("vs_oracle", "vs_oracle_at_tol_1e-8", "vs_tensor_free")}
            out["newton_count_agreement_frac"]["oracle_vs_itself_1ulp_apart_at_tol_1e-9"] = ORACLE_SELF_AGREEMENT
        line = json.dumps(out)
        print(line, flush=True)
        if args.json_out:
            with open(args.json_out, "w") as f:
                f.write(line + "\n")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def cpu_baselines(scene, args, h):
    """The two CPU baselines on the host cores (OpenMP over rollouts, one rollout per thread) on a bounded sample of the same
    workload - the first `nb` rollouts, same Newton constants as the GPU run - plus the in-run parity numbers: max_b |q_gpu -
    q_oracle| / |q_oracle| and the Newton iteration count of every (rollout, step) of the sample, GPU vs CPU.
      cpu_baseline             kind "port": oracle/redmax_oracle.c, the literal restatement of the reference (O(n^3) tensor path)
      cpu_baseline_tensor_free oracle/redmax_tensorfree.c: the algorithm the GPU executes, scalar C
    Both CPU codes and the GPU are stepped ONE step per call here so that the counts can be compared per step (the CPU timings are
    the sums of those calls: the per-call overhead is microseconds against milliseconds of work)."""
    from oracle import oracle as orc
    from redmax_amd import BatchSim, syntheticStates
    cores = os.cpu_count() or 1
    nb = args.cpu_traj if args.cpu_traj > 0 else min(cores, 256, args.batch)     # one rollout per host thread
    cores = min(cores, nb)                  # threads actually used (OpenMP over rollouts)
    q, qd = syntheticStates(scene.nr, nb)
    desc = scene.desc()

    def gpu_counts(K, tol=None):
        sim = BatchSim(scene, batch=nb)
        sim.opts.tol = args.tol if tol is None else tol
        sim.opts.compensated = 0 if args.plain_iterate else 1
        sim.set_state(q, qd)
        per = np.zeros((K, nb), dtype=np.int64)
        for s in range(K):
            per[s] = sim.step_bdf1(1, h=h, stats=True)["newton_iters"]
        qg, _ = sim.get_state()
        sim.close()
        return per, qg

    # ---- literal port: size the sample to ~15 s (time 2 steps first), bounded by --cpu-steps
    orc.set_newton(tol=args.tol)
    qc, qdc = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
    t0 = time.perf_counter()
    orc.batch_step_bdf1(desc, qc, qdc, h, 2, nthreads=cores)
    per_step = (time.perf_counter() - t0) / 2
    ks = int(max(4, min(args.cpu_steps, 15.0 / max(per_step, 1e-6))))
    qc, qdc = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
    per_o = np.zeros((ks, nb), dtype=np.int64)
    dt = 0.0
    for s in range(ks):
        t0 = time.perf_counter()
        c = orc.batch_step_bdf1(desc, qc, qdc, h, 1, nthreads=cores, counters=True)
        dt += time.perf_counter() - t0
        per_o[s] = c["newton_iters"]
    orc.set_newton()
    # ---- Newton counts against the literal port above the lattice of doubles (tol 1e-8): at the reference's 1e-9 the literal port's last
    # iterations of a step wander over lattice points, so its counts are path dependent there
    per_o8 = per_g8 = None
    if args.tol < 1e-8:
        k8 = min(ks, 10)
        orc.set_newton(tol=1e-8)
        q8, qd8 = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
        per_o8 = np.zeros((k8, nb), dtype=np.int64)
        for s in range(k8):
            per_o8[s] = orc.batch_step_bdf1(desc, q8, qd8, h, 1, nthreads=cores, counters=True)["newton_iters"]
        orc.set_newton()
        per_g8, _ = gpu_counts(k8, tol=1e-8)
    # ---- tensor-free: the full 100 steps; whole-rollout calls repeated for ~5 s give the timing, one per-step pass the counts
    kt = 100
    reps, dtt = 0, 0.0
    while dtt < 5.0 and reps < 50:
        qt, qdt = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
        t0 = time.perf_counter()
        orc.tensorfree_batch_step_bdf1(desc, qt, qdt, h, kt, nthreads=cores, tol=args.tol, compensated=not args.plain_iterate)
        dtt += time.perf_counter() - t0
        reps += 1
    qt, qdt = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
    per_t = np.zeros((kt, nb), dtype=np.int64)
    for s in range(kt):
        per_t[s] = orc.tensorfree_batch_step_bdf1(desc, qt, qdt, h, 1, nthreads=cores, tol=args.tol, compensated=not args.plain_iterate)["newton_iters"]
    # ---- the GPU on the same sample, one step per launch
    per_g, qg100 = gpu_counts(kt)
    sim = BatchSim(scene, batch=nb)
    sim.opts.tol = args.tol
    sim.opts.compensated = 0 if args.plain_iterate else 1
    sim.set_state(q, qd)
    sim.step_bdf1(ks, h=h)
    qg, _ = sim.get_state()
    sim.close()
    err = float(np.max(np.linalg.norm(qg - qc, axis=1) / np.linalg.norm(qc, axis=1)))
    errt = float(np.max(np.linalg.norm(qg100 - qt, axis=1) / np.linalg.norm(qt, axis=1)))

    def agree(a, b):
        same = a == b
        return {"rollouts": nb, "steps": int(a.shape[0]), "trajectory_steps": int(a.size), "trajectory_steps_with_equal_count": int(same.sum()),
                "frac": round(float(same.mean()), 5), "rollouts_with_all_steps_equal": int(same.all(axis=0).sum()),
                "gpu_iters": int(a.sum()), "cpu_iters": int(b.sum()), "max_abs_diff_in_a_step": int(np.abs(a - b).max())}
    return {
        "cpu_baseline": {"value": round(nb * ks / dt, 2), "unit": "rollout-steps/s", "cores": cores, "kind": "port",
                         "sample": "first %d rollouts x %d steps of the same workload (oracle/redmax_oracle.c: literal restatement of the "
                                   "reference incl. its O(n^3) dJ/dq tensor path, OpenMP over rollouts, %.1f s); MATLAB is not available, "
                                   "the reference publishes no timing" % (nb, ks, dt)},
        "cpu_baseline_tensor_free": {"value": round(nb * kt * reps / dtt, 1), "unit": "rollout-steps/s", "cores": cores, "kind": "port",
                                     "sample": "first %d rollouts x %d steps x %d repetitions (oracle/redmax_tensorfree.c: the O(n^2) world-frame "
                                               "algorithm the GPU executes, scalar C -O2, same Newton, OpenMP over rollouts, %.1f s)" % (nb, kt, reps, dtt),
                                     "q_l2_relerr_gpu_vs_this_max": errt},
        "q_l2_relerr_vs_oracle_max": err,
        "newton_count_agreement": {
            "vs_oracle": agree(per_g[:ks], per_o), "vs_tensor_free": agree(per_g, per_t),
            "vs_oracle_at_tol_1e-8": agree(per_g8, per_o8) if per_o8 is not None else None,
            "note": "Newton iterations of every (rollout, step) of the sample, GPU vs CPU at the same tol, each side following its own "
                    "trajectory; SURVEY.md 8(d) expects identical counts on >= 99 % of trajectory-steps.  vs_tensor_free: the CPU twin of the "
                    "kernels' algorithm in the same iterate mode, at the run's tol.  vs_oracle: the literal port at the run's tol - at the "
                    "reference's 1e-9 its last iterations of a step wander over the lattice of doubles until |g| < tol is met (it needs MORE "
                    "iterations than the kernels: cpu_iters > gpu_iters), so the >= 99 % statement is made above the lattice, "
                    "vs_oracle_at_tol_1e-8.  Where counts differ otherwise, |g| landed within roundoff of tol at a convergence test"},
    }


def adjoint_main(args, ctx):
    """BASELINE.json configs[3]: adjoint BDF1 (forward + backward sweep) of the 16-DOF chain of scene 100's pattern, B rollouts
    with their own parameter vectors.  The one path of this library whose HBM traffic matters: the forward kernel stores
    H, M, D (3 n^2 doubles) per step and rollout, the backward kernel reads them back."""
    import torch
    from redmax_amd import BatchSim
    from redmax_amd.scenes import sceneAdjointChain
    n = 16
    sc = sceneAdjointChain(n)
    sc.init()
    B = args.batch if args.batch > 0 else 512
    K = args.steps
    rng = np.random.default_rng(20240 + ctx.rank)
    p = 1e-1 * rng.standard_normal((B, sc.nr))       # p ~ N(0, 1e-2)  
    task = dict(sc.task, t=K * sc.h)
    sim = BatchSim(sc, batch=B, device=ctx.device)
    q0, qd0 = sc.getQ()
    # inputs resident in HBM when the timed region starts (the bench contract): the initial state, the parameters and the result arrays
    # are device tensors and the call is rmx_adjoint_bdf1_device; the same job through the host-array entry (what a MATLAB fminunc
    # binds: p in, P / dPdp out over PCIe) is timed once beside it and reported as `value_host_arrays`
    dev = torch.device("cuda", ctx.device)
    q_t = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(q0[None, :], (B, sc.nr)))).to(dev)
    qd_t = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(qd0[None, :], (B, sc.nr)))).to(dev)
    p_t = torch.from_numpy(p).to(dev)
    P_t = torch.zeros(B, dtype=torch.float64, device=dev)
    dP_t = torch.zeros((B, sc.nr), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)

    def run(stats=True):
        # the job: forward + backward sweep from the initial state (resident in HBM, put in place ahead of the call) to P, dP/dp on
        # the device.  stats: also fetch the per-rollout Newton counters (diagnostics, two small device-to-host copies): the timed
        # call leaves them out, as the headline's timed region does, and takes them from the identical repeat that follows
        sim.set_state_device(q_t.data_ptr(), qd_t.data_ptr())
        if not stats:
            ctx.barrier()
            t0 = time.perf_counter()
        info = sim.adjoint_bdf1_device(K, sc.h, task, p_t.data_ptr(), P_t.data_ptr(), dP_t.data_ptr(), stats=stats)
        if not stats:
            ctx.barrier()
            info["elapsed"] = time.perf_counter() - t0
        return None, None, info

    def run_host():
        sim.set_state(q0[None, :], qd0[None, :])
        return sim.adjoint_bdf1(K, sc.h, task, p, stats=True)
    burn_ms, burned = 0.0, 0            # untimed launches of the same job: warm-up and clock ramp (see measure())
    while burned < max(1, min(args.warmup, 2)) or (burn_ms < args.burn_in and burned < 200):
        burn_ms += run()[2]["ms"]
        burned += 1
    run(stats=False)                    # (one untimed rehearsal of the timed form of the call: see measure())
    _, _, timed = run(stats=False)
    elapsed = ctx.max(timed["elapsed"])
    P, dPdp = P_t.cpu().numpy(), dP_t.cpu().numpy()
    ms = [timed["ms"]]
    info = run()[2]                     # the same job once more, with its counters
    ms.append(info["ms"])
    for _ in range(max(args.repeats - 1, 0)):
        ms.append(run()[2]["ms"])
    run_host()
    ctx.barrier()
    t0 = time.perf_counter()
    Ph, dPh, _ = run_host()
    ctx.barrier()
    elapsed_host = ctx.max(time.perf_counter() - t0)
    same_as_host = bool(np.array_equal(Ph, P) and np.array_equal(dPh, dPdp))
    sim.close()
    if ctx.rank == 0:
        kernel_ms = float(np.median(ms))
        iters = int(info["newton_iters"].sum())
        # algorithmic HBM bytes: forward writes H, M, D of the LAST iterate of every step (3 n^2 doubles; earlier iterates of a
        # step are overwritten in L2/HBM: counted once), backward reads H once, M twice (blocks k+1 and k+2), D once
        nn8 = n * n * 8
        alg = B * K * nn8 * (3 + 4)
        bad = int((info["status"] != 0).sum())
        out = {"metric": "sim steps/sec (whole node), 16-DOF chain adjoint BDF1 forward+backward", "value": round(ctx.world * B * K / elapsed, 1),
               "unit": "rollout-steps/s", "n_gpus": ctx.world, "steps": K, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / K, 5),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "adjoint BDF1 forward+backward (rmx_adjoint_bdf1), %d-DOF chain of scene 100's pattern, batch=%d per GPU, "
                                      "horizon %d steps (BASELINE.json configs[3])" % (n, B, K),
                          "batch_per_gpu": B, "links": n, "h": sc.h, "params": "p~N(0,1e-2), rng(20240+rank)",
                          "untimed_burn_in": {"ms": args.burn_in, "launches": burned},
                          "newton_iters_per_step": round(iters / (B * K), 3), "not_converged_trajectories": bad,
                          "all_finite": bool(np.isfinite(P).all() and np.isfinite(dPdp).all()),
                          "timed_region": "rmx_adjoint_bdf1_device (forward + backward kernel) from the initial state: state, p, P, dPdp "
                                          "resident in HBM when the region starts (rmx_set_state_device ahead of it); the per-rollout counters "
                                          "reported here are those of the identical untimed repeat that follows"},
               "value_host_arrays": {"value": round(ctx.world * B * K / elapsed_host, 1), "unit": "rollout-steps/s",
                                     "ms_per_step": round(1e3 * elapsed_host / K, 5), "same_bits_as_the_device_call": same_as_host,
                                     "note": "the same job through rmx_set_state + rmx_adjoint_bdf1: q, qdot, p in and P, dPdp out as "
                                             "host arrays (PCIe-inclusive; what an optimiser on the host binds)"},
               }
        # one Newton iteration of the line-search-free newton() (driverRedMaxAdjointBDF1.m:105-146) = one (g, H) evaluation + one solve;
        # per step on top: M, D assembled once (~2 x (66 n + 12 n(n+1)/2) flops), and in the backward sweep one transposed solve and
        # the M / D products (solve + 6 n^2)
        af = algorithm_flops("adjoint")
        per_step = 0.0
        if af is not None:
            per_step = 2 * (66 * n + 6 * n * (n + 1)) + 6 * n * n + json.load(open(ALGORITHM_FLOPS_FILE))["workloads"]["adjoint16"]["solve"]["flops"]
        rf = roofline(kernel_ms, float(iters), 0.0, B * K, K, args.warmup, B, n, "adjoint", 1e-9, info["newton_iters"], fronts=float(iters),
                      extra_useful=per_step * B * K)
        if rf is not None:
            rf["hbm"] = {"achieved_GBps": round(alg / (kernel_ms * 1e-3) / 1e9, 2), "frac_of_8TBps": round(alg / (kernel_ms * 1e-3) / 8e12, 5),
                         "algorithmic_bytes": alg,
                         "note": "the one path of the library with real HBM traffic, and still not bound by it: algorithmic bytes = B x K x n^2 x 8 x "
                                 "(3 written: H, M, D of each step + 4 read back: H, D once, M twice) per launch pair; the forward kernel "
                                 "rewrites a step's H, M, D once per Newton iteration (%.2f per step)" % (iters / (B * K))}
            rf["kernel_ms_all"] = [round(x, 4) for x in ms]
        out["roofline"] = rf
        line = json.dumps(out)
        print(line, flush=True)
        if args.json_out:
            with open(args.json_out, "w") as f:
                f.write(line + "\n")
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, argv)
    return rank_main(args)


if __name__ == "__main__":
    sys.exit(main())
