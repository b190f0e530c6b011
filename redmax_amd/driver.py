"""Entry points with the reference's names and signatures.

``driverRedMaxBDF1(sceneID, batch)`` / ``driverRedMaxBDF2(sceneID, batch)`` mirror
matlab-diff/driverRedMaxBDF1.m:1-54 and driverRedMaxBDF2.m: build the scene, ``init()``, run
``simLoop`` and check the final energy against ``Hexpected`` (Scene.plotEnergies, Scene.m:164-178).
``simLoop`` is ONE call into the HIP library (all steps run on the device); ``batch`` keeps the
reference's meaning (non-interactive: no drawing, no FD self-tests - which is all this build does).
"""
from __future__ import annotations

from .batch import BatchSim
from .scenes import scenesRedMax


def simLoop(scene, itype=1, device=0):
    """driverRedMaxBDF1.m:57-91 / driverRedMaxBDF2.m:57-125 for a single trajectory."""
    sim = BatchSim(scene, batch=1, device=device)
    q0, qd0 = scene.getQ()
    sim.set_state(q0[None, :], qd0[None, :])
    T0, V0 = sim.energy()                      # Scene.reset: T0, V0 at the initial state (Scene.m:126-127)
    scene.T0, scene.V0 = float(T0[0]), float(V0[0])
    step = sim.step_bdf1 if itype == 1 else sim.step_bdf2
    out = step(scene.nsteps, h=scene.h, stats=True, history="full")
    q, qd = sim.get_state()
    scene.setQ(q[0], qd[0])
    # JointSpherical / JointFree3D keep chart and q together (JointSpherical.m:28-34, 63-102): the final q is expressed in
    # the charts the device ended in, so they go back onto the joints with it.  The per-step history q is in the chart
    # that was current after that step: recorded per step as history[k]["charts"] (the reference's own history has a TODO
    # for it, Scene.m:138); status bit RMX_ST_CHART says whether any switch happened at all.
    if sim.nsph:
        charts = sim.charts()[0]
        sph = [j for j in scene.joints if hasattr(j, "chart")]
        for j, c in zip(sph, charts):
            j.chart = int(c)
    scene.history = []
    for k in range(scene.nsteps):              # Scene.saveHistory (Scene.m:134-161): the record of every step
        scene.t = (k + 1) * scene.h
        scene.k = k + 1
        scene.history.append({"t": scene.t, "T": float(out["T"][k, 0]), "V": float(out["V"][k, 0]),
                              "q": out["q"][k, 0].copy(), "qdot": out["qdot"][k, 0].copy(), "charts": out["charts"][k, 0].copy()})
    scene.solverInfo = {"newton_iters": int(out["newton_iters"][0]), "ls_halvings": int(out["ls_halvings"][0]),
                        "status": int(out["status"][0]), "kernel_ms": out["ms"]}
    sim.close()
    return scene


def _driver(sceneID, batch, itype, device, verbose):
    scene = scenesRedMax(sceneID)
    scene.init()
    if not batch:
        raise NotImplementedError("interactive mode (Scene.test / Scene.draw) is out of scope; call with batch=True")
    if verbose:
        print("(%d) '%s': tEnd=%.1f, nsteps=%d, nr=%d, nm=%d" % (sceneID, scene.name, scene.tEnd, scene.nsteps, scene.countR(), scene.countM()))
    simLoop(scene, itype, device)
    H, passed = scene.plotEnergies(itype, verbose=verbose)
    return scene, H, passed


def driverRedMaxBDF1(sceneID=0, batch=True, device=0, verbose=True):
    return _driver(sceneID, batch, 1, device, verbose)


def driverRedMaxBDF2(sceneID=0, batch=True, device=0, verbose=True):
    return _driver(sceneID, batch, 2, device, verbose)


def testRedMax(sceneID=0, device=0, verbose=True):
    """matlab-simple/testRedMax.m:1-41 (BASELINE.json configs[0]): scene, init, euler() over tspan=[0 2] with
    hEuler=1e-2 (matlab-simple/+redmax/Scene.m:22-23), then the energy check of matlab/testRedMax.m:164-177 against
    Hexpected(REDMAX_EULER) (matlab/testRedMaxScenes.m:39,67,93 - same scenes, h, tspan and gravity)."""
    Hexpected = {0: -5930.8171118834870867, 1: -9423.2594023734018265, 2: -1123.9825362491046690}
    scene = scenesRedMax(sceneID)
    scene.init()
    h, nsteps = 1e-2, 200
    sim = BatchSim(scene, batch=1, device=device)
    q0, qd0 = scene.getQ()
    sim.set_state(q0[None, :], qd0[None, :])
    _, V0 = sim.energy()
    out = sim.step_euler(nsteps, h, history=True)
    q, qd = sim.get_state()
    scene.setQ(q[0], qd[0])
    H = float(out["T"][-1, 0] + out["V"][-1, 0] - V0[0])
    passed = None
    if sceneID in Hexpected:
        passed = bool(abs(H - Hexpected[sceneID]) <= 1e-2)
        if verbose:
            print("### PASS ###" if passed else "### FAIL: %.16f ###" % H)
    sim.close()
    return scene, H, passed


def taskObjective(p, scene, sim=None, device=0, bdf2=False):
    """taskObjective of driverRedMaxAdjointBDF1.m:39-62 (bdf2: of driverRedMaxAdjointBDF2.m:38-62): scene.reset(), forward simLoop
    under the task parameters, task.calcFinal() -> (P, dPdp).  `p` may be [nr] (one rollout) or [B][nr] (a batch of parameter vectors, each its own
    rollout).  Forward and backward sweeps are two kernel launches inside rmx_adjoint_bdf1."""
    import numpy as np
    p = np.atleast_2d(np.asarray(p, dtype=np.float64))
    own = sim is None
    if own:
        sim = BatchSim(scene, batch=p.shape[0], device=device)
    q0, qd0 = scene.qInit, scene.qdotInit                       # Scene.reset (Scene.m:122-131)
    sim.set_state(np.broadcast_to(q0, (sim.B, scene.nr)), np.broadcast_to(qd0, (sim.B, scene.nr)))
    P, dPdp, info = (sim.adjoint_bdf2 if bdf2 else sim.adjoint_bdf1)(scene.nsteps, scene.h, scene.task, p, stats=True)
    if own:
        sim.close()
    return P, dPdp, info


def driverRedMaxAdjointBDF1(nlinks=2, device=0, verbose=True, maxiter=50, bdf2=False):
    """driverRedMaxAdjointBDF1.m:1-36 with scene 100 (or its n-link generalisation): minimise the task objective over the
    joint torques.  MATLAB's fminunc (quasi-Newton, gradient supplied) is replaced by scipy's BFGS with the same
    objective/gradient callback."""
    import numpy as np
    from scipy.optimize import minimize
    from .scenes import sceneAdjointChain
    scene = sceneAdjointChain(nlinks, bdf2=bdf2)
    scene.init()
    sim = BatchSim(scene, batch=1, device=device)

    def fun(p):
        P, dPdp, _ = taskObjective(p, scene, sim=sim, bdf2=bdf2)
        return float(P[0]), dPdp[0]

    res = minimize(fun, np.zeros(scene.nr), jac=True, method="BFGS", options={"maxiter": maxiter, "disp": verbose})
    sim.close()
    return scene, res


def driverRedMaxAdjointBDF2(nlinks=2, device=0, verbose=True, maxiter=50):
    """driverRedMaxAdjointBDF2.m:1-36 with scene 101 (TaskBDF2PointPos): the same optimisation over the SDIRK2 + BDF2 rollout."""
    return driverRedMaxAdjointBDF1(nlinks, device, verbose, maxiter, bdf2=True)


def _main(argv=None):
    """python -m redmax_amd.driver [--bdf2] [sceneID ...]: driverRedMaxBDF1(sceneID,true) / driverRedMaxBDF2(sceneID,true) for the
    listed scenes (default: every scene this build covers), printing the reference's '### PASS ###' line per scene."""
    import argparse
    ap = argparse.ArgumentParser(prog="python -m redmax_amd.driver", description=_main.__doc__)
    ap.add_argument("scenes", nargs="*", type=int, default=[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 14])
    ap.add_argument("--bdf2", action="store_true", help="driverRedMaxBDF2 instead of driverRedMaxBDF1")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    ok = True
    for sid in a.scenes:
        scene, H, passed = (driverRedMaxBDF2 if a.bdf2 else driverRedMaxBDF1)(sid, True, a.device, True)
        print("    H = %.16e  (kernel %.3f ms, %d Newton iterations, status %d)" % (
            H, scene.solverInfo["kernel_ms"], scene.solverInfo["newton_iters"], scene.solverInfo["status"]))
        ok = ok and passed is not False
    return 0 if ok else 1


if __name__ == "__main__":
    import sys
    sys.exit(_main())
