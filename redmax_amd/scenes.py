"""Test scenes: the in-scope cases of matlab-diff/scenesRedMax.m plus the synthetic
benchmark configurations of BASELINE.json / SURVEY.md §8(d).

``scenesRedMax(sceneID)`` keeps the reference's entry-point name and scene numbering
(scenesRedMax.m:1-13): 0 simple serial chain (:52-79), 1 different revolute axes (:80-100),
2 branching (:101-130), 3 prismatic joint (:131-144), 14 joint limits (:371-401).  The
golden energies ``Hexpected(BDF1/BDF2)`` are the reference's own known answers.
"""
from __future__ import annotations

import math

import numpy as np

from . import se3
from .redmax import (BodyCuboid, ForceGroundCuboid, JointFixed, JointFree2D, JointFree3D, JointPlanar, JointPrismatic,
                     JointRevolute, JointSpherical, JointTranslational, JointUniversal, Scene)

BDF1 = 1
BDF2 = 2


def _T(p):
    return se3.transform(p=p)


def scenesRedMax(sceneID):
    scene = Scene()
    density = 1.0
    if sceneID == 0:
        scene.name = "Simple serial chain"
        scene.Hexpected[BDF1 - 1] = -1.2705398823489915e05   # scenesRedMax.m:54
        scene.Hexpected[BDF2 - 1] = 2.6058008179021417e03    # :55
        sides = [10, 1, 1]
        nbodies = 5
        for i in range(1, nbodies + 1):
            scene.bodies.append(BodyCuboid(density, sides))
            if i == 1:
                scene.joints.append(JointRevolute(None, scene.bodies[-1], [0, 1, 0]))
                scene.joints[-1].setJointTransform(np.eye(4))
            else:
                if i % 2 == 1:
                    scene.joints.append(JointRevolute(scene.joints[i - 2], scene.bodies[-1], [0, 1, 0]))
                else:
                    scene.joints.append(JointFixed(scene.joints[i - 2], scene.bodies[-1]))
                scene.joints[-1].setJointTransform(_T([10, 0, 0]))
            scene.bodies[-1].setBodyTransform(_T([5, 0, 0]))
            scene.joints[-1].q[0] = math.pi / 4 if i % 2 == 1 else 0.0
    elif sceneID == 1:
        scene.name = "Different revolute axes"
        scene.Hexpected[BDF1 - 1] = -3.8359074258588909e04   # :82
        scene.Hexpected[BDF2 - 1] = -9.7138545812971279e02   # :83
        sides = [10, 1, 1]
        b = [BodyCuboid(density, sides) for _ in range(3)]
        scene.bodies = b
        j1 = JointRevolute(None, b[0], [0, 0, 1])
        j2 = JointRevolute(j1, b[1], [0, 1, 0])
        j3 = JointRevolute(j2, b[2], [0, 0, 1])
        scene.joints = [j1, j2, j3]
        for bb in b:
            bb.setBodyTransform(_T([5, 0, 0]))
        j1.setJointTransform(np.eye(4))
        j2.setJointTransform(_T([10, 0, 0]))
        j3.setJointTransform(_T([10, 0, 0]))
        j1.q[0] = 0.0
        j2.q[0] = math.pi / 2
        j3.q[0] = math.pi / 2
    elif sceneID == 2:
        scene.name = "Branching"
        scene.Hexpected[BDF1 - 1] = -2.2826101928480086e04   # :108
        scene.Hexpected[BDF2 - 1] = -2.4159349151742754e02   # :109
        b = [BodyCuboid(density, [1, 1, 10]), BodyCuboid(density, [1, 20, 1]),
             BodyCuboid(density, [1, 1, 10]), BodyCuboid(density, [1, 1, 10])]
        scene.bodies = b
        j1 = JointRevolute(None, b[0], [1, 0, 0])
        j2 = JointRevolute(j1, b[1], [0, 0, 1])
        j3 = JointRevolute(j2, b[2], [1, 0, 0])
        j4 = JointRevolute(j2, b[3], [0, 1, 0])
        scene.joints = [j1, j2, j3, j4]
        b[0].setBodyTransform(_T([0, 0, -5]))
        b[1].setBodyTransform(_T([0, 0, 0]))
        b[2].setBodyTransform(_T([0, 0, -5]))
        b[3].setBodyTransform(_T([0, 0, -5]))
        j1.setJointTransform(_T([0, 0, 15]))
        j2.setJointTransform(_T([0, 0, -10]))
        j3.setJointTransform(_T([0, -10, 0]))
        j4.setJointTransform(_T([0, 10, 0]))
        j3.q[0] = math.pi / 4
        j4.q[0] = math.pi / 4
    elif sceneID == 3:
        scene.name = "Prismatic joint"
        scene.Hexpected[BDF1 - 1] = -3.7579402399569808e04   # :133
        scene.Hexpected[BDF2 - 1] = -6.1132876082600706e02   # :134
        b1 = BodyCuboid(density, [20, 1, 1])
        j1 = JointPrismatic(None, b1, [1, 0, 0])
        j1.setJointTransform(np.eye(4))
        b1.setBodyTransform(np.eye(4))
        b2 = BodyCuboid(density, [1, 1, 10])
        j2 = JointRevolute(j1, b2, [0, 1, 0])
        j2.setJointTransform(_T([-10, 0, 0]))
        b2.setBodyTransform(_T([0, 0, -5]))
        j2.q[0] = math.pi / 2
        scene.bodies = [b1, b2]
        scene.joints = [j1, j2]
    elif sceneID == 14:
        scene.name = "Joint limits"
        scene.Hexpected[BDF1 - 1] = -2.5928305306546572e04   # :373
        scene.Hexpected[BDF2 - 1] = -1.8476279319765570e04   # :374
        scene.h = 5e-3
        sides = [10, 1, 1]
        nbodies = 3
        for i in range(1, nbodies + 1):
            scene.bodies.append(BodyCuboid(density, sides))
            if i == 1:
                scene.joints.append(JointRevolute(None, scene.bodies[-1], [0, 1, 0]))
                scene.joints[-1].setJointTransform(se3.transform(R=se3.aaToMat([0, 1, 0], math.pi / 2)))
                scene.joints[-1].q[0] = 0.0
            else:
                scene.joints.append(JointRevolute(scene.joints[i - 2], scene.bodies[-1], [0, 1, 0]))
                scene.joints[-1].setJointTransform(_T([10, 0, 0]))
                scene.joints[-1].q[0] = -math.pi / 6
            scene.bodies[-1].setBodyTransform(_T([5, 0, 0]))
            j = scene.joints[-1]
            j.setLimitLower(-math.pi / 2)
            j.setLimitUpper(0.0)
            j.setLimitStiffness(1e5)
            j.setLimitDamping(1e2)
            j.setDamping(1e2)
    elif sceneID == 4 or sceneID == 5:
        if sceneID == 4:
            scene.name = "Planar joint"
            scene.Hexpected[BDF1 - 1] = -4.5738939646068720e04   # scenesRedMax.m:147
            scene.Hexpected[BDF2 - 1] = -4.7000178355609387e02   # :148
        else:
            scene.name = "Translational joint"
            scene.Hexpected[BDF1 - 1] = 3.3661704151378050e04    # :166
            scene.Hexpected[BDF2 - 1] = 3.3377464890219308e04    # :167
            scene.tEnd = 2.0
            scene.grav = np.zeros(3)
        b = [BodyCuboid(density, [10, 10, 1]), BodyCuboid(density, [1, 1, 10]), BodyCuboid(density, [1, 1, 10])]
        scene.bodies = b
        j1 = JointPlanar(None, b[0]) if sceneID == 4 else JointTranslational(None, b[0])
        j2 = JointRevolute(j1, b[1], [0, 1, 0])
        j3 = JointRevolute(j1, b[2], [1, 0, 0])
        scene.joints = [j1, j2, j3]
        j1.setJointTransform(np.eye(4))
        j2.setJointTransform(_T([-5, 0, 0]))
        j3.setJointTransform(_T([0, -5, 0]))
        b[0].setBodyTransform(np.eye(4))
        b[1].setBodyTransform(_T([0, 0, -5]))
        b[2].setBodyTransform(_T([0, 0, -5]))
        if sceneID == 4:
            j2.q[0] = math.pi / 2                                # :162-163
            j3.q[0] = math.pi / 4
        else:
            j2.qdot[0] = -10.0                                   # :182-185
            j3.qdot[0] = 10.0
    elif sceneID == 6:
        scene.name = "Free2D joint"
        scene.Hexpected[BDF1 - 1] = 2.0322933333333378e04        # :189
        scene.Hexpected[BDF2 - 1] = 2.1283333333333332e04        # :190
        scene.h = 5e-3
        scene.tEnd = 0.4
        scene.grav = np.array([0.0, -980.0, 0.0])
        b = BodyCuboid(density, [1, 1, 1])
        j = JointFree2D(None, b)
        j.q[:] = [-10.0, -10.0, 0.0]
        j.qdot[:] = [50.0, 200.0, 20.0]
        j.setJointTransform(np.eye(4))
        b.setBodyTransform(np.eye(4))
        scene.bodies, scene.joints = [b], [j]
    elif sceneID == 7:
        scene.name = "Spherical joint"
        scene.Hexpected[BDF1 - 1] = -8.7859815791305155e03       # scenesRedMax.m:206
        scene.Hexpected[BDF2 - 1] = 8.6544602745403390e03        # :207 (this run switches Euler charts)
        scene.tEnd = 1.0
        scene.h = 2e-3
        b = [BodyCuboid(density, [1, 1, 10]), BodyCuboid(density, [1, 1, 10])]
        j1 = JointSpherical(None, b[0])
        j1.setJointTransform(np.eye(4))
        j1.q[:] = JointSpherical.getEulerInv(j1.chart, se3.aaToMat([1, 0, 0], math.pi / 8))   # :217
        j1.qdot[:] = [2.0, 2.0, 2.0]
        j2 = JointSpherical(j1, b[1])
        j2.setJointTransform(_T([0, 0, -10]))
        j2.q[0] = math.pi / 2
        for bb in b:
            bb.setBodyTransform(_T([0, 0, -5]))
        scene.bodies, scene.joints = b, [j1, j2]
    elif sceneID == 9:
        scene.name = "Free3D joint"
        scene.Hexpected[BDF1 - 1] = 4.3970920953724946e00        # :250
        scene.Hexpected[BDF2 - 1] = 4.5466508559364156e00        # :251
        scene.h = 5e-2
        scene.tEnd = 6.0
        scene.grav = np.array([0.0, 0.0, -1.0])
        b = BodyCuboid(density, [1, 1, 1])
        j = JointFree3D(None, b)
        j.qdot[:] = [0.0, 0.0, 3.0, 0.2, 0.4, 0.6]
        j.setJointTransform(np.eye(4))
        b.setBodyTransform(np.eye(4))
        scene.bodies, scene.joints = [b], [j]
    elif sceneID == 8:
        scene.name = "Universal joint"
        scene.Hexpected[BDF1 - 1] = -2.5276246935781084e04       # :230
        scene.Hexpected[BDF2 - 1] = -1.3781281283808785e03       # :231
        for i in range(1, 4):
            scene.bodies.append(BodyCuboid(density, [1, 1, 10]))
            j = JointUniversal(scene.joints[-1] if i > 1 else None, scene.bodies[-1])
            j.setJointTransform(np.eye(4) if i == 1 else _T([0, 0, -10]))
            scene.bodies[-1].setBodyTransform(_T([0, 0, -5]))
            j.q[0 if i % 2 == 1 else 1] = math.pi / 8            # :242-246
            scene.joints.append(j)
    elif sceneID == 11:
        # scenesRedMax.m:290-311 'Free2D with ground': pins ForceGroundCuboid through Hexpected
        scene.name = "Free2D with ground"
        scene.Hexpected[BDF1 - 1] = -4.4208045000000002e03    # :292 (the reference notes BDF1 "doesn't work" for this scene)
        scene.Hexpected[BDF2 - 1] = -2.7811251900394832e03    # :293
        scene.h = 5e-4
        scene.tEnd = 0.6
        scene.grav = np.array([0.0, -980.0, 0.0])
        b = BodyCuboid(density, [3, 1, 1])
        j = JointFree2D(None, b)
        j.setJointTransform(np.eye(4))
        b.setBodyTransform(np.eye(4))
        j.q[:] = [-1.0, 2.0, 0.0]
        j.qdot[:] = [5.0, 70.0, 2.0]
        scene.bodies = [b]
        scene.joints = [j]
        f = ForceGroundCuboid(b)
        f.setTransform(se3.transform(R=se3.aaToMat([1, 0, 0], -math.pi / 2)))
        f.setStiffness(1e5, 1e2)
        f.setDamping(3e1)
        f.setFriction(0.5)
        scene.forces = [f]
    elif sceneID == 100:
        return sceneAdjointChain(2)                            # scenesRedMax.m:402-436 ('Adjoint BDF1')
    elif sceneID == 101:
        return sceneAdjointChain(2, bdf2=True)                 # scenesRedMax.m:437-471 ('Adjoint BDF2')
    else:
        raise ValueError("scene %r is out of scope (needs joint/force types outside SURVEY.md §8)" % (sceneID,))
    return scene


IN_SCOPE_SCENES = (0, 1, 2, 3, 14)          # 0/1-DOF joints only
COMPOSITE_SCENES = (4, 5, 6, 8)             # JointPlanar / Translational / Free2D / Universal (lowered to 1-DOF chains)
SPHERICAL_SCENES = (7, 9)                   # JointSpherical / JointFree3D (Euler charts with switching)


def sceneAdjointChain(n=2, bdf2=False):
    """Scene 100 'Adjoint BDF1' (scenesRedMax.m:402-436) generalised to n links (BASELINE.json configs[3] uses n=16):
    revolute y-axis chain of [10 1 1] cuboids, q = pi/2 at the root and pi/4 elsewhere, qdot = 1, joint stiffness and
    damping 1e4, and a TaskBDF1PointPos on the last body (point [5 0 0], target [10 0 -10], pscale 1e5, weights 1e-2/1e2,
    measured at tEnd).  bdf2: scene 101 'Adjoint BDF2' (:437-471), the same chain with a TaskBDF2PointPos whose target is
    [-10 0 -10]."""
    scene = Scene()
    tag = "Adjoint BDF2" if bdf2 else "Adjoint BDF1"
    scene.name = tag if n == 2 else "%s, %d links" % (tag, n)
    for i in range(n):
        scene.bodies.append(BodyCuboid(1.0, [10, 1, 1]))
        parent = scene.joints[i - 1] if i else None
        j = JointRevolute(parent, scene.bodies[-1], [0, 1, 0])
        j.setJointTransform(np.eye(4) if i == 0 else _T([10, 0, 0]))
        j.q[0] = math.pi / 2 if i == 0 else math.pi / 4
        j.qdot[0] = 1.0
        j.setStiffness(1e4)
        j.setDamping(1e4)
        scene.joints.append(j)
        scene.bodies[-1].setBodyTransform(_T([5, 0, 0]))
    scene.task = {"body": n - 1, "xlocal": [5.0, 0.0, 0.0], "xtarget": [-10.0 if bdf2 else 10.0, 0.0, -10.0], "t": scene.tEnd,
                  "pscale": 1e5, "wreg": 1e-2, "wpos": 1e2}
    return scene


def sceneChain(n=32, axis=(0, 1, 0), q0=0.0):
    """Config 2 (north star): n-link serial revolute chain, the pattern of scenesRedMax.m:52-79 with
    every joint revolute: cuboid(density 1, sides [10 1 1]), joint 1 at identity, joints 2..n at
    [10 0 0], bodies at [5 0 0]."""
    scene = Scene()
    scene.name = "%d-link serial revolute chain" % n
    for i in range(n):
        scene.bodies.append(BodyCuboid(1.0, [10, 1, 1]))
        parent = scene.joints[i - 1] if i else None
        scene.joints.append(JointRevolute(parent, scene.bodies[-1], axis))
        scene.joints[-1].setJointTransform(np.eye(4) if i == 0 else _T([10, 0, 0]))
        scene.bodies[-1].setBodyTransform(_T([5, 0, 0]))
        scene.joints[-1].q[0] = q0
    return scene


def sceneChainGround(n=32, ground_z=-2.0, q0=0.0):
    """Config 5: the n-link revolute chain of config 2 over a frictional ground plane (ForceGroundCuboid on every body,
    the stiffness / damping / friction values of scenesRedMax.m:305-309), BDF2 at scene 11's step h = 5e-4
    (scenesRedMax.m:295).  The ground frame is z-up at height ground_z, so the chain swings down into it."""
    scene = sceneChain(n, q0=q0)
    scene.name = "%d-link chain over frictional ground" % n
    scene.h = 5e-4
    scene.tEnd = 0.1
    for b in scene.bodies:
        f = ForceGroundCuboid(b)
        f.setTransform(_T([0, 0, ground_z]))
        f.setStiffness(1e5, 1e2)
        f.setDamping(3e1)
        f.setFriction(0.5)
        scene.forces.append(f)
    return scene


def sceneChainFloorAndWall(n=6, ground_z=-1.0, wall_x=None):
    """Every body of the chain carries TWO ForceGroundCuboid objects - the floor of sceneChainGround and a wall, a plane whose normal is
    the world x axis, softer and with less friction - which the reference allows: its forces are a list (Force.m:26-56, scenesRedMax.m:303
    `scene.forces{end+1} = ...`), nothing ties a body to one.  wall_x: where the wall stands (default: under the chain's first link, so
    that the root-side bodies start inside it)."""
    scene = sceneChainGround(n, ground_z=ground_z)
    scene.name = "%d-link chain over a floor and against a wall" % n
    if wall_x is None:
        wall_x = 4.0
    wall = np.array([[0.0, 0.0, 1.0, wall_x], [0.0, 1.0, 0.0, 0.0], [-1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])      # Z axis of the frame = +x
    for b in list(scene.bodies):
        f = ForceGroundCuboid(b)
        f.setTransform(wall)
        f.setStiffness(4e4, 5e1)
        f.setDamping(2e1)
        f.setFriction(0.2)
        scene.forces.append(f)
    return scene


def sceneChainTwoGrounds(n=8, ground_z=-1.0):
    """sceneChainGround with TWO kinds of ForceGroundCuboid objects, which the reference allows (every object holds its own E, kn, kt,
    mu, kd: ForceGroundCuboid.m:6-13): even bodies over the z-up floor at ground_z with scene 11's constants, odd bodies over a plane
    tilted by 0.3 rad about y through (0, 0, ground_z - 0.5), softer, more damped and nearly frictionless."""
    scene = sceneChain(n)
    scene.name = "%d-link chain over two grounds" % n
    scene.h = 5e-4
    scene.tEnd = 0.1
    c, s = math.cos(0.3), math.sin(0.3)
    tilted = np.array([[c, 0, s, 0.0], [0, 1, 0, 0.0], [-s, 0, c, ground_z - 0.5], [0, 0, 0, 1.0]])
    for i, b in enumerate(scene.bodies):
        f = ForceGroundCuboid(b)
        if i % 2 == 0:
            f.setTransform(_T([0, 0, ground_z]))
            f.setStiffness(1e5, 1e2)
            f.setDamping(3e1)
            f.setFriction(0.5)
        else:
            f.setTransform(tilted)
            f.setStiffness(4e4, 5e1)
            f.setDamping(6e1)
            f.setFriction(0.05)
        scene.forces.append(f)
    return scene


def sceneTree(n=64):
    """Config 3: ~n-DOF branching tree following scene 2's pattern (scenesRedMax.m:101-130):
    binary tree in depth-first order, 1-DOF joints alternating by depth parity between revolute
    (axes cycling x,y,z) and prismatic (axis x), same [10 1 1] cuboids."""
    scene = Scene()
    scene.name = "%d-joint branching tree" % n
    axes = ([1, 0, 0], [0, 1, 0], [0, 0, 1])
    depth_limit = int(math.floor(math.log2(n + 1)))  # full binary levels, remainder hangs as a chain

    state = {"count": 0, "rev": 0}

    def add(parent, depth, offset):
        if state["count"] >= n:
            return None
        body = BodyCuboid(1.0, [10, 1, 1])
        if depth % 2 == 0:
            ax = axes[state["rev"] % 3]
            state["rev"] += 1
            joint = JointRevolute(parent, body, ax)
            joint.q[0] = 0.1
        else:
            joint = JointPrismatic(parent, body, [1, 0, 0])
            joint.q[0] = 0.0
            joint.setStiffness(1e4)   # keep the slider bounded
        joint.setJointTransform(np.eye(4) if parent is None else _T(offset))
        body.setBodyTransform(_T([5, 0, 0]))
        scene.bodies.append(body)
        scene.joints.append(joint)
        state["count"] += 1
        if depth + 1 < depth_limit:
            add(joint, depth + 1, [10, -3, 0])
            add(joint, depth + 1, [10, 3, 0])
        return joint

    add(None, 0, [0, 0, 0])
    # hang whatever is left as a chain below the last joint so that exactly n joints exist
    while state["count"] < n:
        parent = scene.joints[-1]
        body = BodyCuboid(1.0, [10, 1, 1])
        joint = JointRevolute(parent, body, axes[state["rev"] % 3])
        state["rev"] += 1
        joint.setJointTransform(_T([10, 0, 0]))
        joint.q[0] = 0.1
        body.setBodyTransform(_T([5, 0, 0]))
        scene.bodies.append(body)
        scene.joints.append(joint)
        state["count"] += 1
    return scene


# The largest a for which the reference algorithm itself (literal oracle, the reference's Newton constants) survives all 1024 x 100
# trajectory-steps of the 32-link chain from q, qdot ~ U(-a, a): tools/max_valid_amplitude.py, profiles/r04_max_valid_amplitude.json
# (0.1963 already loses rollouts to "Newton diverged" within 5 steps).  bench.py's side leg value_at_max_valid_init runs there.
MAX_VALID_INIT_AMPLITUDE = 0.1856


def syntheticStates(nr, batch, first=0, sq=0.1, sv=0.1):
    """Synthetic initial states of the benchmark configs: trajectory b has q~U(-sq,sq)^nr, qdot~U(-sv,sv)^nr
    from numpy.random.default_rng(20240+b); trajectory 0 is the deterministic state q=0.1, qdot=0 used in
    SURVEY.md §4.  Seeds depend on the GLOBAL trajectory index so results are shard-invariant (§8(e)).

    SURVEY.md §8(d) proposed sq=pi/4, sv=1; with those the reference's own Newton (the oracle, literal
    restatement) stalls in a local minimum of |g| and reports "Newton diverged" within a few steps (the folded
    32-link chain whips violently), so the rollouts are not valid simulations.  sq=sv=0.1 keeps every one of the
    1024 x 100 trajectory-steps convergent (measured; see DESIGN.md "Workload")."""
    q = np.empty((batch, nr))
    qd = np.empty((batch, nr))
    for i in range(batch):
        b = first + i
        if b == 0:
            q[i] = 0.1
            qd[i] = 0.0
        else:
            rng = np.random.default_rng(20240 + b)
            q[i] = rng.uniform(-sq, sq, nr)
            qd[i] = rng.uniform(-sv, sv, nr)
    return q, qd
