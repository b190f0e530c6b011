"""Host-side mirror of the reference's ``+redmax`` class surface for the BDF1/BDF2 path.

The reference builds a scene out of handle objects (matlab-diff/scenesRedMax.m):
``BodyCuboid(density, sides)``, ``JointRevolute(parent, body, axis)``,
``setJointTransform``, ``setBodyTransform``, ``q``, ``qdot`` ... and ``Scene.init()``
links them and numbers the DOFs leaf-to-root (matlab-diff/+redmax/Scene.m:59-119).
The same construction API is kept here so scene files read like the reference's; what
changes is that ``Scene.init()`` flattens the tree into the POD model descriptor of
``include/redmax_hip.h`` and every numerical method (update / computeJacobian /
computeMassGrav / newton ...) is replaced by the HIP kernels behind the C ABI.

Only what the path needs is mirrored: Body/BodyCuboid (Body.m, BodyCuboid.m),
Joint + JointRevolute/JointPrismatic/JointFixed (Joint.m, JointRevolute.m,
JointPrismatic.m, JointFixed.m), the multi-DOF joints whose motion is a product of one-parameter motions
(JointPlanar.m, JointTranslational.m, JointUniversal.m, JointFree2D.m; rmx_model_create lowers them to chains of
1-DOF nodes with massless links), ForceGroundCuboid and Scene (Scene.m init/reset/saveHistory/plotEnergies).
Drawing, FD self-tests, JointSpherical/JointFree3D and the other force types are out of scope.
"""
from __future__ import annotations

import math

import numpy as np

from . import se3

JOINT_FIXED = 0
JOINT_REVOLUTE = 1
JOINT_PRISMATIC = 2
JOINT_PLANAR = 3
JOINT_TRANSLATIONAL = 4
JOINT_UNIVERSAL = 5
JOINT_FREE2D = 6
JOINT_SPHERICAL = 7
JOINT_FREE3D = 8


class Body:
    """Rigid body attached to a joint (matlab-diff/+redmax/Body.m:24-51)."""

    _count = 1

    def __init__(self, density):
        self.name = "body%d" % Body._count
        Body._count += 1
        self.density = float(density)
        self.damping = 0.0
        self.I_i = np.ones(6)
        self.E0_ji = np.eye(4)
        self.joint = None
        self.idxM = None

    def setBodyTransform(self, E):
        """Transform of this body wrt its joint (Body.m:46-51)."""
        self.E0_ji = np.array(E, dtype=np.float64).reshape(4, 4)

    def computeInertia(self):
        self.computeInertia_()

    def computeInertia_(self):  # pragma: no cover - abstract
        raise NotImplementedError


class BodyCuboid(Body):
    """matlab-diff/+redmax/BodyCuboid.m"""

    def __init__(self, density, sides):
        super().__init__(density)
        self.sides = np.array(sides, dtype=np.float64).reshape(3)

    def computeInertia_(self):
        self.I_i = se3.inertiaCuboid(self.sides, self.density)  # BodyCuboid.m:17-20


class Joint:
    """Generic joint between a parent joint's body and ``body`` (Joint.m:56-92)."""

    jtype = None

    def __init__(self, parent, body, ndof):
        pname = parent.body.name if (parent is not None and parent.body is not None) else "NULL"
        self.name = "%s-%s" % (pname, body.name if body is not None else "NULL")
        self.parent = parent
        self.body = body
        self.children = []
        self.ndof = ndof
        self.q = np.zeros(max(ndof, 1))      # q(1)=... on a fixed joint is legal in the reference
        self.qdot = np.zeros(max(ndof, 1))
        self.qLimL = -1e8                     # Joint.m:77-80
        self.qLimU = 1e8
        self.qLimK = 1e8
        self.qLimD = 0.0
        self.tau = 0.0
        self.stiffness = 0.0
        self.damping = 0.0
        self.qRest = 0.0
        self.E0_pj = None
        self.axis = np.zeros(3)
        self.idxR = None
        body.joint = self
        if parent is not None:
            parent.children.append(self)

    # --- setters, same names as the reference (Joint.m:95-131) ---
    def setJointTransform(self, E):
        self.E0_pj = np.array(E, dtype=np.float64).reshape(4, 4)

    def setStiffness(self, stiffness):
        self.stiffness = float(stiffness)

    def setDamping(self, damping):
        self.damping = float(damping)

    def setLimitLower(self, limit):
        self.qLimL = float(limit)

    def setLimitUpper(self, limit):
        self.qLimU = float(limit)

    def setLimitStiffness(self, K):
        self.qLimK = float(K)

    def setLimitDamping(self, D):
        self.qLimD = float(D)

    def getTraversalOrder(self, order=None):
        """Joint.m:134-146: parent before children, depth first."""
        if order is None:
            order = []
        order.append(self)
        for c in self.children:
            c.getTraversalOrder(order)
        return order


class JointRevolute(Joint):
    jtype = JOINT_REVOLUTE

    def __init__(self, parent, body, axis):
        super().__init__(parent, body, 1)
        a = np.array(axis, dtype=np.float64).reshape(3)
        self.axis = a / np.linalg.norm(a)  # JointRevolute.m:14


class JointPrismatic(Joint):
    jtype = JOINT_PRISMATIC

    def __init__(self, parent, body, axis):
        super().__init__(parent, body, 1)
        a = np.array(axis, dtype=np.float64).reshape(3)
        self.axis = a / np.linalg.norm(a)  # JointPrismatic.m:15


class JointFixed(Joint):
    jtype = JOINT_FIXED

    def __init__(self, parent, body):
        super().__init__(parent, body, 0)


class JointPlanar(Joint):
    """2-DOF translation in the plane spanned by the (normalised) columns of ``plane`` (JointPlanar.m:11-19, 24-31)."""
    jtype = JOINT_PLANAR

    def __init__(self, parent, body, plane=None):
        super().__init__(parent, body, 2)
        B = np.array([[1, 0, 0], [0, 1, 0]], dtype=np.float64).T if plane is None else np.array(plane, dtype=np.float64).reshape(3, 2)
        self.plane = B / np.linalg.norm(B, axis=0, keepdims=True)


class JointTranslational(Joint):
    """3-DOF translation, Q(1:3,4) = q (JointTranslational.m:22-26)."""
    jtype = JOINT_TRANSLATIONAL

    def __init__(self, parent, body):
        super().__init__(parent, body, 3)


class JointUniversal(Joint):
    """Rotation about X then Y, R = X1(q1) Y2(q2) (JointUniversal.m:20-28, 71-74)."""
    jtype = JOINT_UNIVERSAL

    def __init__(self, parent, body):
        super().__init__(parent, body, 2)


class JointFree2D(Joint):
    """Free motion in the XY plane, Q = [Rz(q3) [q1;q2;0]] (JointFree2D.m:20-33)."""
    jtype = JOINT_FREE2D

    def __init__(self, parent, body):
        super().__init__(parent, body, 3)


class JointSpherical(Joint):
    """Ball joint parameterised by Euler angles in one of 12 charts, switched when the chart nears gimbal lock
    (JointSpherical.m:4-17, 28-34, 63-102).  The device holds it as three revolute nodes about the chart's axes and runs
    reparam_ after every step; ``chart`` is the construction-time chart (always XYZ in the reference)."""
    jtype = JOINT_SPHERICAL
    CHART_XYX, CHART_XZX, CHART_YZY, CHART_YXY, CHART_ZXZ, CHART_ZYZ = 1, 2, 3, 4, 5, 6
    CHART_XYZ, CHART_XZY, CHART_YZX, CHART_YXZ, CHART_ZXY, CHART_ZYX = 7, 8, 9, 10, 11, 12
    _AXES = {1: (0, 1, 0), 2: (0, 2, 0), 3: (1, 2, 1), 4: (1, 0, 1), 5: (2, 0, 2), 6: (2, 1, 2),
             7: (0, 1, 2), 8: (0, 2, 1), 9: (1, 2, 0), 10: (1, 0, 2), 11: (2, 0, 1), 12: (2, 1, 0)}

    def __init__(self, parent, body):
        super().__init__(parent, body, 3)
        self.radius = 1.0
        self.chart = JointSpherical.CHART_XYZ

    def setGeometry(self, radius):
        self.radius = float(radius)

    @staticmethod
    def getEuler(chart, q):
        """R of getEuler (JointSpherical.m:151-178): the product of the chart's three elementary rotations (codegen :247-262)."""
        R = np.eye(3)
        for a, qa in zip(JointSpherical._AXES[int(chart)], q):
            R = R @ se3.aaToMat(np.eye(3)[a], qa)
        return R

    @staticmethod
    def getEulerInv(chart, R):
        """getEulerInv (JointSpherical.m:181-208, XYXinv..ZYXinv :1809-1949); NaN at gimbal lock."""
        R = np.asarray(R, dtype=np.float64)
        i, j, a3 = JointSpherical._AXES[int(chart)]
        k = 3 - i - j
        e = 1.0 if (j - i) % 3 == 1 else -1.0
        if a3 == i:
            r = R[i, i]
            if -1.0 < r < 1.0:
                return np.array([math.atan2(R[j, i], -e * R[k, i]), math.acos(r), math.atan2(R[i, j], e * R[i, k])])
        else:
            r = R[i, k]
            if -1.0 < r < 1.0:
                return np.array([math.atan2(-e * R[j, k], R[k, k]), math.asin(e * r), math.atan2(-e * R[i, j], R[i, i])])
        return np.full(3, np.nan)


class JointFree3D(Joint):
    """6-DOF free joint = JointTranslational (q1..q3) followed by JointSpherical (q4..q6) (JointFree3D.m:1-34)."""
    jtype = JOINT_FREE3D

    def __init__(self, parent, body):
        super().__init__(parent, body, 6)
        self.chart = JointSpherical.CHART_XYZ


class ForceGroundCuboid:
    """Penalty ground contact with friction on the 8 corners of a cuboid (matlab-diff/+redmax/ForceGroundCuboid.m:1-47).
    Same setters as the reference; the numerics live in the HIP kernels."""

    def __init__(self, cuboid):
        self.name = cuboid.name + "-GROUND"
        self.cuboid = cuboid
        self.E = np.eye(4)
        self.kn, self.kt, self.mu, self.kd = 1.0, 0.0, 0.0, 0.0

    def setTransform(self, E):
        self.E = np.array(E, dtype=np.float64).reshape(4, 4)

    def setStiffness(self, kn, kt):
        self.kn, self.kt = float(kn), float(kt)

    def setDamping(self, kd):
        self.kd = float(kd)

    def setFriction(self, mu):
        self.mu = float(mu)

    def params(self):
        return (self.kn, self.kt, self.mu, self.kd) + tuple(self.E.reshape(-1))


class Scene:
    """Scene container (matlab-diff/+redmax/Scene.m).

    ``init()`` reproduces the index layout parity depends on: joints must be listed
    parent-before-child, reduced DOFs are numbered from the LAST listed joint to the
    first (Scene.m:65-71), ``qRest`` is captured from the initial ``q`` (Joint.m:157).
    """

    def __init__(self):
        Body._count = 1
        self.name = ""
        self.bodies = []
        self.joints = []
        self.forces = []
        self.tEnd = 1.0
        self.qInit = None
        self.qdotInit = None
        self.h = 1e-2
        self.t = 0.0
        self.k = 0
        self.T0 = 0.0
        self.V0 = 0.0
        self.history = []
        self.nsteps = 0
        self.grav = np.array([0.0, 0.0, -980.0])
        self.computeH = True
        self.Hexpected = np.zeros(2)
        self.nr = 0
        self.nm = 0
        self._desc = None

    # -- Scene.init, Scene.m:59-119 --
    def init(self):
        joints = self.joints
        n = len(joints)
        order = joints[0].getTraversalOrder()
        if len(order) != n or any(a is not b for a, b in zip(order, joints)):
            # The reference's getTraversalOrder always returns 1:n, i.e. it silently assumes this.
            raise ValueError("scene joints must be listed in depth-first, parent-before-child order")
        if any(j.body is not b for j, b in zip(joints, self.bodies)):
            raise ValueError("bodies must be listed in the same order as their joints")
        for f in self.forces:
            if not isinstance(f, ForceGroundCuboid):
                raise NotImplementedError("only ForceNull and ForceGroundCuboid are in scope (SURVEY.md §2 row 10)")
            if not isinstance(f.cuboid, BodyCuboid) or f.cuboid not in self.bodies:
                raise ValueError("ForceGroundCuboid needs a BodyCuboid of this scene")
        nr = 0
        nm = 0
        for j in reversed(joints):                   # leaf-to-root numbering
            j.idxR = list(range(nr, nr + j.ndof))
            nr += j.ndof
            j.body.idxM = list(range(nm, nm + 6))
            nm += 6
            j.qRest = float(j.q[0]) if j.ndof else 0.0   # Joint.m:157 (qRest = q; every DOF, see desc()["qRestR"])
            j.qRestAll = np.array(j.q[:j.ndof], dtype=np.float64)
        self.nr, self.nm = nr, nm
        for j in joints:
            if j.parent is not None and j.E0_pj is None:
                raise ValueError("joint %s needs setJointTransform (Joint.m:513 uses E0_jp)" % j.name)
        for b in self.bodies:
            b.computeInertia()
        self.qInit, self.qdotInit = self.getQ()
        self.nsteps = int(math.ceil(self.tEnd / self.h))
        self._desc = None
        self.t = 0.0
        self.k = 0
        self.history = []

    # -- gather / scatter in the reference's reduced ordering (Joint.getQ / setQ) --
    def getQ(self):
        q = np.zeros(self.nr)
        qdot = np.zeros(self.nr)
        for j in self.joints:
            if j.ndof:
                q[j.idxR] = j.q[:j.ndof]
                qdot[j.idxR] = j.qdot[:j.ndof]
        return q, qdot

    def setQ(self, q, qdot=None):
        for j in self.joints:
            if j.ndof:
                j.q[:j.ndof] = np.asarray(q)[j.idxR]
                if qdot is not None:
                    j.qdot[:j.ndof] = np.asarray(qdot)[j.idxR]

    @staticmethod
    def _cm(E):
        return np.asarray(E, dtype=np.float64).reshape(4, 4).T.reshape(16)  # column-major, as MATLAB stores it

    def desc(self):
        """Flatten the tree into the arrays of ``rmx_model_desc`` (include/redmax_hip.h)."""
        if self._desc is not None:
            return self._desc
        joints = self.joints
        n = len(joints)
        index = {id(j): i for i, j in enumerate(joints)}
        d = {
            "njoints": n,
            "parent": np.array([index[id(j.parent)] if j.parent is not None else -1 for j in joints], dtype=np.int32),
            "type": np.array([j.jtype for j in joints], dtype=np.int32),
            "axis": np.ascontiguousarray(np.stack([j.axis for j in joints]), dtype=np.float64),
            "E0_pj": np.ascontiguousarray(np.stack([self._cm(j.E0_pj if j.E0_pj is not None else np.eye(4)) for j in joints])),
            "E0_ji": np.ascontiguousarray(np.stack([self._cm(j.body.E0_ji) for j in joints])),
            "I_i": np.ascontiguousarray(np.stack([j.body.I_i for j in joints]), dtype=np.float64),
            "q": np.array([j.q[0] if j.ndof else 0.0 for j in joints], dtype=np.float64),
            "qdot": np.array([j.qdot[0] if j.ndof else 0.0 for j in joints], dtype=np.float64),
            "qRest": np.array([j.qRest for j in joints], dtype=np.float64),
            "tau": np.array([j.tau for j in joints], dtype=np.float64),
            "stiffness": np.array([j.stiffness for j in joints], dtype=np.float64),
            "damping": np.array([j.damping for j in joints], dtype=np.float64),
            "qLimL": np.array([j.qLimL for j in joints], dtype=np.float64),
            "qLimU": np.array([j.qLimU for j in joints], dtype=np.float64),
            "qLimK": np.array([j.qLimK for j in joints], dtype=np.float64),
            "qLimD": np.array([j.qLimD for j in joints], dtype=np.float64),
            "grav": np.array(self.grav, dtype=np.float64).reshape(3),
            # multi-DOF joints: plane of JointPlanar, and the state / rest positions of every DOF in reduced order
            "plane": np.ascontiguousarray(np.stack([getattr(j, "plane", np.zeros((3, 2))).T.reshape(6) for j in joints])),
            "qR": self.getQ()[0],
            "qdotR": self.getQ()[1],
            "qRestR": np.concatenate([np.zeros(0)] + [getattr(j, "qRestAll", np.array(j.q[:j.ndof], dtype=np.float64)) for j in reversed(joints)]),
        }
        if self.forces:
            # The reference keeps its force objects in a list (Force.m:26-56, Scene.m:87-89): a body may carry several
            # ForceGroundCuboid (a floor and a wall).  The C ABI takes ONE per listing entry (rmx_ground_contact: flags[n], E_body[n] ...),
            # so the second, third ... force of a body is listed as a fixed, massless child of that body's joint with the body's own
            # transform and sides: the same corners moving with the same twist, hence the same wrench, K and D blocks pulled through the
            # same Jacobian rows - as the multi-DOF joints are chains with massless links.  No DOF is added: q, qdot, idxR are unchanged.
            first = {}
            for f in self.forces:
                first.setdefault(id(f.cuboid), f)
            extra = [f for f in self.forces if first[id(f.cuboid)] is not f]
            body_joint = {id(j.body): i for i, j in enumerate(joints)}
            lit = None
            if extra:
                lit = dict(d)                    # the literal listing (what the oracle restates: forces appended to the body's own list)
                for key in ("parent", "type"):
                    d[key] = np.concatenate([d[key], np.array([body_joint[id(f.cuboid)] if key == "parent" else 0 for f in extra], dtype=np.int32)])
                d["axis"] = np.concatenate([d["axis"], np.tile(np.array([[0.0, 0.0, 1.0]]), (len(extra), 1))])
                d["E0_pj"] = np.concatenate([d["E0_pj"], np.tile(self._cm(np.eye(4))[None, :], (len(extra), 1))])
                d["E0_ji"] = np.concatenate([d["E0_ji"], np.stack([self._cm(f.cuboid.E0_ji) for f in extra])])
                d["I_i"] = np.concatenate([d["I_i"], np.zeros((len(extra), 6))])
                d["plane"] = np.concatenate([d["plane"], np.zeros((len(extra), 6))])
                for key in ("q", "qdot", "qRest", "tau", "stiffness", "damping", "qLimK", "qLimD"):
                    d[key] = np.concatenate([d[key], np.zeros(len(extra))])
                d["qLimL"] = np.concatenate([d["qLimL"], np.full(len(extra), -np.inf)])
                d["qLimU"] = np.concatenate([d["qLimU"], np.full(len(extra), np.inf)])
                d["njoints"] = n + len(extra)
            f0 = self.forces[0]
            flag = [1 if id(j.body) in first else 0 for j in joints]
            sides = [getattr(j.body, "sides", np.zeros(3)) for j in joints]
            fb = [first.get(id(j.body), f0) for j in joints]
            d["contact"] = np.array(flag + [1] * len(extra), dtype=np.int32)
            d["sides"] = np.ascontiguousarray(np.stack(sides + [f.cuboid.sides for f in extra]), dtype=np.float64)
            d["ground"] = {"E": f0.E.copy(), "kn": f0.kn, "kt": f0.kt, "mu": f0.mu, "kd": f0.kd}
            # every ForceGroundCuboid object holds its own E, kn, kt, mu, kd (ForceGroundCuboid.m:6-13): per body when they differ

            def per_entry(fs):
                return {"E": np.stack([np.asarray(f.E, dtype=np.float64) for f in fs]),
                        "kn": np.array([f.kn for f in fs], dtype=np.float64), "kt": np.array([f.kt for f in fs], dtype=np.float64),
                        "mu": np.array([f.mu for f in fs], dtype=np.float64), "kd": np.array([f.kd for f in fs], dtype=np.float64)}
            same = all(np.array_equal(f.E, f0.E) and (f.kn, f.kt, f.mu, f.kd) == (f0.kn, f0.kt, f0.mu, f0.kd) for f in self.forces)
            if not same:
                d["ground_body"] = per_entry(fb + extra)
            if lit is not None:
                lit["contact"] = np.array(flag, dtype=np.int32)
                lit["sides"] = np.ascontiguousarray(np.stack(sides), dtype=np.float64)
                lit["ground"] = d["ground"]
                if not same:
                    lit["ground_body"] = per_entry(fb)
                lit["extra_forces"] = [{"body": body_joint[id(f.cuboid)], "E": np.asarray(f.E, dtype=np.float64), "kn": f.kn, "kt": f.kt,
                                        "mu": f.mu, "kd": f.kd} for f in extra]
                self._desc_literal = lit
        self._desc = d
        return d

    def desc_literal(self):
        """The scene as the reference holds it - every force object in its body's own list (Force.m:26-56) - for a checker that
        restates the reference literally (tests: the oracle).  Equal to desc() unless a body carries several ForceGroundCuboid."""
        d = self.desc()
        return getattr(self, "_desc_literal", None) or d

    # -- Scene.reset, Scene.m:122-131 (energies come from the device, see driver.py) --
    def reset(self):
        self.setQ(self.qInit, self.qdotInit)
        self.t = 0.0
        self.k = 0
        self.history = []

    # -- Scene.saveHistory, Scene.m:134-161 --
    def saveHistory(self, q, qdot, T=None, V=None):
        rec = {"q": np.array(q), "qdot": np.array(qdot), "t": self.t}
        if self.computeH:
            rec["T"] = T
            rec["V"] = V
        self.history.append(rec)

    # -- Scene.plotEnergies, Scene.m:164-191 (the PASS/FAIL known-answer check, no plotting) --
    def plotEnergies(self, itype, verbose=True):
        """Returns (H_end, passed). ``itype`` is 1 (BDF1) or 2 (BDF2) as in the reference."""
        T = np.array([self.T0] + [r["T"] for r in self.history])
        V = np.array([self.V0] + [r["V"] for r in self.history])
        V = V - V[0]
        H = T + V
        passed = None
        if self.Hexpected[itype - 1] != 0:
            dH = H[-1] - self.Hexpected[itype - 1]
            passed = bool(abs(dH) <= 1e-2)
            if verbose:
                print("### PASS ###" if passed else "### FAIL: %.16e ###" % H[-1])
        return float(H[-1]), passed

    def countR(self):
        return self.nr

    def countM(self):
        return self.nm
