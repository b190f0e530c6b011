"""SE(3)/se(3) helpers used on the host side when building scenes.

Mirrors the static-method surface of the reference's ``se3`` class
(matlab-diff/se3.m; identical file in matlab/ and matlab-simple/) for the subset the
forward-dynamics path needs: ``inv`` (:11-16), ``Ad`` (:44-52), ``ad`` (:55-69),
``brac`` (:89-98), ``aaToMat`` (:111-176), ``inertiaCuboid`` (:366-379).
Twists are ordered [omega; v]; transforms are 4x4 numpy arrays.
"""
import math

import numpy as np

THRESH = 1e-9  # se3.m:5


def inv(E):
    E = np.asarray(E, dtype=np.float64)
    R = E[:3, :3]
    p = E[:3, 3]
    Ei = np.eye(4)
    Ei[:3, :3] = R.T
    Ei[:3, 3] = -R.T @ p
    return Ei


def brac(x):
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    S = np.array([[0.0, -x[2], x[1]], [x[2], 0.0, -x[0]], [-x[1], x[0], 0.0]])
    if x.size < 6:
        return S
    B = np.zeros((4, 4))
    B[:3, :3] = S
    B[:3, 3] = x[3:6]
    return B


def Ad(E):
    E = np.asarray(E, dtype=np.float64)
    A = np.zeros((6, 6))
    R = E[:3, :3]
    p = E[:3, 3]
    A[:3, :3] = R
    A[3:, 3:] = R
    A[3:, :3] = brac(p) @ R
    return A


def ad(phi):
    phi = np.asarray(phi, dtype=np.float64).reshape(-1)
    a = np.zeros((6, 6))
    W = brac(phi[:3])
    a[:3, :3] = W
    a[3:, :3] = brac(phi[3:6])
    a[3:, 3:] = W
    return a


def aaToMat(axis, angle):
    """Rotation matrix from an (axis, angle) pair, with the reference's exact-zero
    special cases for axis-aligned rotations (se3.m:111-176)."""
    ax, ay, az = (float(v) for v in np.asarray(axis, dtype=np.float64).reshape(-1)[:3])
    R = np.eye(3)
    mag = math.sqrt(ax * ax + ay * ay + az * az)
    if mag > THRESH:
        mag = 1.0 / mag
        ax, ay, az = ax * mag, ay * mag, az * mag
        if abs(ax) < THRESH and abs(ay) < THRESH:
            if az < 0:
                angle = -angle
            s, c = math.sin(angle), math.cos(angle)
            R[0, 0], R[0, 1], R[1, 0], R[1, 1] = c, -s, s, c
        elif abs(ay) < THRESH and abs(az) < THRESH:
            if ax < 0:
                angle = -angle
            s, c = math.sin(angle), math.cos(angle)
            R[1, 1], R[1, 2], R[2, 1], R[2, 2] = c, -s, s, c
        elif abs(az) < THRESH and abs(ax) < THRESH:
            if ay < 0:
                angle = -angle
            s, c = math.sin(angle), math.cos(angle)
            R[0, 0], R[0, 2], R[2, 0], R[2, 2] = c, s, -s, c
        else:
            s, c = math.sin(angle), math.cos(angle)
            t = 1.0 - c
            xz, xy, yz = ax * az, ax * ay, ay * az
            R[0, 0] = t * ax * ax + c
            R[0, 1] = t * xy - s * az
            R[0, 2] = t * xz + s * ay
            R[1, 0] = t * xy + s * az
            R[1, 1] = t * ay * ay + c
            R[1, 2] = t * yz - s * ax
            R[2, 0] = t * xz - s * ay
            R[2, 1] = t * yz + s * ax
            R[2, 2] = t * az * az + c
    return R


def inertiaCuboid(whd, density):
    whd = np.asarray(whd, dtype=np.float64).reshape(-1)
    mass = density * float(np.prod(whd))
    m = np.zeros(6)
    m[0] = (1.0 / 12.0) * mass * (whd[1] ** 2 + whd[2] ** 2)
    m[1] = (1.0 / 12.0) * mass * (whd[2] ** 2 + whd[0] ** 2)
    m[2] = (1.0 / 12.0) * mass * (whd[0] ** 2 + whd[1] ** 2)
    m[3:] = mass
    return m


def transform(R=None, p=None):
    """[R p; 0 0 0 1] convenience constructor (the scenes write this inline)."""
    E = np.eye(4)
    if R is not None:
        E[:3, :3] = np.asarray(R, dtype=np.float64)
    if p is not None:
        E[:3, 3] = np.asarray(p, dtype=np.float64).reshape(3)
    return E
