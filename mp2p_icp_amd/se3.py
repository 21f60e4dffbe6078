"""Host-side SE(3) helpers for the ICP driver (termination criteria of ICP::align,
ICP.cpp:191-256) and for building poses.  Pose = 12 doubles: R row-major (9) + t (3).

Closed forms of the un-vendored MRPT calls: CPose3D(x,y,z,yaw,pitch,
roll) => R = Rz(yaw) Ry(pitch) Rx(roll); Lie::SE<3>::exp/log with tangent ordering [v; w].
"""
import math

import numpy as np


def identity():
    T = np.zeros(12)
    T[0] = T[4] = T[8] = 1.0
    return T


def from_xyzypr(x, y, z, yaw=0.0, pitch=0.0, roll=0.0):
    cy, sy = math.cos(yaw), math.sin(yaw)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cr, sr = math.cos(roll), math.sin(roll)
    return np.array([cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                     sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                     -sp, cp * sr, cp * cr, x, y, z], dtype=np.float64)


def to_xyzypr(T):
    R = np.asarray(T[:9]).reshape(3, 3)
    sp = -R[2, 0]
    pitch = math.asin(max(-1.0, min(1.0, sp)))
    if abs(abs(sp) - 1.0) < 1e-12:
        yaw, roll = math.atan2(-R[0, 1], R[1, 1]), 0.0
    else:
        yaw, roll = math.atan2(R[1, 0], R[0, 0]), math.atan2(R[2, 1], R[2, 2])
    return np.array([T[9], T[10], T[11], yaw, pitch, roll])


def Rt(T):
    T = np.asarray(T, dtype=np.float64)
    return T[:9].reshape(3, 3), T[9:12]


def from_Rt(R, t):
    return np.concatenate([np.asarray(R, dtype=np.float64).reshape(9), np.asarray(t, dtype=np.float64)])


def compose(A, B):
    """A (+) B"""
    Ra, ta = Rt(A)
    Rb, tb = Rt(B)
    return from_Rt(Ra @ Rb, Ra @ tb + ta)


def inverse(A):
    R, t = Rt(A)
    return from_Rt(R.T, -R.T @ t)


def inverse_compose(A, B):
    """A (-) B = B^-1 (+) A   (CPose3D operator-)"""
    return compose(inverse(B), A)


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def exp(xi):
    v, w = np.asarray(xi[:3], float), np.asarray(xi[3:], float)
    th2 = float(w @ w)
    th = math.sqrt(th2)
    if th < 1e-6:
        a, b, c = 1 - th2 / 6, 0.5 - th2 / 24, 1 / 6 - th2 / 120
    else:
        a, b, c = math.sin(th) / th, (1 - math.cos(th)) / th2, (th - math.sin(th)) / (th2 * th)
    W = _skew(w)
    W2 = W @ W
    R = np.eye(3) + a * W + b * W2
    V = np.eye(3) + b * W + c * W2
    return from_Rt(R, V @ v)


def log(T):
    R, t = Rt(T)
    c = max(-1.0, min(1.0, 0.5 * (np.trace(R) - 1.0)))
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = 0.5 * float(np.linalg.norm(v))
    th = math.atan2(s, c)
    if th < 1e-6:
        w = 0.5 * (1 + th * th / 6) * v
    elif math.pi - th < 1e-6:
        ax = np.sqrt(np.maximum(0.0, 0.5 * (np.diag(R) + 1.0)))
        k = int(np.argmax(ax))
        for i in range(3):
            if i != k and (R[k, i] + R[i, k]) < 0:
                ax[i] = -ax[i]
        if float(v @ ax) < 0:
            ax = -ax
        w = th * ax
    else:
        w = th / (2 * s) * v
    th2 = float(w @ w)
    th = math.sqrt(th2)
    if th < 1e-6:
        k = 1 / 12 + th2 / 720
    else:
        k = (1 - (th * math.sin(th)) / (2 * (1 - math.cos(th)))) / th2
    W = _skew(w)
    Vi = np.eye(3) - 0.5 * W + k * (W @ W)
    return np.concatenate([Vi @ t, w])
