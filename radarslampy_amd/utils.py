"""SE(2) helpers with the reference's names and conventions (reference utils.py:29-103,147-165).
Host-side numpy only (a handful of scalars per frame); plotting helpers are out of scope."""
import time

import numpy as np


def tic():
    return time.time()


def toc(t):
    return time.time() - t


def normalize_angles(th):
    """Wrap to [-pi, pi) (utils.py:29-33)."""
    return (th + np.pi) % (2 * np.pi) - np.pi


def getRotationMatrix(th, degrees=False):
    if degrees:
        th = np.deg2rad(th)
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, -s], [s, c]])


def convertPoseToTransform(poses):
    """(3,) or (N,3) [x,y,th] -> (3,3) or (N,3,3) (utils.py:46-72)."""
    poses = np.asarray(poses, dtype=np.float64)
    single = poses.ndim == 1
    p = np.atleast_2d(poses)
    T = np.zeros((len(p), 3, 3))
    c, s = np.cos(p[:, 2]), np.sin(p[:, 2])
    T[:, 0, 0], T[:, 0, 1], T[:, 1, 0], T[:, 1, 1] = c, -s, s, c
    T[:, 0, 2], T[:, 1, 2], T[:, 2, 2] = p[:, 0], p[:, 1], 1
    return T[0] if single else T


def convertTransformToPose(T):
    """(3,3) or (N,3,3) -> (3,) or (N,3) (utils.py:75-92)."""
    T = np.asarray(T, dtype=np.float64)
    single = T.ndim == 2
    A = T[None] if single else T
    out = np.stack([A[:, 0, 2], A[:, 1, 2], np.arctan2(A[:, 1, 0], A[:, 0, 0])], axis=1)
    return out[0] if single else out


def convertRandHtoDeltas(R, h):
    """(utils.py:99-103)"""
    h = np.asarray(h, dtype=np.float64).reshape(-1)
    return np.array([float(h[0]), float(h[1]), np.arctan2(R[1, 0], R[0, 0])])


def invert_transform(T):
    th = np.arctan2(T[1, 0], T[0, 0])
    x, y = T[0, 2], T[1, 2]
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, s, -s * y - c * x], [-s, c, -c * y + s * x], [0, 0, 1]])


def homogenize(points):
    points = np.asarray(points)
    if points.shape[1] == 2:
        return np.concatenate((points, np.ones((points.shape[0], 1))), axis=1)
    return points.copy()


def radarImgPathToTimestamp(radarImgPath):
    import os
    return int(os.path.basename(radarImgPath)[:-4])
