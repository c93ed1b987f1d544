"""Suppression via square covering (reference ANMS.py:5-102) on the MI355X (ssc.hip)."""
import numpy as np

from . import _ffi


def ssc(keypoints, num_ret_points, tolerance, cols, rows):
    """keypoints (B,3) [row, col, sigma] in priority order -> selected rows (same order)."""
    kp = np.ascontiguousarray(keypoints, dtype=np.float64)
    if kp.shape[0] == 0:
        return np.empty((0, 3))
    sel = _ffi.default_context().ssc(kp, num_ret_points, tolerance, cols, rows)
    return kp[sel]
