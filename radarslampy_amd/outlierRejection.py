"""Consistency-graph outlier rejection (reference outlierRejection.py:10-95) on the MI355X
(clique.hip): maximum clique of the |d_prev - d_new| <= 0.5 m graph; when several maximum cliques
exist, the one the reference returns - the first in networkx.find_cliques order (outlierRejection.py:63-75)."""
import numpy as np

from . import _ffi
from .parseData import RANGE_RESOLUTION_CART_M

DIST_THRESHOLD_M = 0.5
DIST_THRESHOLD_PX = DIST_THRESHOLD_M / RANGE_RESOLUTION_CART_M
DISTSQ_THRESHOLD_PX = DIST_THRESHOLD_PX * DIST_THRESHOLD_PX


def rejectOutliers(prev_coord: np.ndarray, new_coord: np.ndarray):
    """-> (pruned_prev_coord, pruned_new_coord, pruning_mask bool (K,))"""
    assert prev_coord.shape == new_coord.shape, "Coordinates should be the same shape"
    K = prev_coord.shape[0]
    if K == 0:
        return prev_coord, new_coord, np.zeros(0, dtype=bool)
    if K > _ffi.MAX_FEATURES:
        raise ValueError(f"at most {_ffi.MAX_FEATURES} correspondences")
    mask, n_in, flags, _ = _ffi.default_context().reject_outliers(prev_coord, new_coord, DIST_THRESHOLD_PX)
    return prev_coord[mask], new_coord[mask], mask
