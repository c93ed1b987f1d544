"""KLT tracking wrapper + 2-D Kabsch with the reference's names and return orders
(reference getTransformKLT.py:77-84,129-162,317-381) on the MI355X (pyrklt.hip, kabsch_mds.hip)."""
import numpy as np

from . import _ffi

LK_PARAMS = dict(maxLevel=3, criteria=(3, 10, 0.03))     # (EPS|COUNT, 10, 0.03), winSize (15,15)
ERR_THRESHOLD = 10
N_FEATURES_BEFORE_RETRACK = 60                           # mutated to 80 by the first internal append, like :348-351


def calculateTransformSVD(srcCoords: np.ndarray, targetCoords: np.ndarray):
    """src ~= R tgt + h -> (R (2,2), h (2,1)), float64 (getTransformKLT.py:129-162)."""
    return _ffi.default_context().kabsch2d(srcCoords, targetCoords)


def getTrackedPointsKLT(srcImg: np.ndarray, targetImg: np.ndarray, blobCoordSrc: np.ndarray):
    """-> (good_new, good_old, bad_new, bad_old, correspondenceStatus (K,1) u8) — note new before old."""
    global N_FEATURES_BEFORE_RETRACK
    featurePtSrc = np.ascontiguousarray(blobCoordSrc[:, :2]).astype(np.float32)
    if featurePtSrc.shape[0] < N_FEATURES_BEFORE_RETRACK:
        from .getFeatures import appendNewFeatures
        featurePtSrc, N_FEATURES_BEFORE_RETRACK = appendNewFeatures(srcImg, featurePtSrc)
        print("WARNING: getTransformKLT added new features!")
    nextPts, status, err = _ffi.default_context().klt_track(srcImg, targetImg, featurePtSrc)
    status &= (err < ERR_THRESHOLD)
    good = (status == 1).flatten()
    bad = ~good
    return nextPts[good, :], featurePtSrc[good, :], nextPts[bad, :], featurePtSrc[bad, :], status
