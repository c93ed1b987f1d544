"""Drop-in for the reference's Tracker (reference Tracker.py:16-127): same constructor, same
track()/getTransform() signatures, return orders and dtypes; the arithmetic runs on the MI355X.

Differences, both documented in DESIGN.md: (1) the Fourier-Mellin rotation estimate is dead
compute in the reference (its result is only printed, RawROAMSystem.py:187-188) and is not
built - slot 3 of track() is 0.0; (2) paramFlags["rejectOutliers"]=False returns an all-ones
pruning mask where the reference raises NameError (Tracker.py:93-104)."""
import time

import numpy as np

from . import getTransformKLT as _klt
from . import outlierRejection as _orj
from .parseData import RANGE_RESOLUTION_CART_M


def getTrackedPointsKLT(srcImg, targetImg, blobCoordSrc):      # module-level hook (tests patch it, like the reference's import)
    return _klt.getTrackedPointsKLT(srcImg, targetImg, blobCoordSrc)


class Tracker():
    def __init__(self, sequenceName, imgPathArr, filePaths, paramFlags) -> None:
        self.sequenceName, self.imgPathArr = sequenceName, imgPathArr
        self.sequenceSize = len(imgPathArr)
        self.filePaths, self.paramFlags = filePaths, paramFlags
        self.estTraj = self.gtTraj = None
        self.verbose = False

    def initTraj(self, estTraj, gtTraj=None):
        self.estTraj, self.gtTraj = estTraj, gtTraj

    def track(self, prevImgCart, currImgCart, prevImgPolar, currImgPolar, featureCoord, seqInd):
        """-> (good_old (K',2) f32, good_new (K',2) f32, angleRotRad, corrStatus (K,1) u8); the polar
        images are accepted for signature compatibility only (they fed the dead FMT estimate)."""
        t0 = time.time()
        new_ok, old_ok, new_bad, _, status = getTrackedPointsKLT(prevImgCart, currImgCart, featureCoord)
        n_all = new_ok.shape[0] + new_bad.shape[0]
        if self.verbose:
            print(f"{seqInd} | Num good features: {new_ok.shape[0]} of {n_all} | Time: {time.time() - t0:.2f}s")
        if self.paramFlags.get("rejectOutliers", True):
            old_ok, new_ok, keep = _orj.rejectOutliers(old_ok, new_ok)
        else:
            keep = np.ones(old_ok.shape[0], dtype=bool)
        alive = np.flatnonzero(status.reshape(-1) != 0)            # rows of corrStatus that KLT kept
        status[alive] &= keep.astype(status.dtype)[:, None]
        return old_ok, new_ok, 0.0, status

    def getTransform(self, srcCoord, targetCoord, pixel: bool):
        """-> (R (2,2), h (2,1)); h in metres when pixel=False (Tracker.py:108-127)."""
        R, h = _klt.calculateTransformSVD(srcCoord, targetCoord)
        return (R, h) if pixel else (R, h * RANGE_RESOLUTION_CART_M)
