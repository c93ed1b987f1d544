"""Drop-in for the reference's Tracker (reference Tracker.py:16-127): same constructor, same
track()/getTransform() signatures, return orders and dtypes; the arithmetic runs on the MI355X.

The Fourier-Mellin rotation estimate that track() computes first and returns in slot 3 (Tracker.py:62-63; the reference
only prints it, RawROAMSystem.py:187-188) runs on the GPU as well (FMT.getRotationUsingFMT, csrc/fmt.hip).
One documented difference: paramFlags["rejectOutliers"]=False returns an all-ones pruning mask where the reference raises
NameError (Tracker.py:93-104)."""
import time

import numpy as np

from . import FMT as _fmt
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
        """-> (good_old (K',2) f32, good_new (K',2) f32, angleRotRad, corrStatus (K,1) u8)"""
        t0 = time.time()
        angleRotRad = 0.0
        if prevImgPolar is not None and currImgPolar is not None:
            angleRotRad, _, _ = _fmt.getRotationUsingFMT(prevImgPolar, currImgPolar)
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
        return old_ok, new_ok, angleRotRad, status

    def getTransform(self, srcCoord, targetCoord, pixel: bool):
        """-> (R (2,2), h (2,1)); h in metres when pixel=False (Tracker.py:108-127)."""
        R, h = _klt.calculateTransformSVD(srcCoord, targetCoord)
        return (R, h) if pixel else (R, h * RANGE_RESOLUTION_CART_M)
