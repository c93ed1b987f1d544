"""Drop-in for the reference's Tracker (reference Tracker.py:16-127): same constructor, same
track()/getTransform() signatures, return orders and dtypes; the arithmetic runs on the MI355X.

Differences, both documented in DESIGN.md: (1) the Fourier-Mellin rotation estimate is dead
compute in the reference (its result is only printed, RawROAMSystem.py:187-188) and is not
built - slot 3 of track() is 0.0; (2) paramFlags["rejectOutliers"]=False returns an all-ones
pruning mask where the reference raises NameError (Tracker.py:93-104)."""
from typing import Tuple

import numpy as np

from .getTransformKLT import calculateTransformSVD, getTrackedPointsKLT
from .outlierRejection import rejectOutliers
from .parseData import RANGE_RESOLUTION_CART_M
from .utils import tic, toc


class Tracker():
    def __init__(self, sequenceName: str, imgPathArr, filePaths, paramFlags) -> None:
        self.sequenceName = sequenceName
        self.imgPathArr = imgPathArr
        self.sequenceSize = len(self.imgPathArr)
        self.filePaths = filePaths
        self.paramFlags = paramFlags
        self.estTraj = None
        self.gtTraj = None
        self.verbose = False

    def initTraj(self, estTraj, gtTraj=None):
        self.estTraj = estTraj
        self.gtTraj = gtTraj

    def track(self, prevImgCart: np.ndarray, currImgCart: np.ndarray, prevImgPolar: np.ndarray,
              currImgPolar: np.ndarray, featureCoord: np.ndarray, seqInd: int) -> Tuple[np.ndarray, np.ndarray, float, np.ndarray]:
        """-> (good_old (K',2) f32, good_new (K',2) f32, angleRotRad, corrStatus (K,1) u8)"""
        start = tic()
        angleRotRad = 0.0
        good_new, good_old, bad_new, bad_old, corrStatus = getTrackedPointsKLT(prevImgCart, currImgCart, featureCoord)
        nFeatures = good_new.shape[0] + bad_new.shape[0]
        if self.verbose:
            print(f"{seqInd} | Num good features: {good_new.shape[0]} of {nFeatures} | Time: {toc(start):.2f}s")
        if self.paramFlags.get("rejectOutliers", True):
            good_old, good_new, pruning_mask = rejectOutliers(good_old, good_new)
        else:
            pruning_mask = np.ones(good_old.shape[0], dtype=bool)
        rng = np.arange(nFeatures)
        corrStatus[rng[corrStatus.flatten().astype(bool)]] &= pruning_mask[:, np.newaxis]
        return good_old, good_new, angleRotRad, corrStatus

    def getTransform(self, srcCoord: np.ndarray, targetCoord: np.ndarray, pixel: bool):
        """-> (R (2,2), h (2,1)); h in metres when pixel=False (Tracker.py:108-127)."""
        R, h = calculateTransformSVD(srcCoord, targetCoord)
        if not pixel:
            h *= RANGE_RESOLUTION_CART_M
        return R, h
