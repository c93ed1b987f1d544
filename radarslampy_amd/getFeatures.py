"""Feature (re)detection with the reference's names (reference getFeatures.py:13-118):
Determinant-of-Hessian blobs (doh.hip) + SSC-ANMS (ssc.hip) + dedupe-append."""
import numpy as np

from . import _ffi
from .ANMS import ssc

DEFAULT_FEATURE_PARAMS = dict(min_sigma=0.01, max_sigma=10, num_sigma=3, threshold=.0005, method="doh")
PERCENT_FEATURE_LOSS_THRESHOLD = 0.75
N_FEATURES_BEFORE_RETRACK = 60


def calculateFeatureLossThreshold(nInitialFeatures):
    return 80


def getBlobsFromCart(cartImage: np.ndarray, min_sigma=1, max_sigma=30, num_sigma=10, threshold=0.01, method="doh") -> np.ndarray:
    """-> (K,3) [r, c, sigma] (getFeatures.py:22-53; only the live method 'doh' is built)."""
    if method != "doh":
        raise NotImplementedError(f"{method} not implemented! Use 'doh'")
    return _ffi.default_context().doh_blobs(cartImage, min_sigma, max_sigma, num_sigma, threshold, 0.5)


def adaptiveNMS(img, blobs, ret_points=200, tolerance=0.1):
    H, W = img.shape
    keypoints = blobs[np.argsort(blobs[:, 2]), :]
    return ssc(keypoints, ret_points, tolerance, W, H)


def getFeatures(img, feature_params: dict = DEFAULT_FEATURE_PARAMS):
    blobs = getBlobsFromCart(img, **feature_params)
    blobs = adaptiveNMS(img, blobs)
    return np.fliplr(blobs[:, :2]), blobs[:, 2]


def dedupe_append(oldFeaturesCoord, newFeatureCoord):
    """vstack + drop exact duplicate rows keeping the first occurrence (getFeatures.py:109-112)."""
    pts = np.vstack((oldFeaturesCoord, newFeatureCoord))
    _, idx = np.unique(pts, axis=0, return_index=True)
    return np.ascontiguousarray(pts[np.sort(idx)]).astype(np.float32)


def appendNewFeatures(srcImg, oldFeaturesCoord):
    newFeatureCoord, _ = getFeatures(srcImg)
    featurePtSrc = dedupe_append(oldFeaturesCoord, newFeatureCoord)
    return featurePtSrc, calculateFeatureLossThreshold(featurePtSrc.shape[0])
