"""Feature (re)detection with the reference's names (reference getFeatures.py:13-118):
Determinant-of-Hessian blobs (doh.hip) + SSC-ANMS (ssc.hip) + dedupe-append."""
import math

import numpy as np

from . import _ffi
from .ANMS import ssc

DEFAULT_FEATURE_PARAMS = dict(min_sigma=0.01, max_sigma=10, num_sigma=3, threshold=.0005, method="doh")
PERCENT_FEATURE_LOSS_THRESHOLD = 0.75
N_FEATURES_BEFORE_RETRACK = 60


def calculateFeatureLossThreshold(nInitialFeatures):
    return 80


def _disk_overlap(d, r1, r2):
    ratio1 = np.clip((d ** 2 + r1 ** 2 - r2 ** 2) / (2 * d * r1), -1, 1)
    ratio2 = np.clip((d ** 2 + r2 ** 2 - r1 ** 2) / (2 * d * r2), -1, 1)
    a, b, c, dd = -d + r2 + r1, d - r2 + r1, d + r2 - r1, d + r2 + r1
    area = r1 ** 2 * math.acos(ratio1) + r2 ** 2 * math.acos(ratio2) - 0.5 * math.sqrt(abs(a * b * c * dd))
    return area / (math.pi * (min(r1, r2) ** 2))


def _blob_overlap(b1, b2):
    """skimage.feature.blob._blob_overlap for 2-D blobs [row, col, sigma]."""
    root2 = math.sqrt(2)
    if b1[2] == b2[2] == 0:
        return 0.0
    if b1[2] > b2[2]:
        ms, r1, r2 = b1[2], 1.0, b2[2] / b1[2]
    else:
        ms, r2, r1 = b2[2], 1.0, b1[2] / b2[2]
    d = math.sqrt(((b2[0] - b1[0]) / (ms * root2)) ** 2 + ((b2[1] - b1[1]) / (ms * root2)) ** 2)
    if d > r1 + r2:
        return 0.0
    if d <= abs(r1 - r2):
        return 1.0
    return _disk_overlap(d, r1, r2)


def _prune_blobs(blobs, overlap):
    """skimage.feature.blob._prune_blobs.  scikit-image walks the candidate pairs in the
    iteration order of a Python set (not reproducible); here they are walked in ascending
    (i, j) order of the response-sorted blob array - documented in DESIGN.md."""
    from scipy import spatial
    sigma = blobs[:, -1].max()
    distance = 2 * sigma * math.sqrt(blobs.shape[1] - 1)
    pairs = spatial.cKDTree(blobs[:, :-1]).query_pairs(distance, output_type="ndarray")
    if len(pairs) == 0:
        return blobs
    pairs = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]
    blobs = blobs.copy()
    for i, j in pairs:
        b1, b2 = blobs[i], blobs[j]
        if _blob_overlap(b1, b2) > overlap:
            if b1[-1] > b2[-1]:
                b2[-1] = 0
            else:
                b1[-1] = 0
    return blobs[blobs[:, -1] > 0]


def getBlobsFromCart(cartImage: np.ndarray, min_sigma=1, max_sigma=30, num_sigma=10, threshold=0.01, method="doh",
                     overlap=0.5) -> np.ndarray:
    """-> (K,3) [r, c, sigma] (getFeatures.py:22-53; only the live method 'doh' is built).
    Image-scale work (integral image, Hessian determinants, 3x3x3 maxima) runs on the MI355X
    (doh.hip); the response ordering / sigma lookup / overlap pruning below follow
    skimage.feature.blob_doh (peak_local_max ordering, _prune_blobs)."""
    if method != "doh":
        raise NotImplementedError(f"{method} not implemented! Use 'doh'")
    sigma_list = np.linspace(min_sigma, max_sigma, num_sigma)
    rcs, val = _ffi.default_context().doh_maxima(cartImage, sigma_list, threshold)
    return blobs_from_maxima(rcs, val, sigma_list, overlap)


def blobs_from_maxima(rcs, val, sigma_list, overlap=0.5):
    """host bookkeeping of blob_doh after the image-scale work: response order, sigma lookup, pruning"""
    if len(rcs) == 0:
        return np.empty((0, 3))
    idx = np.argsort(-val)                        # peak_local_max: highest response first
    lm = rcs[idx].astype(np.float64)
    lm[:, -1] = sigma_list[rcs[idx][:, -1]]
    return _prune_blobs(lm, overlap)


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
