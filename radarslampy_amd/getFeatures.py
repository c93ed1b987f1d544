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


def _prune_blobs(blobs, overlap):
    """skimage.feature.blob._prune_blobs (scikit-image 0.19.2), PAIR ORDER INCLUDED: scikit-image takes the candidate pairs
    out of cKDTree.query_pairs as a Python set and visits them in that set's iteration order; chains of overlapping
    blobs make the survivors depend on it (0-3 blobs per real frame, which then reshuffle ~15 % of the ANMS selection).
    roam_prune_blobs (csrc/blobprune.h) restates cKDTree's emission order and CPython's set order natively; the
    engine's device-side retrack runs the same code on the GPU."""
    import ctypes as C
    bl = np.ascontiguousarray(blobs, np.float64)
    keep = np.zeros(len(bl), np.uint8)
    rc = _ffi.load_library().roam_prune_blobs(_ffi._ptr(bl), len(bl), C.c_double(overlap), _ffi._ptr(keep))
    if rc != _ffi.ROAM_OK:
        raise _ffi.RoamError(rc, "roam_prune_blobs: %d blobs (integer pixel coordinates, <= 32767 candidate pairs)" % len(bl))
    return bl[keep.astype(bool)]


def argsort_numpy122(keys) -> np.ndarray:
    """np.argsort(keys) with the tie order of the reference's pinned NumPy (1.22.3, requirements.txt:3): its default
    sort is an introsort (median of three, Hoare partition, insertion sort for runs of <= 17 elements) that is NOT
    stable, and adaptiveNMS sorts ~400 blobs by a sigma that takes two values (getFeatures.py:69) - the order of
    the ties decides which blobs SSC keeps.  NumPy >= 2 uses vectorised sorts with a different tie order."""
    k = np.ascontiguousarray(keys, np.float64)
    out = np.empty(max(len(k), 1), np.int32)
    rc = _ffi.load_library().roam_argsort_np122(_ffi._ptr(k), len(k), _ffi._ptr(out))
    if rc != _ffi.ROAM_OK:
        raise _ffi.RoamError(rc, "roam_argsort_np122")
    return out[:len(k)].astype(np.int64)


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
    idx = np.argsort(-val, kind="stable")         # peak_local_max: highest response first
    lm = rcs[idx].astype(np.float64)
    lm[:, -1] = sigma_list[rcs[idx][:, -1]]
    return _prune_blobs(lm, overlap)


def adaptiveNMS(img, blobs, ret_points=200, tolerance=0.1):
    H, W = img.shape
    keypoints = blobs[argsort_numpy122(blobs[:, 2]), :]
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
