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
    """skimage.feature.blob._prune_blobs (scikit-image 0.19.2), pair order included: the candidate pairs come out of
    cKDTree.query_pairs as a Python set and are visited in that set's iteration order - chains of overlapping blobs
    make the result order dependent (0-3 blobs per real frame), so the same construct is used here.
    (The engine's device-side retrack restates the same order natively: csrc/blobprune.h.)"""
    from scipy import spatial
    sigma = blobs[:, -1].max()
    distance = 2 * sigma * math.sqrt(blobs.shape[1] - 1)
    pairs = np.array(list(spatial.cKDTree(blobs[:, :-1]).query_pairs(distance)))
    if len(pairs) == 0:
        return blobs
    blobs = blobs.copy()
    for i, j in pairs:
        b1, b2 = blobs[i], blobs[j]
        if _blob_overlap(b1, b2) > overlap:
            if b1[-1] > b2[-1]:
                b2[-1] = 0
            else:
                b1[-1] = 0
    return blobs[blobs[:, -1] > 0]


def argsort_numpy122(keys) -> np.ndarray:
    """np.argsort(keys) with the tie order of the reference's pinned NumPy (1.22.3, requirements.txt:3): its default
    sort is an introsort (median of three, Hoare partition, insertion sort for runs of <= 17 elements) that is NOT
    stable, and adaptiveNMS sorts ~400 blobs by a sigma that takes two values (getFeatures.py:69) - the order of
    the ties decides which blobs SSC keeps.  NumPy >= 2 uses vectorised sorts with a different tie order."""
    v = [float(x) for x in keys]
    n = len(v)
    t = list(range(n))
    if n < 2:
        return np.array(t, np.int64)
    lo, hi, todo, budget = 0, n - 1, [], 2 * (n.bit_length() - 1)
    while True:
        if budget < 0:                                   # depth limit: heap sort of the run (never reached for blob counts)
            t[lo:hi + 1] = _heap_argsort(v, t[lo:hi + 1])
        else:
            while hi - lo > 16:
                mid = lo + ((hi - lo) >> 1)
                if v[t[mid]] < v[t[lo]]: t[mid], t[lo] = t[lo], t[mid]
                if v[t[hi]] < v[t[mid]]: t[hi], t[mid] = t[mid], t[hi]
                if v[t[mid]] < v[t[lo]]: t[mid], t[lo] = t[lo], t[mid]
                pivot = v[t[mid]]
                i, j = lo, hi - 1
                t[mid], t[j] = t[j], t[mid]
                while True:
                    i += 1
                    while v[t[i]] < pivot: i += 1
                    j -= 1
                    while pivot < v[t[j]]: j -= 1
                    if i >= j:
                        break
                    t[i], t[j] = t[j], t[i]
                t[i], t[hi - 1] = t[hi - 1], t[i]
                budget -= 1
                if i - lo < hi - i:                      # the larger part waits, the smaller one is sorted first
                    todo.append((i + 1, hi, budget)); hi = i - 1
                else:
                    todo.append((lo, i - 1, budget)); lo = i + 1
            for i in range(lo + 1, hi + 1):
                cur, j = t[i], i
                while j > lo and v[cur] < v[t[j - 1]]:
                    t[j] = t[j - 1]; j -= 1
                t[j] = cur
        if not todo:
            break
        lo, hi, budget = todo.pop()
    return np.array(t, np.int64)


def _heap_argsort(v, idx):
    a = [None] + list(idx)
    n = len(idx)
    for l in range(n >> 1, 0, -1):
        tmp, i, j = a[l], l, l << 1
        while j <= n:
            if j < n and v[a[j]] < v[a[j + 1]]: j += 1
            if v[tmp] < v[a[j]]:
                a[i] = a[j]; i = j; j += j
            else:
                break
        a[i] = tmp
    while n > 1:
        tmp = a[n]; a[n] = a[1]; n -= 1
        i, j = 1, 2
        while j <= n:
            if j < n and v[a[j]] < v[a[j + 1]]: j += 1
            if v[tmp] < v[a[j]]:
                a[i] = a[j]; i = j; j += j
            else:
                break
        a[i] = tmp
    return a[1:]


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
