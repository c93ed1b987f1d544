"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's (Samleo8/RadarSLAMPy) per-scan hot path, used as the
checker in tests/, in __graft_entry__.smoke() and as bench.py's `cpu_baseline` leg.
Nothing under radarslampy_amd/ imports this package; the product path has no CPU route.

Layout
  oracle/c/*.c      plain-C restatements (gcc -O2 -ffp-contract=off -> oracle/_build/liboracle.so)
  oracle/__init__.py  ctypes wrappers with the reference's call signatures + the small
                    numpy-only pieces (Kabsch, SE(2) helpers, feature dedupe, Tracker glue)

Parity status (details in DESIGN.md §4):
  PINNED by goldens produced by the reference itself (tests/golden/*.npz, make_goldens.py):
      getPointCloudPolarInd, ssc, calculateTransformSVD, rejectOutliers (the reference's MASK, ties between maximum
      cliques included: networkx.find_cliques order restated in oracle/c/clique.c and checked against the live
      networkx / CPython of this image), MotionDistortionSolver (error_vector, undistort,
      compute_time_deltas, optimize_library), utils SE(2) helpers, record decode,
      Tracker.track glue, Keyframe glue.
  PINNED by outputs of the reference's own cv2 / scikit-image / SciPy / NumPy stack that the reference
  repository holds for its 11 real data/tiny scans (tests/golden/tiny_track.npz, make_tiny_track.py;
  tests/test_oracle_reference_dump.py):
      convertPolarImageToCartesian (cv2.warpPolar), getTrackedPointsKLT (cv2.calcOpticalFlowPyrLK incl. the
      pyramid), getBlobsFromCart (skimage blob_doh incl. response order and _prune_blobs' pair order),
      adaptiveNMS (NumPy 1.22 argsort tie order + ssc): all 257 feature rows of img/dead_reckoning/tiny_10.npz
      are reproduced (232 bit for bit, 25 within 0.01 px) and 2170 of 2192 ANMS selections drawn in
      img/blob/tiny/*.jpg (five of the eleven frames without a single difference).  The remainder is traced to
      +-1 grey-level pixels of the reference's warp: its OpenCV build takes the map radius from IPP's
      ippsMagnitude_32f rather than a correctly rounded sqrt; every identified pixel sits on a rounding tie of
      rho*32.  Not reproducible without that binary, and not a property of the algorithm.
  PINNED end to end by the poses the reference printed into img/roam_mapping/tiny_traj/*.jpg (tests/golden/tiny_traj.npz,
  make_tiny_traj.py; tests/test_oracle_tiny_traj.py): OdometryPipeline reproduces frames 1-3 (and 5, given the reference's
  frame-4 clique) to print precision; DESIGN.md §4 has the per-frame table of what accounts for the rest.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB = os.path.join(_BUILD, "liboracle.so")
_SRCS = ["peaks.c", "warp_klt.c", "clique.c", "lm.c", "ssc.c", "doh.c", "prune.c"]

RANGE_RESOLUTION_M = 0.0432          # parseData.py:9
RANGE_RESOLUTION_CART_M = 0.0864     # parseData.py:13
MAX_RANGE_CLIP_PX = int(87.5 / RANGE_RESOLUTION_M)   # 2025, parseData.py:14,49-51
DIST_THRESHOLD_PX = 0.5 / RANGE_RESOLUTION_CART_M    # outlierRejection.py:10-11
ERR_THRESHOLD = 10                   # getTransformKLT.py:84
RADAR_SCAN_FREQUENCY = 4             # motionDistortion.py:36


def build(force: bool = False) -> str:
    """Compile oracle/c/*.c into oracle/_build/liboracle.so (gcc only, a few seconds).  ORACLE_LIB: a library built elsewhere - the
    sanitizer build of profiles/asan_cpu.sh - is used as it is."""
    if os.environ.get("ORACLE_LIB"):
        return os.environ["ORACLE_LIB"]
    srcs = [os.path.join(_HERE, "c", s) for s in _SRCS if os.path.exists(os.path.join(_HERE, "c", s))]
    if not force and os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs):
        return _LIB
    os.makedirs(_BUILD, exist_ok=True)
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
           "-D_GNU_SOURCE", "-o", _LIB] + srcs + ["-lm"]
    subprocess.run(cmd, check=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_peaks_f32.restype = C.c_int64
        _lib.oracle_peaks_u8.restype = C.c_int64
        _lib.oracle_pairwise_sum_f32.restype = C.c_float
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ------------------------------------------------------------------ a1: record decode
def extractDataFromRadarImage(rec_u8: np.ndarray, maxRangeClipM: float = 87.5):
    """parseData.py:17-53 (numpy restatement; same 6-tuple)."""
    ts = rec_u8[:, :8].copy().view(np.int64)
    az = (rec_u8[:, 8:10].copy().view(np.uint16) / float(5600) * 2 * np.pi).astype(np.float32)
    valid = rec_u8[:, 10:11] == 255
    data = rec_u8[:, 11:].astype(np.float32) / 255.
    if maxRangeClipM > 0:
        data = data[:, :int(maxRangeClipM / RANGE_RESOLUTION_M)]
    return data, az, RANGE_RESOLUTION_M, az[1] - az[0], valid, ts


# ------------------------------------------------------------------ a2: polar peaks
def getPointCloudPolarInd(polarImage: np.ndarray) -> np.ndarray:
    """getPointCloud.py:11-54 on the float32 polar image -> (P,2) int64 [azimuthIdx, rangeIdx]."""
    img = np.ascontiguousarray(polarImage, dtype=np.float32)
    rows, cols = img.shape
    cap = rows * ((cols + 1) // 2)
    out = np.empty((cap, 2), np.int32)
    n = lib().oracle_peaks_f32(_p(img, C.c_float), rows, cols, C.c_int64(cols), _p(out, C.c_int32), C.c_int64(cap))
    return out[:n].astype(np.int64)


def peaks_from_record_u8(rec: np.ndarray, payload_off: int = 11, clip: int = MAX_RANGE_CLIP_PX) -> np.ndarray:
    rec = np.ascontiguousarray(rec, dtype=np.uint8)
    rows, stride = rec.shape
    cap = rows * ((clip + 1) // 2)
    out = np.empty((cap, 2), np.int32)
    n = lib().oracle_peaks_u8(_p(rec, C.c_uint8), rows, C.c_int64(stride), payload_off, clip,
                              _p(out, C.c_int32), C.c_int64(cap))
    return out[:n].astype(np.int64)


def pairwise_sum_f32(a: np.ndarray) -> np.float32:
    a = np.ascontiguousarray(a, np.float32)
    return np.float32(lib().oracle_pairwise_sum_f32(_p(a, C.c_float), C.c_int64(a.size)))


# ------------------------------------------------------------------ a3: polar -> Cartesian
def convertPolarImageToCartesian(imgPolar: np.ndarray, want_u8: bool = False):
    """parseData.py:100-135 (downsampleFactor=2): (rows,cols) f32 -> (2R,2R) f32, R=cols//2."""
    img = np.ascontiguousarray(imgPolar, dtype=np.float32)
    rows, cols = img.shape
    W = 2 * (cols // 2)
    cart = np.empty((W, W), np.float32)
    u8 = np.empty((W, W), np.uint8) if want_u8 else None
    lib().oracle_polar_to_cart(_p(img, C.c_float), rows, cols, C.c_int64(cols), _p(cart, C.c_float),
                               _p(u8, C.c_uint8) if want_u8 else None)
    return (cart, u8) if want_u8 else cart


def convertPolarImageToCartesianTies(imgPolar: np.ndarray, tie_eps: float, want_u8: bool = False, every: int = 1):
    """sensitivity probe: the same warp with every `every`-th radial sampling coordinate within tie_eps (in 1/32-px units) of a rounding tie
    rounded the OTHER way - what an ulp of difference in the radius (IPP's magnitude in the reference's cv2 build, DESIGN.md
    section 4) can do.  -> (cart[, u8], number of such pixels)"""
    img = np.ascontiguousarray(imgPolar, dtype=np.float32)
    rows, cols = img.shape
    W = 2 * (cols // 2)
    cart = np.empty((W, W), np.float32)
    u8 = np.empty((W, W), np.uint8) if want_u8 else None
    lib().oracle_polar_to_cart_ties.restype = C.c_int64
    n = lib().oracle_polar_to_cart_ties(_p(img, C.c_float), rows, cols, C.c_int64(cols), _p(cart, C.c_float),
                                        _p(u8, C.c_uint8) if want_u8 else None, C.c_float(tie_eps), int(every))
    return (cart, u8, int(n)) if want_u8 else (cart, int(n))


def quantize_u8(img: np.ndarray) -> np.ndarray:
    """(img*255).astype(np.uint8), getTransformKLT.py:356-357."""
    img = np.ascontiguousarray(img, np.float32)
    out = np.empty(img.shape, np.uint8)
    lib().oracle_quantize_u8(_p(img, C.c_float), C.c_int64(img.size), _p(out, C.c_uint8))
    return out


# ------------------------------------------------------------------ a7: pyramidal LK
def build_pyramid(img_u8: np.ndarray, max_level: int = 3):
    pyr = [np.ascontiguousarray(img_u8, np.uint8)]
    for _ in range(max_level):
        s = pyr[-1]
        h, w = s.shape
        d = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
        lib().oracle_pyr_down_u8(_p(s, C.c_uint8), w, h, _p(d, C.c_uint8))
        pyr.append(d)
    return pyr


def calcOpticalFlowPyrLK(prev_u8, next_u8, pts, winSize=15, maxLevel=3, maxCount=10, epsilon=0.03,
                         minEigThreshold=1e-4):
    """cv2.calcOpticalFlowPyrLK restatement -> (nextPts (K,2) f32, status (K,1) u8, err (K,1) f32)."""
    pp, npyr = build_pyramid(prev_u8, maxLevel), build_pyramid(next_u8, maxLevel)
    return klt_on_pyramids(pp, npyr, pts, winSize, maxCount, epsilon, minEigThreshold)


def klt_on_pyramids(pp, npyr, pts, winSize=15, maxCount=10, epsilon=0.03, minEigThreshold=1e-4):
    L = len(pp)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    K = pts.shape[0]
    arr_t = C.POINTER(C.c_uint8) * L
    P = arr_t(*[_p(a, C.c_uint8) for a in pp])
    N = arr_t(*[_p(a, C.c_uint8) for a in npyr])
    lw = (C.c_int * L)(*[a.shape[1] for a in pp])
    lh = (C.c_int * L)(*[a.shape[0] for a in pp])
    nxt = np.zeros((K, 2), np.float32)
    st = np.zeros((K,), np.uint8)
    err = np.zeros((K,), np.float32)
    lib().oracle_klt_track(P, N, lw, lh, L, _p(pts, C.c_float), K, winSize, maxCount,
                           C.c_float(epsilon), C.c_float(minEigThreshold),
                           _p(nxt, C.c_float), _p(st, C.c_uint8), _p(err, C.c_float))
    return nxt, st.reshape(-1, 1), err.reshape(-1, 1)


def getTrackedPointsKLT(srcImg, targetImg, blobCoordSrc):
    """getTransformKLT.py:317-381 without the internal re-detect branch (:348-352)."""
    pts = np.ascontiguousarray(blobCoordSrc[:, :2]).astype(np.float32)
    s8, t8 = quantize_u8(srcImg), quantize_u8(targetImg)
    nxt, status, err = calcOpticalFlowPyrLK(s8, t8, pts)
    status &= (err < ERR_THRESHOLD)
    good = (status == 1).flatten()
    return nxt[good], pts[good], nxt[~good], pts[~good], status


# ------------------------------------------------------------------ a8: outlier rejection
def consistency_graph(prev, new, thr=DIST_THRESHOLD_PX):
    prev = np.ascontiguousarray(prev, np.float32)
    new = np.ascontiguousarray(new, np.float32)
    K = prev.shape[0]
    nw = max(1, (K + 63) // 64)
    adj = np.zeros((K, nw), np.uint64)
    lib().oracle_consistency_graph(_p(prev, C.c_float), _p(new, C.c_float), K, C.c_double(thr),
                                   _p(adj, C.c_uint64), nw)
    return adj


def max_clique_lex(adj: np.ndarray):
    K, nw = adj.shape
    mask = np.zeros(K, np.uint8)
    nodes = C.c_int64(0)
    size = lib().oracle_max_clique_lex(_p(np.ascontiguousarray(adj), C.c_uint64), K, nw,
                                       _p(mask, C.c_uint8), C.byref(nodes))
    return size, mask.astype(bool), nodes.value


def max_clique_nx(adj: np.ndarray, prune: bool = True):
    """-> (size, mask, stats): the first strictly-largest clique in networkx.find_cliques order (the reference's choice,
    outlierRejection.py:63-75); stats = (maximal cliques yielded, children skipped, existence searches, tree nodes)."""
    K, nw = adj.shape
    mask = np.zeros(max(K, 1), np.uint8)
    stats = np.zeros(4, np.int64)
    lib().oracle_max_clique_nx.restype = C.c_int
    size = lib().oracle_max_clique_nx(_p(np.ascontiguousarray(adj), C.c_uint64), K, nw, int(bool(prune)),
                                      _p(mask, C.c_uint8), _p(stats, C.c_int64))
    return size, mask[:K].astype(bool), tuple(int(v) for v in stats)


def max_cliques_nx_all(adj: np.ndarray, cap: int = 64):
    """every maximum clique, in networkx.find_cliques order -> bool (n, K)"""
    K, nw = adj.shape
    masks = np.zeros((cap, max(K, 1)), np.uint8)
    lib().oracle_max_cliques_nx_all.restype = C.c_int
    n = lib().oracle_max_cliques_nx_all(_p(np.ascontiguousarray(adj), C.c_uint64), K, nw, _p(masks, C.c_uint8), cap)
    return masks[:n, :K].astype(bool)


def pyset_program(ops, keys, spans, out_cap=1 << 20):
    """run a little program of CPython-set operations on the C restatement (tests/test_oracle_clique_order.py)"""
    ops = np.ascontiguousarray(ops, np.int32).reshape(-1, 3)
    keys = np.ascontiguousarray(keys, np.int32)
    spans = np.ascontiguousarray(spans, np.int32).reshape(-1, 2)
    out = np.zeros(out_cap, np.int32)
    lib().oracle_pyset_program.restype = C.c_int64
    n = lib().oracle_pyset_program(_p(ops, C.c_int32), len(ops), _p(keys, C.c_int32), _p(spans, C.c_int32), _p(out, C.c_int32), out_cap)
    assert n <= out_cap
    return out[:n]


def rejectOutliers(prev_coord, new_coord):
    """outlierRejection.py:16-95 -> (pruned_prev, pruned_new, mask bool (K,)): the reference's clique, ties included."""
    assert prev_coord.shape == new_coord.shape, "Coordinates should be the same shape"
    K = prev_coord.shape[0]
    if K == 0:
        return prev_coord, new_coord, np.zeros(0, bool)
    _, mask, _ = max_clique_nx(consistency_graph(prev_coord, new_coord))
    return prev_coord[mask], new_coord[mask], mask


def adjacency_dense(adj: np.ndarray, K: int) -> np.ndarray:
    bits = np.unpackbits(adj.view(np.uint8), axis=1, bitorder="little")
    return bits[:, :K].astype(bool)


def max_clique_bruteforce(A: np.ndarray):
    """Tiny independent checker: enumerate maximal cliques (Bron-Kerbosch with pivot, pure
    Python) and return the lexicographically smallest maximum one.  Small K only."""
    K = A.shape[0]
    nb = [set(np.flatnonzero(A[i]).tolist()) - {i} for i in range(K)]
    best = [[]]

    def bk(R, P, X):
        if not P and not X:
            r = sorted(R)
            if len(r) > len(best[0]) or (len(r) == len(best[0]) and r < best[0]):
                best[0] = r
            return
        u = max(P | X, key=lambda v: len(P & nb[v]))
        for v in list(P - nb[u]):
            bk(R | {v}, P & nb[v], X & nb[v])
            P.remove(v)
            X.add(v)

    bk(set(), set(range(K)), set())
    m = np.zeros(K, bool)
    m[best[0]] = True
    return len(best[0]), m


# ------------------------------------------------------------------ a10: 2-D Kabsch
def calculateTransformSVD(srcCoords, targetCoords):
    """getTransformKLT.py:129-162 restated in float64: src ~= R tgt + h.  (The reference
    runs the same formulas in the input dtype; f32 inputs differ by ~3e-5 m)"""
    s = np.asarray(srcCoords, np.float64)
    t = np.asarray(targetCoords, np.float64)
    ms, mt = s.mean(axis=0, keepdims=True), t.mean(axis=0, keepdims=True)
    Cm = (s - ms).T @ (t - mt)
    U, _, Vt = np.linalg.svd(Cm)
    D = np.eye(2)
    D[1, 1] = np.linalg.det(U @ Vt)
    R = U @ D @ Vt
    h = ms - (R @ mt.T).T
    return R, h.T


def kabsch_closed_form(src, tgt):
    """Same fit without the SVD: theta = atan2(C01 - C10, C00 + C11) of C = sum (t-mt)(s-ms)^T
    arrangement used by the device kernel; equals the SVD solution whenever C != 0."""
    s = np.asarray(src, np.float64)
    t = np.asarray(tgt, np.float64)
    ms, mt = s.mean(axis=0), t.mean(axis=0)
    a, b = s - ms, t - mt
    sxx = (a[:, 0] * b[:, 0]).sum(); sxy = (a[:, 0] * b[:, 1]).sum()
    syx = (a[:, 1] * b[:, 0]).sum(); syy = (a[:, 1] * b[:, 1]).sum()
    th = np.arctan2(syx - sxy, sxx + syy)
    c, sn = np.cos(th), np.sin(th)
    R = np.array([[c, -sn], [sn, c]])
    h = ms - R @ mt
    return R, h.reshape(2, 1)


# ------------------------------------------------------------------ a11-a14: motion distortion
class MotionDistortionSolver:
    """motionDistortion.py:38-325 (live 3-arg constructor :70-78)."""

    def __init__(self, sigma_p, sigma_v, frequency=RADAR_SCAN_FREQUENCY):
        self.total_scan_time = 1 / frequency
        self.sigma_p = np.diag(sigma_p).astype(np.float64)
        self.sigma_v = np.diag(sigma_v).astype(np.float64)

    def update_problem(self, T_wj0, p_w, p_jt, T_wj, debug=False):
        assert p_w.shape == p_jt.shape
        self.T_wj0 = np.asarray(T_wj0, np.float64)
        self.T_wj0_inv = np.linalg.inv(self.T_wj0)
        self.p_w = np.ascontiguousarray(p_w[:, :2], np.float64)
        self.p_jt = np.ascontiguousarray(p_jt[:, :2], np.float64)
        self.T_wj_initial = np.ascontiguousarray(T_wj, np.float64)
        self.dT = self.compute_time_deltas(self.total_scan_time, self.p_jt)

    @staticmethod
    def compute_time_deltas(period, points):
        pts = np.ascontiguousarray(points[:, :2], np.float64)
        dT = np.empty(pts.shape[0])
        lib().oracle_time_deltas(_p(pts, C.c_double), pts.shape[0], C.c_double(period), _p(dT, C.c_double))
        return dT

    @staticmethod
    def undistort(v_j, points, period=1 / RADAR_SCAN_FREQUENCY, times=None):
        pts = np.ascontiguousarray(points[:, :2], np.float64)
        v = np.ascontiguousarray(v_j, np.float64)
        out = np.empty((pts.shape[0], 3))
        xy = np.empty((pts.shape[0], 2))
        lib().oracle_undistort(_p(v, C.c_double), _p(pts, C.c_double), pts.shape[0], C.c_double(period),
                               _p(xy, C.c_double))
        out[:, :2] = xy
        out[:, 2] = 1.0
        return out

    def _solve(self, want_r0=False):
        N = self.p_w.shape[0]
        sig = np.concatenate((self.sigma_p, self.sigma_v)).astype(np.float64)
        out = np.empty(6)
        x0 = np.empty(6)
        r0 = np.empty(2 * N + 3)
        nfev = C.c_int(0)
        T0 = np.ascontiguousarray(self.T_wj0)
        info = lib().oracle_mds_solve(_p(T0, C.c_double), _p(self.p_w, C.c_double), _p(self.p_jt, C.c_double), N,
                                      _p(self.T_wj_initial, C.c_double), _p(sig, C.c_double),
                                      C.c_double(self.total_scan_time), _p(out, C.c_double), C.byref(nfev),
                                      _p(x0, C.c_double), _p(r0, C.c_double))
        self.nfev, self.info = nfev.value, info
        return out, x0, r0

    def optimize_library(self):
        return self._solve()[0]


# ------------------------------------------------------------------ a5/a6: ANMS + dedupe
def ssc(keypoints, num_ret_points, tolerance, cols, rows):
    """ANMS.py:5-102 -> selected rows of `keypoints` (same order)."""
    kp = np.ascontiguousarray(keypoints, np.float64)
    B = kp.shape[0]
    sel = np.empty(max(B, 1), np.int32)
    n = lib().oracle_ssc(_p(kp, C.c_double), B, int(num_ret_points), C.c_double(tolerance), int(cols), int(rows),
                         _p(sel, C.c_int32))
    return kp[sel[:n]]


def doh_maxima(img, sigma_list, threshold, cap=1 << 18):
    """integral image + Hessian determinants + 3x3x3 maxima (oracle/c/doh.c) -> (rcs, values)"""
    img = np.ascontiguousarray(img, np.float32)
    H, W = img.shape
    S = np.empty((H, W), np.float64)
    lib().oracle_integral_image(_p(img, C.c_float), H, W, _p(S, C.c_double))
    layers = []
    for s in sigma_list:
        if int(3 * s) <= 0:
            layers.append(None)
            continue
        d = np.empty((H, W), np.float64)
        lib().oracle_hessian_det(_p(S, C.c_double), H, W, C.c_double(float(s)), _p(d, C.c_double))
        layers.append(d)
    arr = (C.POINTER(C.c_double) * len(layers))(*[(_p(l, C.c_double) if l is not None else None) for l in layers])
    rcs = np.empty((cap, 3), np.int32)
    val = np.empty(cap, np.float64)
    lib().oracle_doh_maxima.restype = C.c_int64
    n = lib().oracle_doh_maxima(arr, len(layers), H, W, C.c_double(float(threshold)), _p(rcs, C.c_int32), _p(val, C.c_double),
                                C.c_int64(cap))
    assert n <= cap
    return rcs[:n], val[:n], layers


def ckdtree_pairs(pts, r):
    """pairs (P,2) int64 in the order scipy.spatial.cKDTree(pts).query_pairs(r) emits them, and the
    tree's index permutation (restated in oracle/c/prune.c; checked against live scipy in the tests)"""
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
    n = pts.shape[0]
    out = C.POINTER(C.c_int64)()
    idx = np.empty(max(n, 1), np.int64)
    lib().oracle_ckdtree_pairs.restype = C.c_int64
    m = lib().oracle_ckdtree_pairs(_p(pts, C.c_double), C.c_int64(n), C.c_double(float(r)), C.byref(out), _p(idx, C.c_int64))
    pairs = np.ctypeslib.as_array(out, shape=(m, 2)).copy() if m else np.empty((0, 2), np.int64)
    lib().oracle_free(out)
    return pairs, idx[:n]


def pyset_order(pairs):
    """iteration order (indices into pairs) of the Python set built by adding the (i, j) tuples in order"""
    pairs = np.ascontiguousarray(pairs, np.int64).reshape(-1, 2)
    order = np.empty(max(len(pairs), 1), np.int64)
    lib().oracle_pyset_order.restype = C.c_int64
    m = lib().oracle_pyset_order(_p(pairs, C.c_int64), C.c_int64(len(pairs)), _p(order, C.c_int64))
    return order[:m]


def prune_blobs(blobs, overlap=0.5):
    """skimage.feature.blob._prune_blobs in scikit-image's own pair order (cKDTree.query_pairs -> Python set)"""
    bl = np.ascontiguousarray(blobs, np.float64).copy()
    lib().oracle_prune_blobs.restype = C.c_int64
    lib().oracle_prune_blobs(_p(bl, C.c_double), C.c_int64(len(bl)), C.c_double(overlap))
    return bl[bl[:, 2] > 0]


def argsort_numpy122(v):
    """np.argsort(v) as numpy 1.22.3 (the reference's pin) computes it: npy_aquicksort, unstable"""
    v = np.ascontiguousarray(v, np.float64)
    out = np.empty(max(len(v), 1), np.int64)
    lib().oracle_aquicksort_f64(_p(v, C.c_double), C.c_int64(len(v)), _p(out, C.c_int64))
    return out[:len(v)]


def blob_doh(image, min_sigma=1, max_sigma=30, num_sigma=10, threshold=0.01, overlap=0.5):
    """skimage.feature.blob_doh restatement (getFeatures.py:47-51): maxima ordered by response
    (peak_local_max), then _prune_blobs in scikit-image's pair order."""
    sig = np.linspace(min_sigma, max_sigma, num_sigma)
    rcs, val, _ = doh_maxima(image, sig, threshold)
    if len(rcs) == 0:
        return np.empty((0, 3))
    idx = np.argsort(-val, kind="stable")            # no ties on real images; C (row, col, sigma) order breaks them
    bl = rcs[idx].astype(np.float64)
    bl[:, 2] = sig[rcs[idx][:, 2]]
    return prune_blobs(bl, overlap)


def getFeatures(img):
    """getFeatures.py:74-95 with DEFAULT_FEATURE_PARAMS (:13-18)."""
    blobs = blob_doh(np.asarray(img, np.float64), min_sigma=0.01, max_sigma=10, num_sigma=3, threshold=.0005)
    blobs = adaptiveNMS(img.shape, blobs)
    return np.fliplr(blobs[:, :2]), blobs[:, 2]


def adaptiveNMS(img_shape, blobs, ret_points=200, tolerance=0.1):
    """getFeatures.py:66-72."""
    H, W = img_shape
    kp = blobs[argsort_numpy122(blobs[:, 2])]
    return ssc(kp, ret_points, tolerance, W, H)


def append_dedupe(oldFeaturesCoord, newFeatureCoord):
    """getFeatures.py:109-112: vstack, drop exact duplicate rows keeping first occurrence."""
    pts = np.vstack((oldFeaturesCoord, newFeatureCoord))
    _, idx = np.unique(pts, axis=0, return_index=True)
    return np.ascontiguousarray(pts[np.sort(idx)]).astype(np.float32)


# ------------------------------------------------------------------ SE(2) helpers (utils.py)
def normalize_angles(th):
    return (th + np.pi) % (2 * np.pi) - np.pi


def convertPoseToTransform(pose):
    x, y, th = pose
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, -s, x], [s, c, y], [0, 0, 1.0]])


def convertTransformToPose(T):
    return np.array([T[0, 2], T[1, 2], np.arctan2(T[1, 0], T[0, 0])])


# ------------------------------------------------------------------ f4: Fourier-Mellin rotation prior (FMT.py)
FMT_DOWNSAMPLE_FACTOR = 10           # FMT.py:10
FMT_RANGE_CLIP_M = 87.5              # FMT.py:11


def _cv_resize_cols_linear(img, new_w):
    """cv2.resize(img, (new_w, H)) with INTER_LINEAR for float32 (rows unchanged -> the vertical weights are (1, 0))"""
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    scale = 1.0 / (float(new_w) / w)
    dx = np.arange(new_w)
    fx = ((dx + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(np.float32)).astype(np.float32)
    lo, hi = sx < 0, sx >= w - 1
    fx[lo | hi] = 0
    sx[lo] = 0
    sx[hi] = w - 1
    a1 = fx
    a0 = (np.float32(1) - fx).astype(np.float32)
    s1 = np.minimum(sx + 1, w - 1)
    return (img[:, sx] * a0 + img[:, s1] * a1).astype(np.float32)


def _get_optimal_dft_size(n):
    """smallest 2^a 3^b 5^c >= n (cv2.getOptimalDFTSize)"""
    best = None
    p2 = 1
    while p2 < 2 * n:
        p3 = p2
        while p3 < 2 * n:
            p5 = p3
            while p5 < 2 * n:
                if p5 >= n and (best is None or p5 < best):
                    best = p5
                p5 *= 5
            p3 *= 3
        p2 *= 2
    return best


def convertPolarImgToLogPolar(imgPolar):
    """parseData.py:138-160: polar -> Cartesian (downsampleFactor 1) -> log-polar of OpenCV's default size"""
    img = np.ascontiguousarray(imgPolar, np.float32)
    rows, cols = img.shape
    R = cols                                           # maxRadius = h (parseData.py:118-121 with downsampleFactor 1)
    cart = np.empty((2 * R, 2 * R), np.float32)
    lib().oracle_warp_polar_inverse(_p(img, C.c_float), rows, cols, C.c_int64(cols), 2 * R, 2 * R, C.c_float(R), C.c_float(R), C.c_double(R), _p(cart, C.c_float))
    maxRadius = (2 * R) / 2                            # convertCartesianImageToPolar: center (h/2, w/2), maxRadius w/2, dsize from OpenCV
    dw, dh = int(np.rint(maxRadius)), int(np.rint(maxRadius * np.pi))
    lp = np.empty((dh, dw), np.float32)
    lib().oracle_warp_polar_forward_log(_p(cart, C.c_float), 2 * R, 2 * R, dw, dh, C.c_float(R), C.c_float(R), C.c_double(maxRadius), _p(lp, C.c_float))
    return lp


def phaseCorrelate(src1, src2):
    """cv2.phaseCorrelate(src1, src2, cv2.createHanningWindow(...)) restated with numpy FFTs in float64 -> ((dx, dy), response)"""
    H, W = src1.shape
    wc = 0.5 * (1.0 - np.cos(2.0 * np.pi / (W - 1) * np.arange(W)))
    wr = 0.5 * (1.0 - np.cos(2.0 * np.pi / (H - 1) * np.arange(H)))
    win = np.sqrt((wr[:, None] * wc[None, :]).astype(np.float32))
    M, N = _get_optimal_dft_size(H), _get_optimal_dft_size(W)
    a = np.zeros((M, N), np.float32); b = np.zeros((M, N), np.float32)
    a[:H, :W] = (win * src1).astype(np.float32); b[:H, :W] = (win * src2).astype(np.float32)
    F1, F2 = np.fft.fft2(a.astype(np.float64)), np.fft.fft2(b.astype(np.float64))
    P = F1 * np.conj(F2)
    mag = np.abs(P)
    eps32 = float(np.finfo(np.float32).eps)
    Cc = np.real(np.fft.ifft2(P * mag / (mag * mag + eps32)))      # divSpectrums(P, |P|): P |P| / (|P|^2 + FLT_EPSILON)
    Cc = np.fft.fftshift(Cc)
    py, px = np.unravel_index(np.argmax(Cc), Cc.shape)
    r0, r1 = max(py - 2, 0), min(py + 2, M - 1)
    c0, c1 = max(px - 2, 0), min(px + 2, N - 1)
    box = Cc[r0:r1 + 1, c0:c1 + 1]
    s = box.sum()
    ys, xs = np.mgrid[r0:r1 + 1, c0:c1 + 1]
    cx, cy = (xs * box).sum() / (s + np.finfo(float).eps), (ys * box).sum() / (s + np.finfo(float).eps)
    return (N / 2.0 - cx, M / 2.0 - cy), s


def getRotationUsingFMT(srcPolarImg, targetPolarImg, downsampleFactor=FMT_DOWNSAMPLE_FACTOR, maxRangeClipM=FMT_RANGE_CLIP_M):
    """FMT.py:36-90 -> (angle rad, scale, response)"""
    assert srcPolarImg.shape == targetPolarImg.shape
    if maxRangeClipM > 0:
        clip = int(maxRangeClipM / RANGE_RESOLUTION_CART_M)
        srcPolarImg, targetPolarImg = srcPolarImg[:, :clip], targetPolarImg[:, :clip]
    H, W = srcPolarImg.shape
    a = convertPolarImgToLogPolar(_cv_resize_cols_linear(srcPolarImg, int(W // downsampleFactor)))
    b = convertPolarImgToLogPolar(_cv_resize_cols_linear(targetPolarImg, int(W // downsampleFactor)))
    (scale, angle), resp = phaseCorrelate(a, b)
    H_lp, W_lp = a.shape
    sz = max(H_lp, W_lp)
    angle = normalize_angles(-float(angle) * 2 * np.pi / sz)
    log_base = np.exp(np.log(H_lp / 2) / sz)
    return float(angle), float(log_base ** scale), float(resp)     # (OpenCV: unscaled IDFT sum / (M N) = numpy's scaled sum)


# ------------------------------------------------------------------ a9: Tracker.track glue
def track_glue(klt_out, do_reject=True):
    """Tracker.py:75-104 given the 5-tuple of getTrackedPointsKLT."""
    good_new, good_old, bad_new, bad_old, corrStatus = klt_out
    nFeatures = good_new.shape[0] + bad_new.shape[0]
    if do_reject:
        good_old, good_new, mask = rejectOutliers(good_old, good_new)
        rng = np.arange(nFeatures)
        corrStatus = corrStatus.copy()
        corrStatus[rng[corrStatus.flatten().astype(bool)]] &= mask[:, np.newaxis]
    return good_old, good_new, 0.0, corrStatus


# ------------------------------------------------------------------ a15: the per-scan loop body
RADAR_CART_CENTER = np.array([1012, 1012])   # RawROAMSystem.py:16
N_FEATURES_BEFORE_RETRACK = 60               # getFeatures.py:57 (import-time binding, RawROAMSystem.py:7)
ROT_THRESHOLD = 0.2                          # Mapping.py:13
TRANS_THRESHOLD_SQ = 4.0                     # Mapping.py:14-15


class Keyframe:
    """Mapping.Keyframe (Mapping.py:21-125), fields the loop uses."""

    def __init__(self, pose, featurePointsLocal, record_u8, velocity, with_peaks=True):
        self.updateInfo(pose, featurePointsLocal, record_u8, velocity, with_peaks)

    def updateInfo(self, pose, featurePointsLocal, record_u8, velocity, with_peaks=True):
        self.pose = np.asarray(pose, np.float64)
        self.featurePointsLocal = featurePointsLocal
        if with_peaks and record_u8 is not None:
            self.pointCloud = peaks_from_record_u8(record_u8)               # Mapping.py:62
        self.velocity = velocity
        self.prunedUndistortedLocals = MotionDistortionSolver.undistort(velocity, featurePointsLocal)[:, :2]

    def pruneFeaturePoints(self, corrStatus):
        self.prunedUndistortedLocals = self.prunedUndistortedLocals[corrStatus.flatten().astype(bool)]

    def getPrunedFeaturesGlobalPosition(self):
        x, y, th = self.pose
        c, s = np.cos(th), np.sin(th)
        R = np.array([[c, -s], [s, c]])
        return (R @ self.prunedUndistortedLocals.T + np.array([[x], [y]])).T


class OdometryPipeline:
    """CPU restatement of the body of RawROAMSystem.run (RawROAMSystem.py:139-298) without
    plotting: one call to step() = one scan pair.  Feature (re)detection is injected
    (`detect(cart_f32) -> (k,2) [x,y]`) because blob_doh is a separate, unpinned stage."""

    def __init__(self, first_record_u8, init_features_xy, init_pose, reject_outliers=True,
                 motion_distortion=True, detect=None, payload_off=11, clip=MAX_RANGE_CLIP_PX, warp=None,
                 keyframe_trans_m=None, keyframe_rot_rad=None):
        self.off, self.clip = payload_off, clip
        # Map.isGoodKeyframe's thresholds (Mapping.py:13-15, 149-174); None = HEAD's constants
        self.kf_trans_sq = TRANS_THRESHOLD_SQ if keyframe_trans_m is None else float(keyframe_trans_m) ** 2
        self.kf_rot = ROT_THRESHOLD if keyframe_rot_rad is None else float(keyframe_rot_rad)
        self.warp = warp or (lambda polar: convertPolarImageToCartesian(polar, want_u8=True))     # polar f32 -> (cart f32, cart u8)
        self.reject, self.md, self.detect = reject_outliers, motion_distortion, detect
        self.MDS = MotionDistortionSolver(np.diag([4, 4]), np.diag([1, 1, (5 * np.pi / 180) ** 2]))
        self.prev_pose = convertPoseToTransform(init_pose)
        self.pose = np.asarray(init_pose, np.float64)
        self.prevCart8 = self._cart_u8(first_record_u8)
        self.prevPyr = build_pyramid(self.prevCart8, 3)
        self.blobCoord = np.ascontiguousarray(init_features_xy, np.float32)
        metric = (self.blobCoord - RADAR_CART_CENTER) * RANGE_RESOLUTION_CART_M
        self.old_kf = Keyframe(self.pose, metric, first_record_u8, np.zeros(3))
        self.last = {}

    def _cart_u8(self, rec):
        polar = rec[:, self.off:self.off + self.clip].astype(np.float32) / 255.
        return self.warp(polar)[1]

    def step(self, rec_u8):
        cur8 = self._cart_u8(rec_u8)
        curPyr = build_pyramid(cur8, 3)
        pts = self.blobCoord
        nxt, status, err = klt_on_pyramids(self.prevPyr, curPyr, pts)
        status &= (err < ERR_THRESHOLD)
        good = (status == 1).flatten()
        klt_out = (nxt[good], pts[good], nxt[~good], pts[~good], status)
        n_good = int(good.sum())
        if self.reject:
            good_old, good_new, _, corrStatus = track_glue(klt_out, True)
        else:
            good_old, good_new, corrStatus = pts[good], nxt[good], status
        self.old_kf.pruneFeaturePoints(corrStatus)
        n = good_new.shape[0]
        peaks = peaks_from_record_u8(rec_u8, self.off, self.clip)
        out = dict(n_tracked=len(pts), n_good=n_good, n_inliers=n, n_peaks=len(peaks))
        if n >= 2:
            R, h = calculateTransformSVD(good_old, good_new)
            h = h * RANGE_RESOLUTION_CART_M
            centered_new = (good_new - RADAR_CART_CENTER) * RANGE_RESOLUTION_CART_M
            if self.md:
                p_w = self.old_kf.getPrunedFeaturesGlobalPosition()
                T_wj = self.prev_pose @ np.block([[R, h], [np.zeros((2,)), 1]])
                self.MDS.update_problem(self.prev_pose, p_w, centered_new, T_wj)
                sol = self.MDS.optimize_library()
                pose, velocity = sol[3:], sol[:3]
            else:
                x, y, th = self.pose
                dx, dy, dth = h[0, 0], h[1, 0], np.arctan2(R[1, 0], R[0, 0])
                pose = np.array([x + dx * np.cos(th) - dy * np.sin(th), y + dx * np.sin(th) + dy * np.cos(th), th + dth])
                velocity = np.zeros(3)
            out.update(R=R, h=h)
        else:
            pose, velocity = self.pose, np.zeros(3)
            centered_new = (good_new - RADAR_CART_CENTER) * RANGE_RESOLUTION_CART_M
        retrack = n <= N_FEATURES_BEFORE_RETRACK
        dth = abs(self.old_kf.pose[2] - pose[2])
        dtr = ((self.old_kf.pose[:2] - pose[:2]) ** 2).sum()
        newkf = retrack or dth >= self.kf_rot or dtr >= self.kf_trans_sq
        if newkf:
            if retrack and self.detect is not None:
                cart_f32 = self.warp(rec_u8[:, self.off:self.off + self.clip].astype(np.float32) / 255.)[0]
                good_new = append_dedupe(good_new, self.detect(cart_f32))
                centered_new = (good_new - RADAR_CART_CENTER) * RANGE_RESOLUTION_CART_M
            self.old_kf = Keyframe(pose, centered_new, None, velocity, with_peaks=False)
        out.update(pose=np.asarray(pose, np.float64), velocity=np.asarray(velocity, np.float64), new_keyframe=newkf,
                   retrack=retrack, peaks=peaks)
        self.blobCoord = np.ascontiguousarray(good_new, np.float32)
        self.prevCart8, self.prevPyr = cur8, curPyr
        self.pose = np.asarray(pose, np.float64)
        self.prev_pose = convertPoseToTransform(self.pose)
        self.last = out
        return out
