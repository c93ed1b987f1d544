"""Seeded input generators shared by make_goldens.py (build container) and the tests
(both containers).  Pure numpy Generator streams: the same seed yields the same bytes
on both boxes (same image, numpy 2.2); the goldens also carry sha256 of the inputs they
were produced from so a drift would be caught, not silently compared.

The recipes follow the reference's ad-hoc fixtures:
  rigid correspondences      genFakeData.py:80-110   (src = R tgt + h convention, getTransformKLT.py:160)
  gross outliers at +-2*thr  outlierRejection.py:125-130, genFakeData.py:194-223
  scan distortion            genFakeData.py:153-180 / motionDistortion.py:107-153
"""
import numpy as np

CART = 2024
CENTER = 1012.0
M_PER_PX = 0.0864
THR_PX = 0.5 / M_PER_PX


def synthetic_polar_u8(seed: int, rows: int = 400, cols: int = 2025) -> np.ndarray:
    """Oxford-like clipped polar scan as u8 codes: exponential speckle (real 'tiny' scans:
    mean 11.5, p50 6, p90 29, p99 75), Gaussian reflector blobs (peak 60..136), plus the
    edge cases SciPy's plateau rule cares about: flat-top plateaus, all-zero rows,
    constant rows, maxima at either row end, saturated runs."""
    rng = np.random.default_rng(1000 + seed)
    img = rng.exponential(11.0, size=(rows, cols))
    nb = 300
    az = rng.uniform(0, rows, nb)
    rg = rng.uniform(30, cols - 5, nb)
    amp = rng.uniform(60, 136, nb)
    aa = np.arange(rows)[:, None]
    for a0, r0, A in zip(az, rg, amp):
        r_lo, r_hi = int(max(0, r0 - 15)), int(min(cols, r0 + 16))
        da = np.minimum(np.abs(aa - a0), rows - np.abs(aa - a0))
        sel = (da[:, 0] < 6)
        rr = np.arange(r_lo, r_hi)[None, :]
        img[sel, r_lo:r_hi] += A * np.exp(-0.5 * (da[sel] / 1.5) ** 2) * np.exp(-0.5 * ((rr - r0) / 3.0) ** 2)
    u8 = np.clip(np.floor(img), 0, 255).astype(np.uint8)
    # edge cases
    u8[3, :] = 0                                   # empty row -> no peaks, NaN threshold
    u8[4, :] = 17                                  # constant row -> one plateau touching both ends
    u8[5, :] = 0
    u8[5, 200:207] = 90                            # single even-length-minus-one plateau
    u8[5, 300:304] = 90                            # single even-length plateau -> midpoint floor
    u8[6, 0] = 255                                 # maximum at the left end: never a peak
    u8[6, 1] = 3
    u8[7, cols - 1] = 255                          # maximum at the right end: never a peak
    u8[8, cols - 3:cols] = 200                     # plateau running into the right end: not a peak
    u8[9, 0:3] = 200                               # plateau starting at the left end: not a peak
    u8[10, :] = (np.arange(cols) % 2) * 40         # 1012 equal-height peaks: thresh == height
    u8[11, :] = (np.arange(cols) % 2) * 40
    u8[11, 1001] = 41                              # ... with one slightly higher
    u8[12, 100:1900] = 255                         # long saturated run (one plateau peak)
    return u8


def ssc_keypoints(B: int, seed: int, clustered: bool = False) -> np.ndarray:
    """(B,3) f64 rows [row, col, sigma], already sorted ascending by sigma the way
    getFeatures.adaptiveNMS (getFeatures.py:66-72) hands them to ssc."""
    rng = np.random.default_rng(2000 + seed)
    if clustered == "lattice":
        # 20 x 20 lattice, pitch 100 px: the selected count jumps 100 <-> 400 and never
        # lands in [180,220] -> ssc leaves through its width-repeat / low>high exit
        g = np.arange(20) * 100.0 + 50.0
        rc = np.array([(r, c) for r in g for c in g])
        B = len(rc)
    elif clustered:
        c = rng.uniform(200, 1800, size=(12, 2))
        idx = rng.integers(0, 12, B)
        rc = c[idx] + rng.normal(0, 35, size=(B, 2))
        rc = np.clip(np.rint(rc), 0, CART - 1)
    else:
        rc = rng.integers(0, CART, size=(B, 2)).astype(np.float64)
    sig = rng.choice(np.array([0.01, 5.005, 10.0]), size=B)
    kp = np.column_stack((rc, sig))
    return kp[np.argsort(kp[:, 2], kind="stable")]


def rigid_pairs(n: int, seed: int, noise: float = 0.0):
    """src = R(theta) tgt + h (+ noise) in pixels; returns (src, tgt, theta, h)."""
    rng = np.random.default_rng(3000 + seed)
    tgt = rng.uniform(0, CART, size=(n, 2))
    theta = rng.uniform(-0.35, 0.35)
    h = np.array([rng.uniform(-35, 35), rng.uniform(-3, 3)])
    c, s = np.cos(theta), np.sin(theta)
    R = np.array([[c, -s], [s, c]])
    src = tgt @ R.T + h
    if noise > 0:
        src = src + rng.normal(0, noise, size=src.shape)
    return src, tgt, theta, h


def unique_clique_pairs(K: int, seed: int, outlier_frac: float):
    """(prev, new, inlier_mask) f32 pixel correspondences whose consistency graph has a
    UNIQUE maximum clique = the inlier set: inlier noise 0.3 px << thr (5.787 px), each
    outlier displaced by its own 40..120 px vector."""
    rng = np.random.default_rng(4000 + seed)
    prev = rng.uniform(150, CART - 150, size=(K, 2))
    theta = rng.uniform(-0.1, 0.1)
    c, s = np.cos(theta), np.sin(theta)
    R = np.array([[c, -s], [s, c]])
    t = np.array([rng.uniform(-25, 25), rng.uniform(-2, 2)])
    new = (prev - CENTER) @ R.T + CENTER + t + rng.normal(0, 0.3, size=(K, 2))
    n_out = int(round(K * outlier_frac))
    out_idx = rng.choice(K, size=n_out, replace=False)
    ang = rng.uniform(0, 2 * np.pi, n_out)
    mag = rng.uniform(40, 120, n_out)
    new[out_idx] += np.column_stack((mag * np.cos(ang), mag * np.sin(ang)))
    inl = np.ones(K, dtype=bool)
    inl[out_idx] = False
    return prev.astype(np.float32), new.astype(np.float32), inl


def _se2(x, y, th):
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, -s, x], [s, c, y], [0, 0, 1.0]])


def mds_problem(N: int, seed: int, yaw_per_frame: float, big: bool = False):
    """A motion-distortion problem in metres: returns (T_wj0, p_w (N,2), p_jt (N,2),
    T_wj_initial, truth6 = [vx,vy,vth,x,y,th])."""
    rng = np.random.default_rng(5000 + seed)
    T0 = _se2(rng.uniform(-200, 200), rng.uniform(-200, 200), rng.uniform(-np.pi, np.pi)) if big \
        else _se2(rng.uniform(-20, 20), rng.uniform(-20, 20), rng.uniform(-0.5, 0.5))
    dx, dy = rng.uniform(0.5, 2.5), rng.uniform(-0.1, 0.1)
    delta = _se2(dx, dy, yaw_per_frame)
    Tt = T0 @ delta
    v = np.array([dx, dy, yaw_per_frame]) / 0.25
    p_j = rng.uniform(-80, 80, size=(N, 2))              # undistorted, in frame j
    ang = np.arctan2(-p_j[:, 1], -p_j[:, 0])
    dT = 0.25 * ang / (2 * np.pi)
    p_jt = np.empty_like(p_j)
    for i in range(N):                                   # p_jt = SE2(v dT)^-1 p_j
        Ti = _se2(*(v * dT[i]))
        p_jt[i] = (np.linalg.inv(Ti) @ np.array([p_j[i, 0], p_j[i, 1], 1.0]))[:2]
    p_jt += rng.normal(0, 0.05, size=p_jt.shape)
    p_w = (Tt @ np.column_stack((p_j, np.ones(N))).T).T[:, :2]
    n_bad = max(1, N // 20)                              # a few gross mismatches for the Cauchy loss
    bad = rng.choice(N, n_bad, replace=False)
    p_w[bad] += rng.normal(0, 3.0, size=(n_bad, 2))
    Tinit = Tt @ _se2(rng.normal(0, 0.15), rng.normal(0, 0.15), rng.normal(0, 0.01))
    pose = np.array([Tt[0, 2], Tt[1, 2], np.arctan2(Tt[1, 0], Tt[0, 0])])
    return T0, p_w, p_jt, Tinit, np.hstack((v, pose))
