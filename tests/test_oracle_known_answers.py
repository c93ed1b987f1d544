"""CPU (-m "not gpu"): known-answer tests for the three PARITY-UNPINNED restatements
(cv2.warpPolar, cv2.calcOpticalFlowPyrLK, skimage blob_doh): since no reference output can be
captured for them here, their correctness is pinned by analytic cases."""
import numpy as np

import oracle


def test_warp_geometry_known_answer():
    # a polar image that encodes its own coordinates: value = range_bin / cols on every row ->
    # the Cartesian image must be the radial distance; and an angular ramp must come back as atan2
    rows, cols = 400, 512
    polar = np.tile(np.arange(cols, dtype=np.float32) / cols, (rows, 1))
    cart = oracle.convertPolarImageToCartesian(polar)
    R = cols // 2
    yy, xx = np.mgrid[0:2 * R, 0:2 * R]
    rho = np.hypot(xx - R, yy - R)
    inside = rho < R - 2
    want = rho * (cols / R) / cols
    assert np.abs(cart[inside] - want[inside]).max() < 2.5 / cols
    assert cart[0, 0] == 0.0                                  # beyond max radius: filled with zeros
    ang = np.tile((np.arange(rows, dtype=np.float32) / rows)[:, None], (1, cols))
    cart = oracle.convertPolarImageToCartesian(ang)
    phi = (np.arctan2(yy - R, xx - R) % (2 * np.pi)) / (2 * np.pi)
    sel = inside & (rho > 20) & (phi > 0.01) & (phi < 0.99)
    assert np.abs(cart[sel] - phi[sel]).max() < 1.5 / rows


def _texture(shift, H=320, W=320, seed=3):
    rng = np.random.default_rng(seed)
    base = rng.random((H // 8 + 6, W // 8 + 6))
    yy, xx = np.mgrid[0:H, 0:W]
    X = (xx + shift[0]) / 8.0 + 2.5
    Y = (yy + shift[1]) / 8.0 + 2.5
    x0, y0 = np.floor(X).astype(int), np.floor(Y).astype(int)
    fx, fy = X - x0, Y - y0
    v = (base[y0, x0] * (1 - fx) * (1 - fy) + base[y0, x0 + 1] * fx * (1 - fy) + base[y0 + 1, x0] * (1 - fx) * fy +
         base[y0 + 1, x0 + 1] * fx * fy)
    return np.clip(v * 255, 0, 255).astype(np.uint8)


def test_klt_recovers_subpixel_and_multilevel_shifts():
    rng = np.random.default_rng(4)
    pts = rng.uniform(50, 270, size=(80, 2)).astype(np.float32)
    for true in [(0.4, -0.3), (-3.3, 1.7), (9.5, -6.25)]:
        I0, I1 = _texture((0, 0)), _texture((-true[0], -true[1]))
        nxt, st, err = oracle.calcOpticalFlowPyrLK(I0, I1, pts)
        good = st.flatten() == 1
        assert good.sum() >= 70
        d = (nxt - pts)[good]
        assert np.abs(np.median(d, axis=0) - np.array(true)).max() < 0.06, true
        assert (err[good] < 10).mean() > 0.9
    # identical images: zero flow, zero error
    nxt, st, err = oracle.calcOpticalFlowPyrLK(I0, I0, pts)
    assert np.abs(nxt - pts).max() < 1e-3 and err.max() == 0.0
    # flat image: min-eigenvalue test rejects every point
    flat = np.full((128, 128), 77, np.uint8)
    nxt, st, err = oracle.calcOpticalFlowPyrLK(flat, flat, np.array([[60, 60]], np.float32))
    assert st[0, 0] == 0


def test_pyr_down_preserves_constants_and_halves_size():
    img = np.full((101, 64), 200, np.uint8)
    d = oracle.build_pyramid(img, 3)
    assert [a.shape for a in d] == [(101, 64), (51, 32), (26, 16), (13, 8)]
    assert all((a == 200).all() for a in d)


def test_doh_finds_gaussian_blobs_at_the_right_place():
    H = W = 300
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.zeros((H, W))
    centres = [(60, 70, 4.0), (150, 200, 7.0), (230, 90, 5.0), (90, 240, 8.0)]
    for r, c, s in centres:
        img += 0.8 * np.exp(-((yy - r) ** 2 + (xx - c) ** 2) / (2 * s * s))
    blobs = oracle.blob_doh(img, min_sigma=0.01, max_sigma=10, num_sigma=3, threshold=0.0005)
    assert len(blobs) >= len(centres)
    for r, c, s in centres:
        dist = np.hypot(blobs[:, 0] - r, blobs[:, 1] - c)
        assert dist.min() <= 2.0, (r, c, dist.min())
    assert set(np.unique(blobs[:, 2])) <= {5.005, 10.0}      # the sigma=0.01 layer is degenerate
    # a blank image has no blobs
    assert oracle.blob_doh(np.zeros((64, 64)), 0.01, 10, 3, 0.0005).shape == (0, 3)


def test_doh_hessian_det_matches_direct_transcription():
    rng = np.random.default_rng(6)
    img = rng.random((37, 45)).astype(np.float32)
    S = img.astype(np.float64).cumsum(0).cumsum(1)
    H, W = S.shape

    def integ(r, c, rl, cl):
        r = min(max(r, 0), H - 1); c = min(max(c, 0), W - 1)
        r2 = min(max(r + rl, 0), H - 1); c2 = min(max(c + cl, 0), W - 1)
        return max(S[r, c] + S[r2, c2] - S[r, c2] - S[r2, c], 0)

    for sigma in (1.0, 2.5, 5.005):
        size = int(3 * sigma); s2 = (size - 1) // 2; s3 = size // 3; w = size; wi = 1.0 / size / size
        want = np.zeros((H, W))
        for r in range(H):
            for c in range(W):
                dxy = -(integ(r - s3, c + 1, s3, s3) + integ(r + 1, c - s3, s3, s3) - integ(r - s3, c - s3, s3, s3) - integ(r + 1, c + 1, s3, s3)) * wi
                dxx = -(integ(r - s3 + 1, c - s2, 2 * s3 - 1, w) - 3 * integ(r - s3 + 1, c - s3 // 2, 2 * s3 - 1, s3)) * wi
                dyy = -(integ(r - s2, c - s3 + 1, w, 2 * s3 - 1) - 3 * integ(r - s3 // 2, c - s3 + 1, s3, 2 * s3 - 1)) * wi
                want[r, c] = dxx * dyy - 0.81 * dxy * dxy
        _, _, layers = oracle.doh_maxima(img, [sigma], 0.0)
        assert np.abs(layers[0] - want).max() < 1e-15


def test_oracle_pipeline_tracks_a_synthetic_sequence():
    """BASELINE configs[0] analogue (CPU plumbing, no GPU): the oracle's restatement of the
    RawROAMSystem.run loop body on a short synthetic sequence stays close to the rendered ground
    truth, both with the motion-distortion LM and with Kabsch dead reckoning, and is deterministic."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from radarslampy_amd import synth          # pure-numpy generator (no GPU needed)
    recs, poses, feat = synth.make_sequence(31, 4)
    for md in (True, False):
        P = oracle.OdometryPipeline(recs[0], feat, poses[0], motion_distortion=md)
        Q = oracle.OdometryPipeline(recs[0], feat, poses[0], motion_distortion=md)
        for t in range(1, 4):
            a, b = P.step(recs[t]), Q.step(recs[t])
            assert np.array_equal(a["pose"], b["pose"])
            assert a["n_inliers"] > 150
            if md:
                assert np.abs(a["pose"][:2] - poses[t][:2]).max() < 0.35, (md, t, a["pose"], poses[t])
            # dead reckoning (RawROAMSystem.py:236,301-317) inherits the reference's quirk: h comes from
            # UNCENTRED pixel coordinates (:190), so its translation is off by (R - I) * centre under
            # rotation (SURVEY 8a quirk i); only the heading is meaningful there
            assert abs(a["pose"][2] - poses[t][2]) < 0.01
            assert len(a["peaks"]) > 3000


def test_fmt_rotation_known_answer():
    """f4 (FMT.py:36-90): a scan rotated by k azimuth rows comes back as -k * 2 pi / 400 within the method's resolution
    (one log-polar row = 2 pi / 317 rad, refined by the 5 x 5 centroid); identical scans give exactly 0 and scale 1"""
    from radarslampy_amd import synth
    recs, _, _ = synth.make_sequence(3, 1)
    polar = recs[0][:, 11:11 + 2025].astype(np.float32) / np.float32(255.)
    a, sc, resp = oracle.getRotationUsingFMT(polar, polar)
    assert abs(a) < 1e-9 and abs(sc - 1) < 1e-9 and resp > 0.9
    for k in (3, -5, 20):
        a, sc, resp = oracle.getRotationUsingFMT(polar, np.roll(polar, k, axis=0))
        assert abs(a + k * 2 * np.pi / 400) < 5e-3, (k, a)          # a quarter of a log-polar row
        assert abs(sc - 1) < 5e-3 and resp > 0.3
