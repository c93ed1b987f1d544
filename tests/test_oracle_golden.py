"""CPU (-m "not gpu"): the oracle against the golden vectors captured from the reference."""
import hashlib

import numpy as np
import pytest

import oracle
from gen_inputs import synthetic_polar_u8

POS_TOL = 1e-4   # metres  (north_star tolerance)
ANG_TOL = 1e-5   # radians


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_pairwise_sum_matches_numpy():
    rng = np.random.default_rng(0)
    for n in [0, 1, 5, 7, 8, 9, 15, 16, 17, 100, 127, 128, 129, 255, 256, 257, 300, 777, 1012, 2025]:
        a = rng.random(n, dtype=np.float32) * 3
        assert oracle.pairwise_sum_f32(a) == np.add.reduce(a), n


def test_peaks_real_and_synthetic(golden):
    g = golden("peaks")
    for i in (0, 1):
        u8 = g[f"real{i}_u8"]
        want = g[f"real{i}_out"]
        got = oracle.getPointCloudPolarInd(u8.astype(np.float32) / 255.)
        assert np.array_equal(got, want)
        got8 = oracle.peaks_from_record_u8(u8, payload_off=0, clip=u8.shape[1])
        assert np.array_equal(got8, want)
    for seed in (0, 1, 2):
        u8 = synthetic_polar_u8(seed)
        assert _sha(u8) == str(g[f"synth{seed}_sha"])
        got = oracle.peaks_from_record_u8(u8, payload_off=0, clip=u8.shape[1])
        assert np.array_equal(got, g[f"synth{seed}_out"])
    got = oracle.getPointCloudPolarInd(g["f32img"])
    assert np.array_equal(got, g["f32img_out"])


def test_record_format(golden):
    g = golden("record_format")
    rec = np.zeros((400, 3779), np.uint8)
    rec[:, :11] = g["meta_u8"]
    rec[:, 11:11 + 64] = g["payload_head_u8"]
    data, az, rres, ares, valid, ts = oracle.extractDataFromRadarImage(rec)
    assert data.shape == (400, 2025) and data.dtype == np.float32
    assert np.array_equal(data[:, :64], g["polar_head"])
    assert np.array_equal(az, g["azimuths"]) and np.array_equal(valid, g["valid"])
    assert np.array_equal(ts, g["timestamps"])


def test_ssc(golden):
    g = golden("ssc")
    for tag in ["b500", "b5000", "clus", "b230", "b190", "lattice"]:
        sel = oracle.ssc(g[f"{tag}_kp"], 200, 0.1, 2024, 2024)
        assert np.array_equal(sel, g[f"{tag}_sel"]), tag


def test_ssc_small_sets_terminate():
    # B < 180: the reference runs out of memory; the restatement must still terminate and
    # return every distinct keypoint (no keypoint can cover another at sub-pixel widths).
    rng = np.random.default_rng(5)
    kp = np.column_stack((rng.integers(0, 2024, (50, 2)).astype(float), np.full(50, 5.005)))
    sel = oracle.ssc(kp, 200, 0.1, 2024, 2024)
    assert sel.shape[0] == len(np.unique(kp[:, :2], axis=0))


def _pose_err(R, h, R2, h2):
    dth = abs(np.arctan2(R[1, 0], R[0, 0]) - np.arctan2(R2[1, 0], R2[0, 0]))
    return np.abs(np.asarray(h).ravel() - np.asarray(h2).ravel()).max(), dth


def test_kabsch(golden):
    g = golden("kabsch")
    for tag in ["real95_f64", "real95_f32", "clean100", "noisy100", "noisy250_f32", "n3", "n2"]:
        s, t = g[f"{tag}_src"], g[f"{tag}_tgt"]
        for fn in (oracle.calculateTransformSVD, oracle.kabsch_closed_form):
            R, h = fn(s, t)
            dpx, dth = _pose_err(R, h, g[f"{tag}_R"], g[f"{tag}_h"])
            # h is in pixels here; 1e-4 m = 1.16e-3 px
            assert dpx * 0.0864 <= POS_TOL and dth <= ANG_TOL, (tag, fn.__name__, dpx, dth)
            assert h.shape == (2, 1) and R.shape == (2, 2)


def test_outlier_rejection(golden):
    g = golden("outliers")
    assert abs(float(g["thr_px"]) - oracle.DIST_THRESHOLD_PX) < 1e-15
    for tag in ["npz139", "npz139b", "real95", "u64", "u128", "u256", "u40"]:
        p, n = g[f"{tag}_prev"], g[f"{tag}_new"]
        pp, nn, mask = oracle.rejectOutliers(p, n)
        assert mask.sum() == int(g[f"{tag}_size"]), tag
        A = oracle.adjacency_dense(oracle.consistency_graph(p, n), len(p))
        idx = np.flatnonzero(mask)
        assert A[np.ix_(idx, idx)][~np.eye(len(idx), dtype=bool)].all(), "not a clique"
        assert np.array_equal(pp, p[mask]) and np.array_equal(nn, n[mask])
        assert np.array_equal(mask, g[f"{tag}_mask"]), tag       # the reference's own mask, ties included (networkx order)


def test_adjacency_matches_numpy_cdist(golden):
    g = golden("outliers")
    p, n = g["npz139_prev"].astype(np.float64), g["npz139_new"].astype(np.float64)
    d0 = np.sqrt(((p[:, None] - p[None]) ** 2).sum(-1))
    d1 = np.sqrt(((n[:, None] - n[None]) ** 2).sum(-1))
    want = np.abs(d0 - d1) <= oracle.DIST_THRESHOLD_PX
    np.fill_diagonal(want, False)
    A = oracle.adjacency_dense(oracle.consistency_graph(g["npz139_prev"], g["npz139_new"]), len(p))
    assert np.array_equal(A, want)


def test_clique_lex_rule_against_bruteforce():
    rng = np.random.default_rng(9)
    for K, dens in [(12, 0.5), (20, 0.7), (28, 0.8), (35, 0.6), (30, 0.9)]:
        for rep in range(6):
            A = rng.random((K, K)) < dens
            A = np.triu(A, 1)
            A = A | A.T
            nw = (K + 63) // 64
            adj = np.zeros((K, nw), np.uint64)
            for i in range(K):
                for j in np.flatnonzero(A[i]):
                    adj[i, j >> 6] |= np.uint64(1) << np.uint64(j & 63)
            size, mask, _ = oracle.max_clique_lex(adj)
            bsize, bmask = oracle.max_clique_bruteforce(A)
            assert size == bsize and np.array_equal(mask, bmask)


def test_mds(golden):
    g = golden("mds")
    cov_p = np.diag([4, 4])
    cov_v = np.diag([1, 1, (5 * np.pi / 180) ** 2])
    for tag in ["n60", "n150", "n250", "n150big", "n8"]:
        M = oracle.MotionDistortionSolver(cov_p, cov_v)
        M.update_problem(g[f"{tag}_T0"], g[f"{tag}_p_w"], g[f"{tag}_p_jt"], g[f"{tag}_Tinit"])
        assert np.allclose(M.dT, g[f"{tag}_dT"], rtol=0, atol=1e-15)
        sol, x0, r0 = M._solve()
        assert np.allclose(x0, g[f"{tag}_x0"], rtol=1e-12, atol=1e-12)
        assert np.allclose(r0, g[f"{tag}_r0"], rtol=1e-9, atol=1e-11), np.abs(r0 - g[f"{tag}_r0"]).max()
        want = g[f"{tag}_sol"]
        assert np.abs(sol[3:5] - want[3:5]).max() <= POS_TOL, (tag, sol, want)
        assert abs(sol[5] - want[5]) <= ANG_TOL, (tag, sol, want)
        assert np.abs(sol[:3] - want[:3]).max() <= 1e-3, (tag, sol, want)
        und = oracle.MotionDistortionSolver.undistort(g[f"{tag}_truth"][:3], g[f"{tag}_p_jt"])
        assert np.allclose(und, g[f"{tag}_undist"], rtol=0, atol=1e-12)


def test_se2_utils(golden):
    g = golden("se2_utils")
    for p, T in zip(g["poses"], g["T"]):
        assert np.allclose(oracle.convertPoseToTransform(p), T, atol=1e-15)
    for T, p in zip(g["T"], g["poses_back"]):
        assert np.allclose(oracle.convertTransformToPose(T), p, atol=1e-15)
    assert np.allclose(oracle.normalize_angles(g["ang"]), g["ang_norm"], atol=1e-15)


def test_tracker_glue(golden):
    g = golden("tracker_glue")
    st = g["klt_status"]
    good = st.flatten().astype(bool)
    p, n = g["prev"], g["new"]
    klt_out = (n[good], p[good], n[~good], p[~good], st.copy())
    g_old, g_new, _, cs = oracle.track_glue(klt_out)
    assert np.array_equal(g_old, g["good_old"]) and np.array_equal(g_new, g["good_new"])
    assert np.array_equal(cs, g["corrStatus"])
    R, h = oracle.calculateTransformSVD(g_old, g_new)
    dpx, dth = _pose_err(R, h * 0.0864, g["R"], g["h"])
    assert dpx <= POS_TOL and dth <= ANG_TOL


def test_append_dedupe():
    old = np.array([[5., 6.], [1., 2.], [5., 6.]], np.float32)
    new = np.array([[1., 2.], [9., 9.], [0., 0.]])
    out = oracle.append_dedupe(old, new)
    assert out.dtype == np.float32
    assert np.array_equal(out, np.array([[5, 6], [1, 2], [9, 9], [0, 0]], np.float32))
