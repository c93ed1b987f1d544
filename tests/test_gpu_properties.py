"""GPU (-m gpu): size-independent properties at BASELINE's full sizes and at the build's limits."""
import numpy as np
import pytest

import oracle
from gen_inputs import unique_clique_pairs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from radarslampy_amd import _ffi
    c = _ffi.Context(0)
    yield c
    c.close()


def test_peaks_properties_full_width_record(ctx):
    """400 x 3768 bins (unclipped Oxford row): every reported bin is a strict local maximum / plateau
    midpoint, rows and bins ascend, and every kept height clears its row's mean + std."""
    from radarslampy_amd import synth
    rec = synth.make_sequence(11, 1)[0][0]
    pts = ctx.peaks_record_u8(rec, payload_off=11, clip=3768)
    assert len(pts) > 1000
    key = pts[:, 0].astype(np.int64) * 4096 + pts[:, 1]
    assert np.all(np.diff(key) > 0)                                      # azimuth-major, range ascending
    row = rec[:, 11:].astype(np.int32)
    a, r = pts[:, 0], pts[:, 1]
    assert np.all((r >= 1) & (r <= 3766))
    v = row[a, r]
    assert np.all(row[a, r - 1] <= v) and np.all(row[a, r + 1] <= v)       # plateau midpoints allow equality
    assert np.array_equal(pts, oracle.peaks_from_record_u8(rec, 11, 3768))


def test_clique_properties_at_capacity(ctx):
    """K = 1024 (the build's K_max, adjacency read through L2) and K = 600: the mask is a clique, it
    is maximal (no outside vertex is adjacent to all members), and equals the planted inlier set."""
    for K, seed in [(600, 31), (1024, 32)]:
        p, n, inl = unique_clique_pairs(K, seed, 0.3)
        mask, n_in, flags, adj = ctx.reject_outliers(p, n, oracle.DIST_THRESHOLD_PX, want_adj=True)
        assert flags & 1
        A = oracle.adjacency_dense(adj, K)
        idx = np.flatnonzero(mask)
        assert A[np.ix_(idx, idx)][~np.eye(len(idx), dtype=bool)].all()
        outside = np.flatnonzero(~mask)
        assert not A[np.ix_(outside, idx)].all(axis=1).any()
        assert n_in == inl.sum() and np.array_equal(mask, inl)


def test_klt_many_points_and_determinism(ctx, golden):
    g = golden("peaks")
    a = oracle.convertPolarImageToCartesian(g["real0_u8"].astype(np.float32) / 255., want_u8=True)[1]
    b = oracle.convertPolarImageToCartesian(g["real1_u8"].astype(np.float32) / 255., want_u8=True)[1]
    rng = np.random.default_rng(3)
    pts = rng.uniform(-30, 2054, size=(1024, 2)).astype(np.float32)          # includes points outside the image
    n1, s1, e1 = ctx.klt_track(a, b, pts)
    n2, s2, e2 = ctx.klt_track(a, b, pts)
    assert np.array_equal(n1, n2) and np.array_equal(s1, s2) and np.array_equal(e1, e2)
    wn, ws, we = oracle.calcOpticalFlowPyrLK(a, b, pts)
    assert np.array_equal(n1, wn) and np.array_equal(s1, ws) and np.array_equal(e1, we)
    # tracking an image onto itself leaves every trackable point where it is
    n0, s0, e0 = ctx.klt_track(a, a, pts)
    good = s0.flatten() == 1
    assert np.abs(n0[good] - pts[good]).max() < 1e-3 and e0[good].max() == 0


def test_lm_large_small_problems_and_determinism(ctx):
    from gen_inputs import mds_problem
    sigma5 = np.array([4, 4, 1, 1, (5 * np.pi / 180) ** 2])
    for N, seed in [(12, 1), (333, 2), (1024, 3)]:
        T0, p_w, p_jt, Ti, truth = mds_problem(N, seed, 0.08)
        s1, nf1, info1, x0, r0 = ctx.mds_solve(T0, p_w, p_jt, Ti, sigma5, want_debug=True)
        s2, nf2, info2, _, _ = ctx.mds_solve(T0, p_w, p_jt, Ti, sigma5)
        assert np.array_equal(s1, s2) and nf1 == nf2
        assert np.isfinite(s1).all() and 1 <= info1 <= 4
        want = oracle.MotionDistortionSolver(np.diag([4, 4]), np.diag([1, 1, (5 * np.pi / 180) ** 2]))
        want.update_problem(T0, p_w, p_jt, Ti)
        ws = want.optimize_library()
        assert np.abs(s1[3:5] - ws[3:5]).max() <= 1e-4 and abs(s1[5] - ws[5]) <= 1e-5, N


def test_engine_lanes_are_independent_and_deterministic():
    """Identical lanes give identical results; a lane's result does not depend on its neighbours."""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, feat = synth.make_sequence(21, 3)
    recs2, poses2, feat2 = synth.make_sequence(22, 3)
    ctx = _ffi.Context(0)
    eng = Engine(3, 6, ctx=ctx)
    for t in range(3):
        eng.upload_scan(t, recs[t])
        eng.upload_scan(3 + t, recs2[t])
    eng.init_lane(0, 0, feat, poses[0])
    eng.init_lane(1, 3, feat2, poses2[0])
    eng.init_lane(2, 0, feat, poses[0])
    out = []
    for t in (1, 2):
        eng.step([t, 3 + t, t])
        out.append(eng.results())
    for r in out:
        assert np.array_equal(r[0]["pose"], r[2]["pose"]) and r[0]["n_inliers"] == r[2]["n_inliers"]
    assert np.array_equal(eng.lane_features(0), eng.lane_features(2))
    solo = Engine(1, 3, ctx=_ffi.Context(0))
    for t in range(3):
        solo.upload_scan(t, recs[t])
    solo.init_lane(0, 0, feat, poses[0])
    for i, t in enumerate((1, 2)):
        solo.step([t])
        assert np.array_equal(solo.results()[0]["pose"], out[i][0]["pose"])
    eng.close()
    ctx.close()


def test_peaks_u8_ragged_clips(ctx):
    """register-based u8 row kernel (clip <= 2048) and the generic kernel (clip > 2048) across awkward widths,
    payload offsets and row counts, with plateaus, saturated runs and empty rows"""
    rng = np.random.default_rng(17)
    for rows, clip, off in [(1, 3, 0), (2, 8, 1), (3, 9, 11), (5, 63, 2), (4, 64, 11), (4, 65, 0), (7, 1000, 11), (3, 2047, 5),
                            (3, 2048, 11), (2, 2049, 11), (2, 3768, 11)]:
        stride = off + clip + int(rng.integers(0, 7))
        rec = rng.integers(0, 6, size=(rows, stride), dtype=np.uint8) * 40       # few levels -> many plateaus
        rec[0, off:off + clip] = 255                                              # saturated row
        if rows > 1:
            rec[1, off:off + clip] = 0                                            # empty row
        got = ctx.peaks_record_u8(rec, payload_off=off, clip=clip)
        want = oracle.peaks_from_record_u8(rec, off, clip)
        assert np.array_equal(got, want), (rows, clip, off)


def test_peaks_u8_long_plateaus(ctx):
    """flat-topped peaks of every length 1..200 at offsets that straddle the 32-bin lane and 64-bin
    lookahead boundaries of the wave-per-row kernel, plateaus touching both ends of the row, staircases"""
    rng = np.random.default_rng(23)
    off, clip, stride = 11, 2025, 3779
    rows = []
    for length in list(range(1, 70)) + [95, 96, 97, 127, 128, 129, 200, 500, 1500, 2023]:
        for start in (1, 2, 31, 32, 33, 63, 64, 65, 700, clip - length - 1, clip - length):
            if start < 0 or start + length > clip:
                continue
            row = np.full(clip, 10, np.uint8)
            row[start:start + length] = 200                       # plateau (a peak iff it does not touch either end)
            rows.append(row)
    stair = np.repeat(np.arange(1, 64, dtype=np.uint8), 33)[:clip]             # rising staircase then a drop
    stair = np.concatenate([stair, np.zeros(clip - stair.size, np.uint8)])
    rows.append(stair)
    rows.append(stair[::-1].copy())
    rows.append(np.zeros(clip, np.uint8))
    noisy = rng.integers(0, 3, size=(40, clip), dtype=np.uint8) * 100                         # long equal runs of 3 levels
    rows.extend(list(noisy))
    rec = np.zeros((len(rows), stride), np.uint8)
    rec[:, off:off + clip] = np.stack(rows)
    rec[:, off + clip:] = 255                                                                # bytes after the payload must be ignored
    got = ctx.peaks_record_u8(rec, payload_off=off, clip=clip)
    want = oracle.peaks_from_record_u8(rec, off, clip)
    assert want.shape[0] > 300
    assert np.array_equal(got, want)


def test_clique_tie_break_in_every_table_regime(ctx):
    """the networkx-order walk where its set tables take every form: K up to 520 (tables of 512 slots with keys beyond them: the
    fixed-point replay in LDS), sparse graphs (small neighbour sets iterate in hashed, not ascending, order; `cand - adj[u]` takes
    CPython's copy-and-discard branch), dense tie-heavy graphs (the register replay, the bulk prefix) - the mask equals the
    oracle's networkx-order clique on every one (the oracle itself is pinned against the live networkx on CPU)"""
    rng = np.random.default_rng(2024)
    cases = []
    for K in (9, 33, 70, 129, 200, 300, 520):
        # scan-pair-like: a rigid set + movers + jitter straddling the threshold
        p = rng.uniform(60, 1960, size=(K, 2)).astype(np.float32)
        n = (p + rng.normal(0, 1.9, size=(K, 2))).astype(np.float32)
        mv = rng.permutation(K)[:max(1, K // 4)]
        n[mv] += rng.normal(0, 15, size=(len(mv), 2)).astype(np.float32)
        cases.append((p, n))
    for K in (40, 150, 400):
        # sparse: almost everything moves on its own, a few small rigid groups
        p = rng.uniform(60, 1960, size=(K, 2)).astype(np.float32)
        n = (p + rng.normal(0, 40, size=(K, 2))).astype(np.float32)
        for g in range(K // 12):
            idx = rng.permutation(K)[:6]
            n[idx] = p[idx] + rng.normal(0, 25, size=2).astype(np.float32) + rng.normal(0, 0.8, size=(6, 2)).astype(np.float32)
        cases.append((p, n))
    differ = 0
    for p, n in cases:
        mask, n_in, flags, adj = ctx.reject_outliers(p, n, oracle.DIST_THRESHOLD_PX, want_adj=True)
        size, omask, _ = oracle.max_clique_nx(adj)
        assert flags & 1
        assert n_in == size and np.array_equal(mask, omask), len(p)
        differ += not np.array_equal(omask, oracle.max_clique_lex(adj)[1])
    assert differ >= 4, differ
