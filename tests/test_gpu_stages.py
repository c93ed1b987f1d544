"""GPU parity tests (-m gpu): every stage of the HIP path, called through the C-ABI
(radarslampy_amd._ffi -> libroam_hip.so), against the oracle and the reference goldens.

Bars: bit-exact for index / byte / integer work (peaks, warp+quantise, pyramid, KLT,
adjacency, clique mask, SSC); 1e-4 m / 1e-5 rad for poses (Kabsch, motion-distortion LM)."""
import os

import numpy as np
import pytest

import oracle
from gen_inputs import synthetic_polar_u8

pytestmark = pytest.mark.gpu

POS_TOL = 1e-4
ANG_TOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    from radarslampy_amd import _ffi
    c = _ffi.Context(0)
    info = c.device_info()
    assert "gfx950" in info["arch"], info
    yield c
    c.close()


@pytest.fixture(scope="module")
def real_scans(golden):
    g = golden("peaks")
    return g["real0_u8"], g["real1_u8"]


# ------------------------------------------------------------------ a1/a2 peaks
def test_peaks_goldens(ctx, golden):
    g = golden("peaks")
    for i in (0, 1):
        u8 = g[f"real{i}_u8"]
        want = g[f"real{i}_out"]
        got = ctx.peaks_record_u8(u8, payload_off=0, clip=u8.shape[1])
        assert np.array_equal(got, want), f"real{i} u8"
        got = ctx.peaks_polar_f32(u8.astype(np.float32) / 255.)
        assert np.array_equal(got, want), f"real{i} f32"
    for seed in (0, 1, 2):
        u8 = synthetic_polar_u8(seed)
        got = ctx.peaks_record_u8(u8, payload_off=0, clip=u8.shape[1])
        assert np.array_equal(got, g[f"synth{seed}_out"]), f"synth{seed}"
    got = ctx.peaks_polar_f32(g["f32img"])
    assert np.array_equal(got, g["f32img_out"])


def test_peaks_full_record_layout(ctx, real_scans):
    # full Oxford record layout: 11 metadata bytes + 3768 power bins, clip 2025
    rng = np.random.default_rng(3)
    rec = rng.integers(0, 256, size=(400, 3779), dtype=np.uint8)
    rec[:, 11:11 + 2025] = real_scans[0]
    got = ctx.peaks_record_u8(rec, payload_off=11, clip=2025)
    want = oracle.peaks_from_record_u8(rec, 11, 2025)
    assert np.array_equal(got, want)
    # unclipped row (3768 bins)
    got = ctx.peaks_record_u8(rec, payload_off=11, clip=3768)
    want = oracle.peaks_from_record_u8(rec, 11, 3768)
    assert np.array_equal(got, want)


def test_peaks_ragged_shapes(ctx):
    rng = np.random.default_rng(4)
    for rows, cols in [(1, 3), (2, 2), (3, 1), (5, 7), (17, 129), (9, 1000), (4, 4096)]:
        img = (rng.random((rows, cols), dtype=np.float32) * 4).astype(np.float32)
        img = np.round(img * 8) / 8          # plateaus
        got = ctx.peaks_polar_f32(img)
        want = oracle.getPointCloudPolarInd(img)
        assert np.array_equal(got, want), (rows, cols)


# ------------------------------------------------------------------ a3 warp
def test_warp_bit_exact(ctx, real_scans):
    polar = real_scans[0].astype(np.float32) / 255.
    want_f, want_u = oracle.convertPolarImageToCartesian(polar, want_u8=True)
    got_f, got_u = ctx.polar_to_cart_f32(polar, want_f32=True, want_u8=True)
    assert got_f.shape == (2024, 2024)
    assert np.array_equal(got_f, want_f), f"{(got_f != want_f).sum()} px differ, max {np.abs(got_f - want_f).max()}"
    assert np.array_equal(got_u, want_u)
    # fused record path
    rec = np.zeros((400, 3779), np.uint8)
    rec[:, 11:11 + 2025] = real_scans[0]
    f2, u2 = ctx.polar_to_cart_record_u8(rec, 11, 2025, want_f32=True, want_u8=True)
    assert np.array_equal(f2, want_f) and np.array_equal(u2, want_u)


def test_warp_small_shapes(ctx):
    rng = np.random.default_rng(5)
    for rows, cols in [(8, 10), (16, 33), (37, 64), (400, 101)]:
        polar = rng.random((rows, cols), dtype=np.float32)
        want_f, want_u = oracle.convertPolarImageToCartesian(polar, want_u8=True)
        got_f, got_u = ctx.polar_to_cart_f32(polar, want_f32=True, want_u8=True)
        assert np.array_equal(got_f, want_f), (rows, cols)
        assert np.array_equal(got_u, want_u), (rows, cols)


# ------------------------------------------------------------------ pyramid + a7 KLT
@pytest.fixture(scope="module")
def cart_pair(real_scans):
    a = oracle.convertPolarImageToCartesian(real_scans[0].astype(np.float32) / 255., want_u8=True)
    b = oracle.convertPolarImageToCartesian(real_scans[1].astype(np.float32) / 255., want_u8=True)
    return a, b


def test_pyr_down(ctx, cart_pair):
    u8 = cart_pair[0][1]
    img = u8
    for _ in range(3):
        want = oracle.build_pyramid(img, 1)[1]
        got = ctx.pyr_down_u8(img)
        assert np.array_equal(got, want), img.shape
        img = want
    rng = np.random.default_rng(6)
    for h, w in [(16, 16), (17, 31), (65, 130), (129, 64), (20, 18), (33, 22), (9, 2046), (12, 2048), (70, 1030), (5, 20), (4, 16)]:
        im = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        assert np.array_equal(ctx.pyr_down_u8(im), oracle.build_pyramid(im, 1)[1]), (h, w)


def _feature_points(u8, n, seed):
    # bright local structures + border/out-of-range stress points
    rng = np.random.default_rng(seed)
    ys, xs = np.nonzero(u8 > 60)
    sel = rng.choice(len(ys), size=min(n, len(ys)), replace=False)
    pts = np.column_stack((xs[sel], ys[sel])).astype(np.float32)
    pts += rng.random(pts.shape, dtype=np.float32)
    extra = np.array([[0, 0], [2023, 2023], [3.5, 1000.25], [2020.75, 5.5], [1012, 1012], [-20, 50], [2100, 2100],
                      [7, 7], [2016.9, 2016.9]], np.float32)
    return np.vstack((pts, extra)).astype(np.float32)


def test_klt_bit_exact(ctx, cart_pair):
    (af, au), (bf, bu) = cart_pair
    pts = _feature_points(au, 300, 7)
    want_n, want_s, want_e = oracle.calcOpticalFlowPyrLK(au, bu, pts)
    got_n, got_s, got_e = ctx.klt_track(au, bu, pts)
    assert np.array_equal(got_s, want_s), f"{(got_s != want_s).sum()} status differ"
    assert np.array_equal(got_n, want_n), f"max diff {np.abs(got_n - want_n).max()}"
    assert np.array_equal(got_e, want_e)
    assert want_s.sum() > 100         # the test is not vacuous
    # f32 entry point quantises on the device
    got_n2, got_s2, got_e2 = ctx.klt_track(af, bf, pts)
    assert np.array_equal(got_n2, want_n) and np.array_equal(got_s2, want_s) and np.array_equal(got_e2, want_e)


def test_klt_recovers_known_shift(ctx):
    # known-answer: smooth random texture shifted by a sub-pixel translation
    rng = np.random.default_rng(8)
    H = W = 512
    base = rng.random((H // 8 + 4, W // 8 + 4))
    yy, xx = np.mgrid[0:H, 0:W]

    def sample(dx, dy):
        X = (xx + dx) / 8.0 + 1.5
        Y = (yy + dy) / 8.0 + 1.5
        x0, y0 = np.floor(X).astype(int), np.floor(Y).astype(int)
        fx, fy = X - x0, Y - y0
        v = (base[y0, x0] * (1 - fx) * (1 - fy) + base[y0, x0 + 1] * fx * (1 - fy) +
             base[y0 + 1, x0] * (1 - fx) * fy + base[y0 + 1, x0 + 1] * fx * fy)
        return np.clip(v * 255, 0, 255).astype(np.uint8)

    I0 = sample(0, 0)
    I1 = sample(-3.3, 1.7)            # content moves by (+3.3, -1.7)
    pts = rng.uniform(60, W - 60, size=(100, 2)).astype(np.float32)
    nxt, st, err = ctx.klt_track(I0, I1, pts)
    good = st.flatten() == 1
    assert good.sum() > 80
    d = (nxt - pts)[good]
    assert np.abs(np.median(d, axis=0) - np.array([3.3, -1.7])).max() < 0.1


# ------------------------------------------------------------------ a8 outlier rejection
def test_reject_outliers(ctx, golden):
    g = golden("outliers")
    thr = float(g["thr_px"])
    for tag in ["npz139", "npz139b", "real95", "u64", "u128", "u256", "u40"]:
        p, n = g[f"{tag}_prev"], g[f"{tag}_new"]
        mask, n_in, flags, adj = ctx.reject_outliers(p, n, thr, want_adj=True)
        assert flags & 1, tag
        assert np.array_equal(adj, oracle.consistency_graph(p, n, thr)), tag
        assert n_in == int(g[f"{tag}_size"]) == mask.sum(), tag
        _, omask, _ = oracle.max_clique_nx(adj)
        assert np.array_equal(mask, omask), tag
        # THE REFERENCE'S mask (tests/golden/make_goldens.py ran its rejectOutliers): its own three real fixtures have 16-,
        # 4- and several-way ties between maximum cliques; the first one in networkx.find_cliques order is returned
        assert np.array_equal(mask, g[f"{tag}_mask"]), tag


def test_reject_outliers_random_graphs(ctx):
    # correspondences whose consistency graph is irregular: compare with the oracle's set
    rng = np.random.default_rng(10)
    for K in [2, 3, 5, 33, 64, 65, 100, 200, 300]:
        p = rng.uniform(0, 2024, size=(K, 2)).astype(np.float32)
        n = p + rng.normal(0, 4.0, size=(K, 2)).astype(np.float32)
        mask, n_in, flags, adj = ctx.reject_outliers(p, n, oracle.DIST_THRESHOLD_PX, want_adj=True)
        assert np.array_equal(adj, oracle.consistency_graph(p, n))
        size, omask, _ = oracle.max_clique_nx(adj)
        assert flags & 1
        assert n_in == size and np.array_equal(mask, omask), K


def test_reject_outliers_tie_break_is_networkx_order(ctx):
    """tie-heavy consistency graphs from point sets (static points + movers + jitter that straddles the threshold), 20 to 320
    correspondences: the mask equals the oracle's networkx-order clique (itself pinned against the live networkx
    in tests/test_oracle_clique_order.py) and differs from the lexicographic rule of rounds 1-3 on many of them"""
    rng = np.random.default_rng(77)
    differ = 0
    for t in range(60):
        K = int(rng.integers(20, 320))
        p = rng.uniform(100, 1900, size=(K, 2)).astype(np.float32)
        th = rng.uniform(-0.02, 0.02)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        n = ((p - 1012) @ R.T + 1012 + rng.uniform(-20, 20, 2)).astype(np.float32)
        n += rng.normal(0, rng.choice([0.8, 1.6, 2.4]), size=(K, 2)).astype(np.float32)      # jitter near the 5.8 px threshold / 2
        movers = rng.permutation(K)[:int(K * rng.uniform(0.05, 0.4))]
        n[movers] += rng.normal(0, 12, size=(len(movers), 2)).astype(np.float32)
        mask, n_in, flags, adj = ctx.reject_outliers(p, n, oracle.DIST_THRESHOLD_PX, want_adj=True)
        size, omask, st = oracle.max_clique_nx(adj)
        assert flags & 1
        assert n_in == size and np.array_equal(mask, omask), (t, K)
        differ += not np.array_equal(omask, oracle.max_clique_lex(adj)[1])
    assert differ >= 15, differ


def test_reject_outliers_tie_break_on_large_tie_heavy_graphs(ctx):
    """the same construction at 600 to 1024 correspondences (round-4 advisor: the device walk had only been compared with the oracle
    up to K = 520; larger sets were checked on unique cliques only): sets above 512 keys use the 2048-slot table path of nx_walk"""
    rng = np.random.default_rng(2025)
    differ = ties = 0
    for K in (600, 700, 801, 903, 960):                       # (the oracle's walk takes a minute at 1024: unique cliques cover that size)
        p = rng.uniform(60, 1960, size=(K, 2)).astype(np.float32)
        th = rng.uniform(-0.02, 0.02)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        n = ((p - 1012) @ R.T + 1012 + rng.uniform(-20, 20, 2)).astype(np.float32)
        n += rng.normal(0, 1.2, size=(K, 2)).astype(np.float32)
        movers = rng.permutation(K)[:int(K * 0.25)]
        n[movers] += rng.normal(0, 12, size=(len(movers), 2)).astype(np.float32)
        mask, n_in, flags, adj = ctx.reject_outliers(p, n, oracle.DIST_THRESHOLD_PX, want_adj=True)
        size, omask, st = oracle.max_clique_nx(adj)
        assert flags & 1, K
        assert n_in == size and np.array_equal(mask, omask), K
        differ += not np.array_equal(omask, oracle.max_clique_lex(adj)[1])
    assert differ >= 2, differ


def test_reject_outliers_on_the_correspondence_sets_of_real_pairs(ctx, golden):
    """the sixteen correspondence sets the loop produces on the reference's ten real data/tiny pairs and on six bench-like pairs
    (tests/golden/clique_lone_sets.npz, made by profiles/clique_lone.py from the oracle's loop): 53-178 correspondences, 2- to 14-way
    ties between maximum cliques - the device's mask is the oracle's networkx-order clique on every one, proven"""
    g = golden("clique_lone_sets")
    tags = sorted(k[:-5] for k in g.files if k.endswith("_prev"))
    assert len(tags) == 16
    for tag in tags:
        p, n = g[tag + "_prev"], g[tag + "_new"]
        mask, n_in, flags, adj = ctx.reject_outliers(p, n, oracle.DIST_THRESHOLD_PX, want_adj=True)
        size, omask, _ = oracle.max_clique_nx(adj)
        assert flags & 1 and n_in == size and np.array_equal(mask, omask), tag


def test_reject_outliers_one_wavefront_variant_gives_the_same_masks(golden):
    """max_clique_kernel runs two wavefronts per problem (solver + walk on the first, the cand chain of the walk's descents and half of
    the solver's degree pass on the second); ROAM_CLIQUE_TWO_WAVES=0 (read at the first launch of a process) is the one-wavefront kernel
    kept for A/B runs.  Both give the oracle's networkx-order clique on the sixteen real / bench-like sets and the reference's fixtures."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "import oracle\n"
        "from radarslampy_amd import _ffi\n"
        "ctx = _ffi.Context(0)\n"
        "G = os.path.join(sys.path[0], 'tests', 'golden')\n"
        "g = np.load(os.path.join(G, 'clique_lone_sets.npz')); o = np.load(os.path.join(G, 'outliers.npz'))\n"
        "sets = [(g[k[:-5] + '_prev'], g[k[:-5] + '_new']) for k in sorted(g.files) if k.endswith('_prev')]\n"
        "sets += [(o[t + '_prev'], o[t + '_new']) for t in ('npz139', 'real95', 'u256')]\n"
        "bad = 0\n"
        "for p, n in sets:\n"
        "    mask, n_in, flags, adj = ctx.reject_outliers(p, n, oracle.DIST_THRESHOLD_PX, want_adj=True)\n"
        "    size, omask, _ = oracle.max_clique_nx(adj)\n"
        "    bad += not (flags & 1 and n_in == size and np.array_equal(mask, omask))\n"
        "print('CHECKED', len(sets), 'BAD', bad)\n")
    for two in ("0", "1"):
        env = dict(os.environ, ROAM_CLIQUE_TWO_WAVES=two)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert "CHECKED 19 BAD 0" in out.stdout, (two, out.stdout[-500:], out.stderr[-2000:])


# ------------------------------------------------------------------ a10 Kabsch
def test_kabsch(ctx, golden):
    g = golden("kabsch")
    for tag in ["real95_f64", "real95_f32", "clean100", "noisy100", "noisy250_f32", "n3", "n2"]:
        R, h = ctx.kabsch2d(g[f"{tag}_src"], g[f"{tag}_tgt"])
        Rw, hw = g[f"{tag}_R"], g[f"{tag}_h"]
        assert R.shape == (2, 2) and h.shape == (2, 1)
        assert np.abs(h - hw).max() * 0.0864 <= POS_TOL, tag
        assert abs(np.arctan2(R[1, 0], R[0, 0]) - np.arctan2(Rw[1, 0], Rw[0, 0])) <= ANG_TOL, tag
        assert abs(np.linalg.det(R) - 1) < 1e-12


# ------------------------------------------------------------------ a11-a14 motion distortion
def test_mds_goldens(ctx, golden):
    g = golden("mds")
    sigma5 = np.array([4, 4, 1, 1, (5 * np.pi / 180) ** 2], np.float64)
    for tag in ["n60", "n150", "n250", "n150big", "n8"]:
        sol, nfev, info, x0, r0 = ctx.mds_solve(g[f"{tag}_T0"], g[f"{tag}_p_w"], g[f"{tag}_p_jt"], g[f"{tag}_Tinit"],
                                                sigma5, 0.25, want_debug=True)
        assert np.allclose(x0, g[f"{tag}_x0"], rtol=1e-12, atol=1e-12), tag
        assert np.allclose(r0, g[f"{tag}_r0"], rtol=1e-9, atol=1e-11), (tag, np.abs(r0 - g[f"{tag}_r0"]).max())
        want = g[f"{tag}_sol"]
        assert 1 <= info <= 4, (tag, info)
        assert np.abs(sol[3:5] - want[3:5]).max() <= POS_TOL, (tag, sol, want)
        assert abs(sol[5] - want[5]) <= ANG_TOL, (tag, sol, want)
        assert np.abs(sol[:3] - want[:3]).max() <= 1e-3, (tag, sol, want)
        xy, dT = ctx.mds_undistort(g[f"{tag}_truth"][:3], g[f"{tag}_p_jt"])
        assert np.allclose(xy, g[f"{tag}_undist"][:, :2], rtol=0, atol=1e-12)
        assert np.allclose(dT, g[f"{tag}_dT"], rtol=0, atol=1e-15)


def test_mds_large_problem_global_workspace(ctx):
    # N large enough that the LM working set leaves LDS (global slab path)
    from gen_inputs import mds_problem
    T0, p_w, p_jt, Tinit, truth = mds_problem(700, 77, 0.05)
    sigma5 = np.array([4, 4, 1, 1, (5 * np.pi / 180) ** 2], np.float64)
    sol, nfev, info, _, _ = ctx.mds_solve(T0, p_w, p_jt, Tinit, sigma5)
    M = oracle.MotionDistortionSolver(np.diag([4, 4]), np.diag([1, 1, (5 * np.pi / 180) ** 2]))
    M.update_problem(T0, p_w, p_jt, Tinit)
    want = M.optimize_library()
    assert np.abs(sol[3:5] - want[3:5]).max() <= POS_TOL and abs(sol[5] - want[5]) <= ANG_TOL


def test_mds_both_kernel_forms_around_the_switch(ctx):
    """up to 254 points a problem is one wavefront with its working set in registers (csrc/lm_wave.inc), above that the workgroup form
    (mds_lm_kernel); both are launched when the bound allows both and each skips the other's problems - sizes on both sides of the switch
    and at the slot boundaries of the wave form (62 / 63, 126 / 127, 190 / 191 points + the two virtual ones)"""
    from gen_inputs import mds_problem
    sigma5 = np.array([4, 4, 1, 1, (5 * np.pi / 180) ** 2], np.float64)
    for N in (2, 3, 7, 61, 62, 63, 64, 126, 127, 190, 191, 253, 254, 255, 256, 300):
        T0, p_w, p_jt, Tinit, truth = mds_problem(N, 100 + N, 0.05)
        sol, nfev, info, x0, r0 = ctx.mds_solve(T0, p_w, p_jt, Tinit, sigma5, want_debug=True)
        M = oracle.MotionDistortionSolver(np.diag([4, 4]), np.diag([1, 1, (5 * np.pi / 180) ** 2]))
        M.update_problem(T0, p_w, p_jt, Tinit)
        want, x0w, r0w = M._solve()
        assert np.allclose(x0, x0w, rtol=1e-12, atol=1e-12), N
        assert np.allclose(r0, r0w, rtol=1e-9, atol=1e-11), N                        # the initial residual, every row in MINPACK's order
        assert 1 <= info <= 4, (N, info)
        assert np.abs(sol[3:5] - want[3:5]).max() <= POS_TOL and abs(sol[5] - want[5]) <= ANG_TOL, (N, sol, want)
        assert np.abs(sol[:3] - want[:3]).max() <= 1e-3, (N, sol, want)


# ------------------------------------------------------------------ a5 SSC
def test_ssc(ctx, golden):
    g = golden("ssc")
    for tag in ["b500", "b5000", "clus", "b230", "b190", "lattice"]:
        kp = g[f"{tag}_kp"]
        sel = ctx.ssc(kp, 200, 0.1, 2024, 2024)
        assert np.array_equal(kp[sel], g[f"{tag}_sel"]), tag
    rng = np.random.default_rng(5)
    kp = np.column_stack((rng.integers(0, 2024, (50, 2)).astype(float), np.full(50, 5.005)))
    sel = ctx.ssc(kp, 200, 0.1, 2024, 2024)          # pairwise (tiny-width) mode
    assert np.array_equal(kp[sel], oracle.ssc(kp, 200, 0.1, 2024, 2024))


# ------------------------------------------------------------------ a4 DoH + a5/a6 feature detection
def test_doh_maxima_bit_exact(ctx, cart_pair):
    cart = cart_pair[0][0]
    sig = np.linspace(0.01, 10, 3)
    want_rcs, want_val, _ = oracle.doh_maxima(cart, sig, 0.0005)
    got_rcs, got_val = ctx.doh_maxima(cart, sig, 0.0005)
    assert len(want_rcs) > 100
    assert np.array_equal(got_rcs, want_rcs)
    assert np.array_equal(got_val, want_val)
    rng = np.random.default_rng(12)
    for h, w in [(40, 50), (65, 64), (130, 97)]:
        img = rng.random((h, w), dtype=np.float32)
        for sg, thr in [([1.0, 2.0, 3.0], 0.001), ([0.01, 1.5], 0.0), ([2.0], 0.0005)]:
            a, b, _ = oracle.doh_maxima(img, sg, thr)
            c, d = ctx.doh_maxima(img, sg, thr)
            assert np.array_equal(a, c) and np.array_equal(b, d), (h, w, sg)


def test_get_features_matches_oracle(ctx, cart_pair):
    from radarslampy_amd.getFeatures import getFeatures, appendNewFeatures, getBlobsFromCart, DEFAULT_FEATURE_PARAMS
    cart = cart_pair[0][0]
    blobs = getBlobsFromCart(cart, **DEFAULT_FEATURE_PARAMS)
    want = oracle.blob_doh(cart.astype(np.float64), 0.01, 10, 3, 0.0005)
    assert np.array_equal(blobs, want)
    xy, rad = getFeatures(cart)
    wxy, wrad = oracle.getFeatures(cart)
    assert np.array_equal(xy, wxy) and np.array_equal(rad, wrad)
    assert 180 <= len(xy) <= 220
    old = xy[:50].astype(np.float32)
    pts, thr = appendNewFeatures(cart, old)
    assert thr == 80 and pts.dtype == np.float32
    assert np.array_equal(pts, oracle.append_dedupe(old, wxy))


# ------------------------------------------------------------------ f4 Fourier-Mellin rotation prior
def test_fmt_rotation_matches_oracle(ctx):
    """roam_fmt_rotation (resize, two warpPolar steps, windowed phase correlation by direct DFTs) vs the oracle's numpy
    restatement of FMT.getRotationUsingFMT: 1e-5 rad; and through the reference-named wrappers (FMT, Tracker.track slot 3)"""
    from radarslampy_amd import synth
    from radarslampy_amd.FMT import getRotationUsingFMT
    recs, poses, feat = synth.make_sequence(3, 2, n_movers=6)
    p0 = recs[0][:, 11:11 + 2025].astype(np.float32) / np.float32(255.)
    p1 = recs[1][:, 11:11 + 2025].astype(np.float32) / np.float32(255.)
    for a, b in ((p0, p0), (p0, p1), (p0, np.roll(p0, 7, axis=0)), (p1, np.roll(p0, -31, axis=0))):
        got = ctx.fmt_rotation(a, b)
        want = oracle.getRotationUsingFMT(a, b)
        assert abs(got[0] - want[0]) <= 1e-5, (got, want)
        assert abs(got[1] - want[1]) <= 1e-5 and abs(got[2] - want[2]) <= 1e-4 * max(1.0, abs(want[2])), (got, want)
    assert getRotationUsingFMT(p0, p1)[0] == ctx.fmt_rotation(p0, p1)[0]
    assert abs(ctx.fmt_rotation(p0, np.roll(p0, 7, axis=0))[0] + 7 * 2 * np.pi / 400) < 5e-3
