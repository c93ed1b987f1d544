"""GPU (-m gpu): feature (re)detection INSIDE roam_engine_step (retrack.hip) - appendNewFeatures of the reference's loop
(RawROAMSystem.py:250-271, getFeatures.py:74-118) without a host round trip - against the oracle's loop body, on synthetic
sequences and on the reference's 11 real data/tiny scans."""
import numpy as np
import pytest

import oracle
from test_oracle_reference_dump import FIX, W

pytestmark = pytest.mark.gpu

POS_TOL = 1e-4   # m
ANG_TOL = 1e-5   # rad


def _detect(cart):
    return oracle.getFeatures(cart)[0]


def _same_pose(got, want, tag):
    assert np.abs(got["pose"][:2] - want["pose"][:2]).max() <= POS_TOL and abs(got["pose"][2] - want["pose"][2]) <= ANG_TOL, (tag, got["pose"], want["pose"])


def test_init_lane_detect_matches_get_features():
    """first-frame appendNewFeatures(prevImgCart, empty) on the device == oracle getFeatures (DoH, skimage-order pruning,
    numpy-1.22 sigma order, SSC), as float32 [x, y] in ANMS order"""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, _ = synth.make_sequence(21, 2, n_movers=8)
    ctx = _ffi.Context(0)
    eng = Engine(2, 2, ctx=ctx, retrack_on_device=True)
    for t in range(2):
        eng.upload_scan(t, recs[t])
    for b in range(2):
        eng.init_lane_detect(b, b, poses[0])
        cart = oracle.convertPolarImageToCartesian(recs[b][:, 11:11 + 2025].astype(np.float32) / np.float32(255.))
        want = oracle.append_dedupe(np.empty((0, 2)), _detect(cart))
        got = eng.lane_features(b)
        assert 180 <= len(want) <= 220
        assert np.array_equal(got, want), (b, len(got), len(want))
        kf = eng.map_keyframe(b, 0)
        assert kf["scan"] == b and np.allclose(kf["pose"], poses[0])
        okf = oracle.Keyframe(poses[0], (want - oracle.RADAR_CART_CENTER) * oracle.RANGE_RESOLUTION_CART_M, None, np.zeros(3), with_peaks=False)
        assert np.abs(kf["prunedUndistortedLocals"] - okf.prunedUndistortedLocals).max() <= 1e-9
    eng.close()
    ctx.close()


def test_engine_device_retrack_matches_oracle():
    """lanes starting with few features run into the retrack branch at different steps; lanes with plenty never do.  No host
    involvement between steps; features, keyframes and poses follow the oracle's loop body step by step."""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, feat = synth.make_sequence(1, 6, n_movers=6, distortion=True)
    T = len(recs)
    starts = [feat[:64], feat[:20], feat, feat[:0], feat[100:161]]
    B = len(starts)
    ctx = _ffi.Context(0)
    eng = Engine(B, T, ctx=ctx, retrack_on_device=True, retrack_slots=2)      # 5 lanes through 2 scratch slots: chunking
    eng.map_reserve(8)
    for t in range(T):
        eng.upload_scan(t, recs[t])
    pipes = []
    for b, f in enumerate(starts):
        eng.init_lane(b, 0, f, poses[0])
        pipes.append(oracle.OdometryPipeline(recs[0], f, poses[0], detect=_detect))
    n_rt = 0
    for t in range(1, T):
        eng.step([t] * B)
        res = eng.results()
        for b in range(B):
            want = pipes[b].step(recs[t])
            got = res[b]
            tag = (t, b)
            assert got["n_tracked"] == want["n_tracked"] and got["n_inliers"] == want["n_inliers"], tag
            assert got["retrack"] == bool(want["retrack"]) and got["retracked_on_device"] == bool(want["retrack"]), tag
            assert got["detect_overflow"] == 0, tag
            assert np.array_equal(eng.lane_features(b), pipes[b].blobCoord), tag
            if got["retrack"]:
                n_rt += 1
                assert got["n_after_retrack"] == len(pipes[b].blobCoord), tag
            _same_pose(got, want, tag)
            kf = eng.live_keyframe(b)
            assert kf["prunedUndistortedLocals"].shape == pipes[b].old_kf.prunedUndistortedLocals.shape, tag
            assert np.abs(kf["prunedUndistortedLocals"] - pipes[b].old_kf.prunedUndistortedLocals).max() <= 1e-4, tag
    assert n_rt >= 4
    eng.close()
    ctx.close()


def test_large_chunk_takes_the_one_sweep_integral_kernel():
    """chunks of >= 200 detections use rt_integral_kernel (one sweep, image written once), smaller ones the two-pass kernels
    (retrack.hip: rt_one_sweep): 224 lanes on three distinct sequences all run out of features in the same step; every lane
    must end with the oracle's features for its sequence, bit for bit - and so must the same lanes through 2-detection chunks"""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    seqs = [synth.make_sequence(31 + k, 2, n_movers=6, distortion=True) for k in range(3)]
    B = 224
    ctx = _ffi.Context(0)
    want = []
    for recs, poses, feat in seqs:
        pipe = oracle.OdometryPipeline(recs[0], feat[:24], poses[0], detect=_detect)
        w = pipe.step(recs[1])
        assert w["retrack"]
        want.append((w, pipe.blobCoord.copy()))
    for slots, lanes in ((B, B), (2, 6)):
        eng = Engine(lanes, 2 * lanes, ctx=ctx, retrack_on_device=True, retrack_slots=slots)
        for k, (recs, poses, feat) in enumerate(seqs):
            for t in range(2):
                eng.upload_scan(2 * k + t, recs[t])
        for b in range(3, lanes):
            for t in range(2):
                eng.copy_scan(2 * b + t, 2 * (b % 3) + t)
        eng.synchronize()
        for b in range(lanes):
            recs, poses, feat = seqs[b % 3]
            eng.init_lane(b, 2 * b, feat[:24], poses[0])
        eng.step(np.arange(lanes, dtype=np.int32) * 2 + 1)
        res = eng.results()
        for name in ("doh_integral", "doh_det_maxima"):          # live event pairs around the first detection chunk of the step
            ms, n = eng.kernel_avg(name, 1)
            assert n == 1 and 0.0 < ms < 1000.0, (name, ms, n)
            m = eng.kernel_chunk_ms(name, 1)                     # ... and around every chunk: lanes / slots of them, all busy here
            assert m.shape == (1, -(-lanes // slots)) and (m > 0.0).all() and (m < 1000.0).all(), (name, m)
            assert abs(float(m[0, 0]) - ms) < 1e-6
        for b in range(lanes):
            w, blobs = want[b % 3]
            assert res[b]["retracked_on_device"] and res[b]["detect_overflow"] == 0, (slots, b)
            assert res[b]["n_after_retrack"] == len(blobs), (slots, b)
            assert np.array_equal(eng.lane_features(b), blobs), (slots, b)
            _same_pose(res[b], w, (slots, b))
        eng.close()
    ctx.close()


def test_detection_overflow_is_flagged_not_fatal():
    """a scan of white noise has far more determinant maxima than the 2048 candidates a detection keeps: the lane's result
    carries the overflow bits (roam_abi.h: flags bits 8..11), nothing hangs or faults, the lane keeps a usable feature set and
    a well-behaved neighbour lane in the same chunk is not disturbed"""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, feat = synth.make_sequence(41, 3, n_movers=4)
    rng = np.random.default_rng(7)
    noise = [r.copy() for r in recs]
    for r in noise:
        r[:, 11:] = rng.integers(0, 256, size=r[:, 11:].shape, dtype=np.uint8)
    ctx = _ffi.Context(0)
    eng = Engine(2, 6, ctx=ctx, retrack_on_device=True)
    for t in range(3):
        eng.upload_scan(t, noise[t])
        eng.upload_scan(3 + t, recs[t])
    eng.init_lane_detect(0, 0, poses[0])
    eng.init_lane_detect(1, 3, poses[0])
    cart = oracle.convertPolarImageToCartesian(recs[0][:, 11:11 + 2025].astype(np.float32) / np.float32(255.))
    want1 = oracle.append_dedupe(np.empty((0, 2)), _detect(cart))
    assert np.array_equal(eng.lane_features(1), want1)
    f0 = eng.lane_features(0)
    assert 0 < len(f0) <= 1024 and np.isfinite(f0).all() and (f0 >= 0).all() and (f0 < W).all()
    eng.set_retrack(2)                                       # both lanes detect again in the step, in one chunk
    eng.step([1, 4])
    res = eng.results()
    assert res[0]["retracked_on_device"] and res[0]["detect_overflow"] != 0
    assert res[1]["retracked_on_device"] and res[1]["detect_overflow"] == 0
    eng.set_retrack(1)
    eng.step([2, 5])
    res = eng.results()
    assert np.isfinite(res[0]["pose"]).all() and np.isfinite(res[1]["pose"]).all()
    eng.close()
    ctx.close()


def test_empty_scan_detects_nothing_and_keeps_running():
    """an all-zero scan has no maxima at all: the detection appends nothing (no candidates, an empty tree, no pairs), the lane
    reports zero features step after step and its neighbour is not disturbed"""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, feat = synth.make_sequence(43, 3, n_movers=4)
    dark = [r.copy() for r in recs]
    for r in dark:
        r[:, 11:] = 0
    ctx = _ffi.Context(0)
    eng = Engine(2, 6, ctx=ctx, retrack_on_device=True)
    for t in range(3):
        eng.upload_scan(t, dark[t])
        eng.upload_scan(3 + t, recs[t])
    eng.init_lane_detect(0, 0, poses[0])
    eng.init_lane(1, 3, feat, poses[0])
    assert len(eng.lane_features(0)) == 0
    pipe = oracle.OdometryPipeline(recs[0], feat, poses[0], detect=_detect)
    for t in (1, 2):
        eng.step([t, 3 + t])
        res = eng.results()
        want = pipe.step(recs[t])
        assert res[0]["n_tracked"] == 0 and res[0]["n_after_retrack"] == 0 and res[0]["detect_overflow"] == 0, t
        assert np.isfinite(res[0]["pose"]).all()
        assert res[1]["n_inliers"] == want["n_inliers"], t
        _same_pose(res[1], want, t)
    eng.close()
    ctx.close()


def test_device_retrack_on_the_reference_real_scans():
    """the reference's 11 real data/tiny scans through a 1-lane engine with device-side detection and retracks (features
    collapse from ~200 to < 60 within two or three real frames) vs the oracle's loop body, every step"""
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    fix = np.load(FIX)
    pay = fix["payload"]
    T, rows, clip = pay.shape
    ctx = _ffi.Context(0)
    eng = Engine(1, T, ctx=ctx, rows=rows, stride=clip, payload_off=0, clip=clip, retrack_on_device=True)
    for t in range(T):
        eng.upload_scan(t, np.ascontiguousarray(pay[t]))
    pose0 = np.zeros(3)
    eng.init_lane_detect(0, 0, pose0)
    cart0 = oracle.convertPolarImageToCartesian(pay[0].astype(np.float32) / np.float32(255.))
    feat0 = oracle.append_dedupe(np.empty((0, 2)), _detect(cart0))
    assert np.array_equal(eng.lane_features(0), feat0)
    pipe = oracle.OdometryPipeline(np.ascontiguousarray(pay[0]), feat0, pose0, detect=_detect, payload_off=0, clip=clip)
    n_rt = 0
    for t in range(1, T):
        eng.step([t])
        got = eng.results()[0]
        want = pipe.step(np.ascontiguousarray(pay[t]))
        assert got["n_tracked"] == want["n_tracked"] and got["n_good"] == want["n_good"] and got["n_inliers"] == want["n_inliers"], t
        assert got["retrack"] == bool(want["retrack"]), t
        assert np.array_equal(eng.lane_features(0), pipe.blobCoord), t
        _same_pose(got, want, t)
        n_rt += got["retrack"]
    assert n_rt >= 2, n_rt
    # the vehicle of data/tiny drives ~2 m per frame: ten frames of dead-reckoned motion-distortion poses
    assert 10.0 < np.hypot(*got["pose"][:2]) < 40.0, got["pose"]
    eng.close()
    ctx.close()


def test_step_results_ring_does_not_drain_the_pipeline():
    """every step's records stay retrievable (ring of 8) while later steps are enqueued: results(step) == the records a
    synchronised run produced for that step"""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, feat = synth.make_sequence(3, 7, n_movers=4, distortion=True)
    T = len(recs)
    ctx = _ffi.Context(0)

    def run(sync):
        eng = Engine(2, T, ctx=ctx, retrack_on_device=True)
        for t in range(T):
            eng.upload_scan(t, recs[t])
        for b in range(2):
            eng.init_lane(b, 0, feat[:70 + 300 * b], poses[0])
        out = []
        for t in range(1, T):
            eng.step([t, t])
            if sync:
                out.append(eng.results_array())
        if not sync:
            assert eng.steps_enqueued() == T - 1
            out = [eng.results_array(step=k) for k in range(T - 1)]
        eng.close()
        return out

    a, b = run(True), run(False)
    for k, (x, y) in enumerate(zip(a, b)):
        assert x.tobytes() == y.tobytes(), k
    ctx.close()


def _blob_payload(clip, seed, rows=400):
    """speckle + Gaussian blobs in (azimuth, range), a third of them at the far end of the range axis = along the borders of the
    Cartesian image, where the box corners of the determinant are clipped"""
    rng = np.random.default_rng(seed)
    img = rng.exponential(6.0, size=(rows, clip))
    for i in range(max(18, clip // 5)):
        a, A = rng.uniform(0, rows), rng.uniform(80, 200)
        r = rng.uniform(clip - 24, clip - 2) if i % 3 == 0 else rng.uniform(4, clip - 2)
        sa, sr = max(1.5, 240.0 / max(r, 4.0)), rng.uniform(2.0, 7.0)
        aa = np.arange(int(a - 4 * sa), int(a + 4 * sa) + 1)
        rr = np.arange(max(0, int(r - 4 * sr)), min(clip, int(r + 4 * sr) + 1))
        img[np.ix_(aa % rows, rr)] += A * np.exp(-0.5 * ((aa - a) / sa) ** 2)[:, None] * np.exp(-0.5 * ((rr - r) / sr) ** 2)[None, :]
    return np.clip(np.floor(img), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("clip", [132, 300, 508, 1020])
def test_detection_on_image_sizes_that_do_not_fill_the_strips(clip):
    """the determinant kernel marches 126-column strips in steps of 32 rows: images of 132 ... 1020 pixels end inside a strip and
    inside a step (last strip 6 ... 12 columns wide), most of their positions clip box corners at the border, and blobs sit on
    the border itself - detections must equal the oracle's getFeatures"""
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    pay = _blob_payload(clip, clip)
    ctx = _ffi.Context(0)
    eng = Engine(2, 2, ctx=ctx, rows=pay.shape[0], stride=clip, payload_off=0, clip=clip, retrack_on_device=True)
    eng.upload_scan(0, pay)
    eng.upload_scan(1, np.ascontiguousarray(pay[::-1]))
    for b in range(2):
        eng.init_lane_detect(b, b, np.zeros(3))
        src = pay if b == 0 else np.ascontiguousarray(pay[::-1])
        cart = oracle.convertPolarImageToCartesian(src.astype(np.float32) / np.float32(255.))
        want = oracle.append_dedupe(np.empty((0, 2)), _detect(cart))
        got = eng.lane_features(b)
        assert len(want) >= 15, len(want)
        assert np.array_equal(got, want), (clip, b, len(got), len(want))
    eng.close()
    ctx.close()


def test_batched_first_detection_equals_lane_by_lane():
    """roam_engine_init_lanes_detect (one warp / pyramid / detection pass over n lanes, 5 lanes through 2 scratch slots) leaves
    exactly the state n calls of roam_engine_init_lane_detect leave: features, keyframes, and the next step's records"""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, _ = synth.make_sequence(23, 3, n_movers=8, distortion=True)
    ctx = _ffi.Context(0)
    B = 5
    idx = [0, 1, 0, 1, 1]
    p0 = np.array([poses[i] for i in idx])

    def run(batched):
        eng = Engine(B, 3, ctx=ctx, retrack_on_device=True, retrack_slots=2)
        for t in range(3):
            eng.upload_scan(t, recs[t])
        if batched:
            eng.init_lanes_detect(1, idx[1:4], p0[1:4])          # a range in the middle, then the rest lane by lane
            eng.init_lane_detect(0, idx[0], p0[0])
            eng.init_lanes_detect(4, idx[4:], p0[4:])
        else:
            for b in range(B):
                eng.init_lane_detect(b, idx[b], p0[b])
        feats = [eng.lane_features(b) for b in range(B)]
        kfs = [eng.live_keyframe(b) for b in range(B)]
        eng.step([i + 1 for i in idx])
        out = eng.results_array().tobytes()
        eng.close()
        return feats, kfs, out

    fa, ka, ra = run(False)
    fb, kb, rb = run(True)
    for b in range(B):
        assert len(fa[b]) >= 180 and np.array_equal(fa[b], fb[b]), b
        for k in ("pose", "velocity", "prunedUndistortedLocals"):
            assert np.array_equal(ka[b][k], kb[b][k]), (b, k)
        assert ka[b]["scan"] == kb[b]["scan"]
    assert ra == rb
    ctx.close()


def test_lanes_that_grow_past_320_features_are_tracked_in_full():
    """a forced detection (set_retrack(2)) on lanes that already hold ~200 features takes them past the 320 the host used to
    assume for device-side retracks: the next step must track ALL of them (dead reckoning, so that the pose depends on the
    tracker alone) - counts and pose equal the oracle's on the same feature set"""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, _ = synth.make_sequence(29, 4, n_movers=6)
    ctx = _ffi.Context(0)
    eng = Engine(2, 4, ctx=ctx, motion_distortion=False, retrack_on_device=True)
    for t in range(4):
        eng.upload_scan(t, recs[t])
    eng.init_lanes_detect(0, [0, 0], np.array([poses[0], poses[0]]))
    eng.set_retrack(2)
    eng.step([1, 1])                                          # ~150 inliers + up to 220 new blobs per lane
    eng.set_retrack(1)
    res1 = eng.results()
    feats = [eng.lane_features(b) for b in range(2)]
    assert min(len(f) for f in feats) > 320, [len(f) for f in feats]
    eng.step([2, 2])
    res2 = eng.results()
    for b in range(2):
        pipe = oracle.OdometryPipeline(recs[1], feats[b], res1[b]["pose"], motion_distortion=False, detect=_detect)
        want = pipe.step(recs[2])
        got = res2[b]
        assert got["n_tracked"] == len(feats[b]) == want["n_tracked"], b
        assert got["n_good"] == want["n_good"] and got["n_inliers"] == want["n_inliers"], (b, got["n_good"], want["n_good"])
        _same_pose(got, want, b)
    eng.close()
    ctx.close()


def test_new_sequence_flag_restarts_a_lane_inside_the_step():
    """scan_idx | ROAM_STEP_NEW_SEQUENCE: the lane drops its features before the pair - nothing is tracked, the pose stays, the
    first-frame detection runs on this scan (retrack path) - while its neighbour goes on; the following pairs equal the
    oracle's loop body started from that state"""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, feat = synth.make_sequence(31, 5, n_movers=6, distortion=True)
    ctx = _ffi.Context(0)
    eng = Engine(2, 5, ctx=ctx, retrack_on_device=True)
    for t in range(5):
        eng.upload_scan(t, recs[t])
    pipes = []
    for b in range(2):
        eng.init_lane(b, 0, feat, poses[0])
        pipes.append(oracle.OdometryPipeline(recs[0], feat, poses[0], detect=_detect))
    for t in (1, 2, 3, 4):
        restart = t == 2
        eng.step([t | (_ffi.STEP_NEW_SEQUENCE if restart else 0), t])
        res = eng.results()
        if restart:
            pipes[0].blobCoord = np.empty((0, 2), np.float32)            # the new sequence starts with no features
        for b in range(2):
            want = pipes[b].step(recs[t])
            got = res[b]
            assert got["n_tracked"] == want["n_tracked"] and got["n_inliers"] == want["n_inliers"], (t, b)
            assert got["retrack"] == bool(want["retrack"]), (t, b)
            assert np.array_equal(eng.lane_features(b), pipes[b].blobCoord), (t, b)
            _same_pose(got, want, (t, b))
        if restart:
            assert res[0]["n_tracked"] == 0 and res[0]["retracked_on_device"] and 180 <= res[0]["n_after_retrack"] <= 220
            assert not res[1]["retracked_on_device"]
    eng.close()
    ctx.close()


def test_one_sweep_integral_kernel_on_a_small_image():
    """208 first detections in one pass (>= 200: the one-sweep integral kernel with its footprint table, fast and general box
    staging, the tail of the row chain) on a 300 x 300 image, lanes alternating between two scans: every lane's features equal the
    oracle's"""
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    clip, B = 300, 208
    pay = [_blob_payload(clip, 7), np.ascontiguousarray(_blob_payload(clip, 7)[::-1])]
    ctx = _ffi.Context(0)
    eng = Engine(B, 2, ctx=ctx, rows=400, stride=clip, payload_off=0, clip=clip, retrack_on_device=True, retrack_slots=B)
    for t in range(2):
        eng.upload_scan(t, pay[t])
    eng.init_lanes_detect(0, [b % 2 for b in range(B)], np.zeros((B, 3)))
    want = []
    for t in range(2):
        cart = oracle.convertPolarImageToCartesian(pay[t].astype(np.float32) / np.float32(255.))
        want.append(oracle.append_dedupe(np.empty((0, 2)), _detect(cart)))
    for b in (0, 1, 2, 101, 206, 207):
        assert np.array_equal(eng.lane_features(b), want[b % 2]), b
    eng.close()
    ctx.close()


@pytest.mark.parametrize("nb,amp,sq,seed", [(40, 6, 3, 140), (90, 3, 2, 11)])
def test_detection_of_the_full_candidate_class_matches_get_features(nb, amp, sq, seed):
    """a detection with more than 1024 and at most 2048 determinant maxima takes rt_blobs_kernel<false> (global-memory pair lists, the
    big set tables) instead of the LDS-resident class every other test reaches: 1 385 / 1 792 candidates on these two scans, within the
    32 767 candidate pairs the bookkeeping holds - the features are the oracle's getFeatures (skimage pruning order, NumPy-1.22 sigma
    order, SSC)"""
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    clip = 2025
    rng = np.random.default_rng(seed)
    pay = (rng.random((400, clip)) * amp).astype(np.uint8)
    for _ in range(nb):
        a, r = int(rng.integers(0, 400)), int(rng.integers(200, clip - 5))
        pay[max(0, a - sq):a + sq, max(0, r - sq):r + sq] = rng.integers(150, 255)
    ctx = _ffi.Context(0)
    eng = Engine(1, 1, ctx=ctx, rows=400, stride=clip, payload_off=0, clip=clip, retrack_on_device=True)
    eng.upload_scan(0, pay)
    eng.init_lane_detect(0, 0, np.zeros(3))
    got = eng.lane_features(0)
    eng.step([0])                                                             # (the scratch of debug_detect is that of a step's detection)
    n = eng.debug_detect(False, 1)[0]
    assert 1024 < n[0] <= 2048, n
    cart = oracle.convertPolarImageToCartesian(pay.astype(np.float32) / np.float32(255.))
    want = oracle.append_dedupe(np.empty((0, 2)), _detect(cart))
    assert len(want) > 100 and np.array_equal(got, want), (len(got), len(want))
    eng.close()
    ctx.close()
