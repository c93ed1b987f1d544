"""GPU twin of tests/test_oracle_tiny_traj.py: the HIP engine (one lane, device-side detection and retracks, the device's
networkx-order clique) over the reference's 11 real data/tiny scans from the ground-truth start pose, against the numbers the
REFERENCE ITSELF printed into img/roam_mapping/tiny_traj/00NN.jpg (fixture tests/golden/tiny_traj.npz): frames 1-3 to print
precision (1e-3 m, 1e-3 deg), every frame equal to the oracle's loop body within the north_star tolerance."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PRINT = 1.1e-3
# HEAD's keyframe policy against the pictures' (a keyframe on every frame) and the one swapped near-tie of frame 2: what this code's
# poses differ from the printed ones by, frame by frame - [x m, y m, theta deg], oracle (= engine to 1e-4 m / 1e-5 rad) minus print.
# Frames 4-5: the near-tie of frame 2's detector responses taken in the other order (it renumbers the features: another of frame 4's nine
# tied cliques); 6-8: the pair solved against a two-frame-old keyframe; 9-10: a second near-tie, at the re-detection of frame 7 (one
# feature exchanged: profiles/r06_frame_markers_8_10.txt, r06_frame8_swap_search.txt).  With both swaps and the pictures' keyframe policy
# the oracle prints ALL TEN frames (tests/test_oracle_tiny_traj.py::test_two_swapped_near_ties_reproduce_all_ten_printed_frames); neither
# can be injected into the engine, so here the KNOWN differences are pinned - to 1.6e-3 (1e-3 + the prints' rounding) instead of the
# blanket 0.15 m / 0.3 deg of rounds 4-5.
KNOWN_DIFF = {4: (0.016327, 0.026739, -0.029540), 5: (0.016128, 0.024040, -0.028859), 6: (0.019532, -0.061734, 0.095391),
              7: (0.025198, -0.058284, 0.089876), 8: (0.026503, -0.054397, 0.092879), 9: (-0.010646, -0.050218, 0.192600),
              10: (-0.082521, -0.068067, 0.238190)}


def test_engine_reproduces_the_reference_prints_on_data_tiny():
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    traj = np.load(os.path.join(HERE, "golden", "tiny_traj.npz"))
    pay = np.load(os.path.join(HERE, "golden", "tiny_track.npz"))["payload"]
    # the reference's OWN loop code (RawROAMSystem.run at HEAD) run with the oracle's front end - tests/golden/make_tiny_hybrid.py: the
    # engine against the reference's loop in one hop
    hybrid = np.load(os.path.join(HERE, "golden", "tiny_hybrid.npz"))["poses_plain_head"]
    T, rows, clip = pay.shape
    det = lambda c: oracle.getFeatures(c)[0]                                    # noqa: E731
    ctx = _ffi.Context(0)
    eng = Engine(1, T, ctx=ctx, rows=rows, stride=clip, payload_off=0, clip=clip, retrack_on_device=True)
    for t in range(T):
        eng.upload_scan(t, np.ascontiguousarray(pay[t]))
    pose0 = traj["gt_pose"][0]
    eng.init_lane_detect(0, 0, pose0)
    cart0 = oracle.convertPolarImageToCartesian(pay[0].astype(np.float32) / np.float32(255.))
    feat0 = oracle.append_dedupe(np.empty((0, 2)), det(cart0))
    pipe = oracle.OdometryPipeline(np.ascontiguousarray(pay[0]), feat0, pose0, detect=det, payload_off=0, clip=clip)
    est = [pose0]
    retracks = []
    for t in range(1, T):
        eng.step([t])
        got = eng.results()[0]
        want = pipe.step(np.ascontiguousarray(pay[t]))
        assert (got["n_tracked"], got["n_good"], got["n_inliers"]) == (want["n_tracked"], want["n_good"], want["n_inliers"]), t
        assert got["clique_proven"], t
        assert np.abs(got["pose"][:2] - want["pose"][:2]).max() <= 1e-4 and abs(got["pose"][2] - want["pose"][2]) <= 1e-5, t
        assert np.abs(got["pose"][:2] - hybrid[t][:2]).max() <= 1.2e-4 and abs(got["pose"][2] - hybrid[t][2]) <= 1.2e-5, (t, got["pose"], hybrid[t])
        est.append(np.array(got["pose"]))
        printed = np.array([got["pose"][0], got["pose"][1], np.rad2deg(got["pose"][2])])
        d = np.abs(printed - traj["roam_mapping_est_pose"][t - 1])
        rmse = float(np.sqrt(np.mean(((traj["gt_pose"][:t + 1, :2] - np.array(est)[:, :2]) ** 2).sum(1))))
        if t <= 3:
            assert d.max() <= PRINT, (t, printed, traj["roam_mapping_est_pose"][t - 1])          # the reference's own print
            assert abs(rmse - traj["roam_mapping_rmse"][t - 1]) <= 5.1e-3, t
        else:
            sd = printed - traj["roam_mapping_est_pose"][t - 1]
            assert np.abs(sd - np.array(KNOWN_DIFF[t])).max() <= 1.6e-3, (t, sd, KNOWN_DIFF[t])       # the known difference, not a blanket
            assert abs(rmse - traj["roam_mapping_rmse"][t - 1]) < 0.02, t
        if got["retrack"]:
            retracks.append(t)
    assert retracks == [1, 2, 4, 7, 9], retracks           # the frames whose reference picture shows freshly appended features
    eng.close()
    ctx.close()


def test_engine_with_a_keyframe_on_every_frame_reproduces_the_pictured_deltas():
    """The reference's pictures were made by a run that added a keyframe on every frame (tests/test_oracle_tiny_traj.py,
    DESIGN.md section 4).  The engine with Map.isGoodKeyframe's translation threshold at 1e-9 m (roam_engine_cfg.keyframe_trans_m):
    every step equals the oracle's loop body under the same policy (1e-4 m / 1e-5 rad), frames 1-3 print the reference's poses,
    and the per-frame EST Deltas - which do not carry the 16 / 27 mm the nine-way clique tie of frame 4 leaves in the dead-reckoned
    pose - equal the reference's prints on frames 1-3 and 5-7 to 2 units of the last printed digit, on frame 8 to 3."""
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    traj = np.load(os.path.join(HERE, "golden", "tiny_traj.npz"))
    pay = np.load(os.path.join(HERE, "golden", "tiny_track.npz"))["payload"]
    T, rows, clip = pay.shape
    det = lambda c: oracle.getFeatures(c)[0]                                    # noqa: E731
    ctx = _ffi.Context(0)
    eng = Engine(1, T, ctx=ctx, rows=rows, stride=clip, payload_off=0, clip=clip, retrack_on_device=True, keyframe_trans_m=1e-9)
    for t in range(T):
        eng.upload_scan(t, np.ascontiguousarray(pay[t]))
    pose0 = traj["gt_pose"][0]
    eng.init_lane_detect(0, 0, pose0)
    cart0 = oracle.convertPolarImageToCartesian(pay[0].astype(np.float32) / np.float32(255.))
    feat0 = oracle.append_dedupe(np.empty((0, 2)), det(cart0))
    pipe = oracle.OdometryPipeline(np.ascontiguousarray(pay[0]), feat0, pose0, detect=det, payload_off=0, clip=clip, keyframe_trans_m=1e-9)
    prev = np.array(pose0)
    for t in range(1, 9):
        eng.step([t])
        got = eng.results()[0]
        want = pipe.step(np.ascontiguousarray(pay[t]))
        assert got["new_keyframe"] and want["new_keyframe"], t
        assert (got["n_tracked"], got["n_good"], got["n_inliers"]) == (want["n_tracked"], want["n_good"], want["n_inliers"]), t
        assert np.abs(got["pose"][:2] - want["pose"][:2]).max() <= 1e-4 and abs(got["pose"][2] - want["pose"][2]) <= 1e-5, t
        Trel = np.linalg.inv(oracle.convertPoseToTransform(prev)) @ oracle.convertPoseToTransform(got["pose"])
        deltas = np.array([Trel[0, 2], Trel[1, 2], np.rad2deg(np.arctan2(Trel[1, 0], Trel[0, 0]))])
        d = np.abs(deltas - traj["roam_mapping_est_deltas"][t - 1])
        if t != 4:
            assert d.max() <= (2.1e-3 if t <= 7 else 3.1e-3), (t, deltas, traj["roam_mapping_est_deltas"][t - 1])
        if t <= 3:
            printed = np.array([got["pose"][0], got["pose"][1], np.rad2deg(got["pose"][2])])
            assert np.abs(printed - traj["roam_mapping_est_pose"][t - 1]).max() <= PRINT, t
        prev = np.array(got["pose"])
    eng.close()
    ctx.close()
