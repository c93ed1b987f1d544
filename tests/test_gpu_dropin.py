"""GPU (-m gpu): the drop-in modules (reference names / signatures / return orders) end to end."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def test_tracker_track_glue_matches_reference_golden(golden, monkeypatch):
    g = golden("tracker_glue")
    from radarslampy_amd import Tracker as T
    st = g["klt_status"]
    good = st.flatten().astype(bool)
    p, n = g["prev"], g["new"]
    monkeypatch.setattr(T, "getTrackedPointsKLT", lambda a, b, c: (n[good], p[good], n[~good], p[~good], st.copy()))
    tr = T.Tracker("x", ["a", "b"], {}, {"rejectOutliers": True, "useFMT": False})
    g_old, g_new, ang, cs = tr.track(None, None, None, None, p, 1)
    assert np.array_equal(g_old, g["good_old"]) and np.array_equal(g_new, g["good_new"])
    assert np.array_equal(cs, g["corrStatus"]) and cs.dtype == np.uint8 and cs.shape == (len(p), 1)
    R, h = tr.getTransform(g_old, g_new, pixel=False)
    assert np.abs(h - g["h"]).max() <= 1e-4 and np.abs(R - g["R"]).max() <= 1e-5
    assert isinstance(ang, float)


def test_keyframe_glue_matches_reference_golden(golden):
    g = golden("keyframe_glue")
    from radarslampy_amd.Mapping import Keyframe
    from gen_inputs import synthetic_polar_u8
    polar = synthetic_polar_u8(0)[:, :2025].astype(np.float32) / 255.
    k = Keyframe(g["pose"], g["feats"], polar, g["vel"])
    assert np.allclose(k.featurePointsLocalUndistorted, g["undist"], atol=1e-12)
    k.pruneFeaturePoints(g["corrStatus"])
    assert np.allclose(k.getPrunedFeaturesGlobalPosition(), g["pruned_global"], atol=1e-12)
    assert k.pointCloud.shape[0] == int(g["n_cloud"]) and k.pointCloud.dtype == np.int64


def test_full_track_call_on_images(golden):
    """Tracker.track on real-scan Cartesian images == oracle track (KLT -> err<10 -> clique)."""
    from radarslampy_amd.Tracker import Tracker
    from radarslampy_amd.parseData import convertPolarImageToCartesian
    g = golden("peaks")
    pa, pb = g["real0_u8"].astype(np.float32) / 255., g["real1_u8"].astype(np.float32) / 255.
    ca, cb = convertPolarImageToCartesian(pa), convertPolarImageToCartesian(pb)
    assert np.array_equal(ca, oracle.convertPolarImageToCartesian(pa))
    rng = np.random.default_rng(2)
    ys, xs = np.nonzero(ca > 0.25)
    sel = rng.choice(len(ys), 250, replace=False)
    pts = np.column_stack((xs[sel], ys[sel])).astype(np.float32)
    tr = Tracker("seq", ["a", "b"], {}, {"rejectOutliers": True})
    g_old, g_new, ang, cs = tr.track(ca, cb, pa, pb, pts, 1)
    w_old, w_new, _, w_cs = oracle.track_glue(oracle.getTrackedPointsKLT(ca, cb, pts))
    assert np.array_equal(g_old, w_old) and np.array_equal(g_new, w_new) and np.array_equal(cs, w_cs)
    assert g_old.dtype == np.float32 and len(g_old) > 30


def test_mds_class_matches_reference_golden(golden):
    from radarslampy_amd.motionDistortion import MotionDistortionSolver
    g = golden("mds")
    M = MotionDistortionSolver(np.diag([4, 4]), np.diag([1, 1, (5 * np.pi / 180) ** 2]))
    M.update_problem(g["n150_T0"], g["n150_p_w"], g["n150_p_jt"], g["n150_Tinit"])
    assert np.allclose(M.dT, g["n150_dT"], atol=1e-15) and np.allclose(M.info_vector, g["n150_info"])
    sol = M.optimize_library()
    assert np.abs(sol[3:5] - g["n150_sol"][3:5]).max() <= 1e-4 and abs(sol[5] - g["n150_sol"][5]) <= 1e-5
    assert np.allclose(MotionDistortionSolver.undistort(g["n150_truth"][:3], g["n150_p_jt"]), g["n150_undist"], atol=1e-12)


def test_misc_dropins(golden):
    from radarslampy_amd.ANMS import ssc
    from radarslampy_amd.outlierRejection import rejectOutliers
    from radarslampy_amd.getPointCloud import getPointCloudPolarInd
    from radarslampy_amd.getTransformKLT import calculateTransformSVD
    s = golden("ssc")
    assert np.array_equal(ssc(s["b500_kp"], 200, 0.1, 2024, 2024), s["b500_sel"])
    o = golden("outliers")
    pp, nn, mask = rejectOutliers(o["u128_prev"], o["u128_new"])
    assert np.array_equal(mask, o["u128_mask"]) and mask.dtype == bool and np.array_equal(pp, o["u128_prev"][mask])
    pk = golden("peaks")
    out = getPointCloudPolarInd(pk["f32img"])
    assert out.dtype == np.int64 and np.array_equal(out, pk["f32img_out"])
    k = golden("kabsch")
    R, h = calculateTransformSVD(k["real95_f32_src"], k["real95_f32_tgt"])
    assert np.abs(h - k["real95_f32_h"]).max() * 0.0864 <= 1e-4


def test_raw_roam_system_driver_on_png_sequence(tmp_path):
    """The non-plotting RawROAMSystem driver over a synthetic Oxford-format sequence stored as PNGs:
    frame-by-frame poses equal the oracle's loop body (detect -> track -> reject -> Kabsch -> LM ->
    keyframe / retrack) within 1e-4 m / 1e-5 rad."""
    from PIL import Image
    from radarslampy_amd import synth
    from radarslampy_amd.RawROAMSystem import RawROAMSystem
    recs, poses, _ = synth.make_sequence(9, 4)
    root = tmp_path / "data"
    rad = root / "synth" / "radar"
    rad.mkdir(parents=True)
    stamps = [1547131046353776 + 250000 * i for i in range(len(recs))]
    with open(root / "synth" / "radar.timestamps", "w") as f:
        for s_, r in zip(stamps, recs):
            Image.fromarray(r).save(rad / f"{s_}.png")
            f.write(f"{s_} 1\n")
    sysm = RawROAMSystem("synth", {"rejectOutliers": True}, hasGroundTruth=False, dataRoot=str(root))
    sysm.run(0, -1, initPose=poses[0])
    feat0 = oracle.getFeatures(oracle.convertPolarImageToCartesian(recs[0][:, 11:11 + 2025].astype(np.float32) / 255.))[0]
    feat0 = oracle.append_dedupe(np.empty((0, 2)), feat0)
    pipe = oracle.OdometryPipeline(recs[0], feat0, poses[0], detect=lambda c: oracle.getFeatures(c)[0])
    assert len(sysm.frameLog) == 3
    for t, log in zip(range(1, 4), sysm.frameLog):
        want = pipe.step(recs[t])
        assert log["n_tracked"] == want["n_tracked"] and log["n_inliers"] == want["n_inliers"], t
        assert np.abs(log["pose"][:2] - want["pose"][:2]).max() <= 1e-4 and abs(log["pose"][2] - want["pose"][2]) <= 1e-5, t
        assert log["new_keyframe"] == bool(want["new_keyframe"])
    assert sysm.estTraj.poses.shape == (4, 3)
