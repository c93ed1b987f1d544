#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER.

Imports the read-only reference (Samleo8/RadarSLAMPy, /root/reference) with import stubs
for the three wheels that are absent here (cv2, skimage.feature, tkinter.messagebox),
feeds seeded inputs / the reference's own data files through its importable hot-path
functions and stores INPUTS + OUTPUTS as small .npz fixtures next to this script.

Nothing of the reference's source travels: only arrays.  The GPU box never runs this
file (it has no /root/reference); tests read the committed .npz files.

Functions exercised (reference file:line):
  getPointCloud.getPointCloudPolarInd      getPointCloud.py:11-54
  ANMS.ssc                                 ANMS.py:5-102
  getTransformKLT.calculateTransformSVD    getTransformKLT.py:129-162
  outlierRejection.rejectOutliers          outlierRejection.py:16-95
  motionDistortion.MotionDistortionSolver  motionDistortion.py:70-205,295-325
  utils.*                                  utils.py:29-103,147-165
  parseData.extractDataFromRadarImage      parseData.py:17-53
  Tracker.track (glue, KLT/FMT mocked)     Tracker.py:35-106
  Mapping.Keyframe (glue)                  Mapping.py:37-125
"""
import contextlib
import io
import os
import sys
import types
import hashlib

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.TERM_CRITERIA_EPS = 2
    cv2.TERM_CRITERIA_COUNT = 1
    sys.modules["cv2"] = cv2
    sk = types.ModuleType("skimage")
    skf = types.ModuleType("skimage.feature")
    skf.blob_doh = skf.blob_dog = skf.blob_log = None
    sk.feature = skf
    sys.modules["skimage"] = sk
    sys.modules["skimage.feature"] = skf
    tk = types.ModuleType("tkinter")
    tkm = types.ModuleType("tkinter.messagebox")
    tkm.NO = "no"
    tk.messagebox = tkm
    sys.modules["tkinter"] = tk
    sys.modules["tkinter.messagebox"] = tkm
    import matplotlib
    matplotlib.use("Agg")
    # matplotlib.ft2font.BOLD is imported by Tracker.py and is gone in matplotlib 3.10
    import matplotlib.ft2font as ft
    if not hasattr(ft, "BOLD"):
        ft.BOLD = 0


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# --------------------------------------------------------------------------------------
# seeded input generators shared with the tests (tests/golden/gen_inputs.py re-exports)
# --------------------------------------------------------------------------------------
sys.path.insert(0, OUT)
from gen_inputs import (synthetic_polar_u8, ssc_keypoints, rigid_pairs,  # noqa: E402
                        unique_clique_pairs, mds_problem)


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    os.chdir(REF)
    with quiet():
        import getPointCloud as r_pc
        import ANMS as r_anms
        import getTransformKLT as r_klt
        import outlierRejection as r_or
        import motionDistortion as r_md
        import utils as r_ut
        import parseData as r_pd
        import Tracker as r_tr
        import Mapping as r_map
    from PIL import Image

    # ---------------- record format + peaks on real scans ----------------
    radar_dir = os.path.join(REF, "data", "tiny", "radar")
    names = sorted(os.listdir(radar_dir))
    rec = {}
    peaks = {}
    for i, nm in enumerate(names[:2]):
        png = np.array(Image.open(os.path.join(radar_dir, nm)))
        assert png.shape == (400, 3779) and png.dtype == np.uint8
        polar, az, rres, ares, valid, ts = r_pd.extractDataFromRadarImage(png)
        assert polar.shape == (400, 2025) and polar.dtype == np.float32
        if i == 0:
            rec = dict(meta_u8=png[:, :11].copy(), payload_head_u8=png[:, 11:11 + 64].copy(),
                       azimuths=az, valid=valid, timestamps=ts,
                       range_resolution=np.float64(rres), azimuth_resolution=np.asarray(ares),
                       polar_head=polar[:, :64].copy())
        pts = r_pc.getPointCloudPolarInd(polar)
        peaks[f"real{i}_u8"] = png[:, 11:11 + 2025].copy()
        peaks[f"real{i}_out"] = pts.astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "record_format.npz"), **rec)

    for seed in (0, 1, 2):
        u8 = synthetic_polar_u8(seed)
        pts = r_pc.getPointCloudPolarInd(u8.astype(np.float32) / 255.)
        peaks[f"synth{seed}_sha"] = np.array(sha(u8))
        peaks[f"synth{seed}_out"] = pts.astype(np.int32)
    # generic float32 input (not k/255) - exercises the f32 entry point
    rng = np.random.default_rng(77)
    f32img = (rng.random((37, 513), dtype=np.float32) ** 3).astype(np.float32)
    f32img[5, :] = 0.25          # all-flat row: one big plateau touching both ends -> no peak
    f32img[6, :] = 0.0
    f32img[6, 100:104] = 0.5     # a single plateau peak -> midpoint (100+103)//2
    peaks["f32img"] = f32img
    peaks["f32img_out"] = r_pc.getPointCloudPolarInd(f32img).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "peaks.npz"), **peaks)

    # ---------------- SSC ----------------
    ssc = {}
    # NOTE: B < 180 is not a usable golden: the reference's binary search then shrinks the
    # width towards 0 and allocates a (2024/c)^2 Python list grid -> it runs out of memory.
    for tag, (B, seed, clustered) in dict(b500=(500, 4, False), b5000=(5000, 5, False),
                                          clus=(3000, 6, True), b230=(230, 7, False),
                                          b190=(190, 8, False), lattice=(400, 0, "lattice")).items():
        kp = ssc_keypoints(B, seed, clustered)
        sel = r_anms.ssc(kp, 200, 0.1, 2024, 2024)
        ssc[f"{tag}_kp"] = kp
        ssc[f"{tag}_sel"] = sel
    np.savez_compressed(os.path.join(OUT, "ssc.npz"), **ssc)

    # ---------------- Kabsch ----------------
    kab = {}
    # correspondence arrays embedded as literals in the reference's archive script (data only)
    src_txt = open(os.path.join(REF, "archive", "testTransformKLT2.py")).read()
    a0 = src_txt.index("srcCoord = np.array(")
    a1 = src_txt.index("targetCoord = np.array(")
    a2 = src_txt.index("dx,dy,dth")
    # the reference is untrusted public content: nothing of it is executed - the bracketed number lists are cut out and
    # parsed as literals
    import ast

    def literal_after(txt):
        lo = txt.index("(") + 1
        hi = txt.rindex(")")
        return np.asarray(ast.literal_eval(txt[lo:hi].strip().rstrip(",")), dtype=np.float64)

    real_src = literal_after(src_txt[a0:a1])
    real_tgt = literal_after(src_txt[a1:a2])
    assert real_src.shape == (95, 2) and real_tgt.shape == (95, 2)
    cases = {"real95_f64": (real_src, real_tgt),
             "real95_f32": (real_src.astype(np.float32), real_tgt.astype(np.float32))}
    for tag, (n, seed, noise, dt) in dict(clean100=(100, 11, 0.0, np.float64),
                                          noisy100=(100, 12, 1.5, np.float64),
                                          noisy250_f32=(250, 13, 1.0, np.float32),
                                          n3=(3, 14, 0.0, np.float64),
                                          n2=(2, 15, 0.0, np.float64)).items():
        s, t, _, _ = rigid_pairs(n, seed, noise)
        cases[tag] = (s.astype(dt), t.astype(dt))
    for tag, (s, t) in cases.items():
        R, h = r_klt.calculateTransformSVD(s, t)
        kab[f"{tag}_src"], kab[f"{tag}_tgt"] = s, t
        kab[f"{tag}_R"], kab[f"{tag}_h"] = np.asarray(R, np.float64), np.asarray(h, np.float64)
    np.savez_compressed(os.path.join(OUT, "kabsch.npz"), **kab)

    # ---------------- outlier rejection ----------------
    orj = {}
    d = np.load(os.path.join(REF, "outlier_test.npz"))
    tie_cases = {"npz139": (d["prev_coord"], d["new_coord"]),
                 "npz139b": (d["prev_old_coord"], d["prev_coord"]),
                 "real95": (real_src.astype(np.float32), real_tgt.astype(np.float32))}
    for tag, (p, n) in tie_cases.items():
        with quiet():
            pp, nn, mask = r_or.rejectOutliers(p, n)
        orj[f"{tag}_prev"], orj[f"{tag}_new"], orj[f"{tag}_mask"] = p, n, mask
        orj[f"{tag}_size"] = np.int32(mask.sum())
    for tag, (K, seed, frac) in dict(u64=(64, 21, 0.25), u128=(128, 22, 0.3), u256=(256, 23, 0.4),
                                     u40=(40, 24, 0.2)).items():
        p, n, inl = unique_clique_pairs(K, seed, frac)
        with quiet():
            pp, nn, mask = r_or.rejectOutliers(p, n)
        assert np.array_equal(mask, inl), tag
        orj[f"{tag}_prev"], orj[f"{tag}_new"], orj[f"{tag}_mask"] = p, n, mask
        orj[f"{tag}_size"] = np.int32(mask.sum())
    orj["thr_px"] = np.float64(r_or.DIST_THRESHOLD_PX)
    np.savez_compressed(os.path.join(OUT, "outliers.npz"), **orj)

    # ---------------- motion distortion ----------------
    md = {}
    cov_p = np.diag([4, 4])
    cov_v = np.diag([1, 1, (5 * np.pi / 180) ** 2])
    for tag, (N, seed, yaw, big) in dict(n60=(60, 31, 0.02, False), n150=(150, 32, 0.15, False),
                                         n250=(250, 33, 0.0, False), n150big=(150, 34, -0.6, True),
                                         n8=(8, 35, 0.05, False)).items():
        T0, p_w, p_jt, T_init, truth = mds_problem(N, seed, yaw, big)
        MDS = r_md.MotionDistortionSolver(cov_p, cov_v)
        MDS.update_problem(T0, p_w, p_jt, T_init)
        T0i = T_init
        x0 = np.hstack((MDS.v_j_initial, [T0i[0, 2], T0i[1, 2], np.arctan2(T0i[1, 0], T0i[0, 0])]))
        md[f"{tag}_T0"], md[f"{tag}_p_w"], md[f"{tag}_p_jt"], md[f"{tag}_Tinit"] = T0, p_w, p_jt, T_init
        md[f"{tag}_truth"] = truth
        md[f"{tag}_x0"] = x0
        md[f"{tag}_r0"] = MDS.error_vector(x0)
        md[f"{tag}_dT"] = MDS.dT
        md[f"{tag}_info"] = MDS.info_vector
        md[f"{tag}_sol"] = MDS.optimize_library()
        v = truth[:3]
        md[f"{tag}_undist"] = r_md.MotionDistortionSolver.undistort(v, p_jt)
    np.savez_compressed(os.path.join(OUT, "mds.npz"), **md)

    # ---------------- SE(2) utils ----------------
    ut = {}
    rng = np.random.default_rng(41)
    poses = rng.normal(size=(16, 3)) * np.array([50, 50, 2.0])
    ut["poses"] = poses
    ut["T"] = r_ut.convertPoseToTransform(poses)
    ut["poses_back"] = r_ut.convertTransformToPose(ut["T"])
    ang = rng.normal(size=64) * 6
    ut["ang"], ut["ang_norm"] = ang, r_ut.normalize_angles(ang)
    R = r_ut.getRotationMatrix(0.3)
    ut["deltas"] = r_ut.convertRandHtoDeltas(R, np.array([[1.5], [-0.25]]))
    ut["Tinv"] = np.stack([r_ut.invert_transform(T) for T in ut["T"]])
    ut["homog"] = r_ut.homogenize(poses[:, :2])
    np.savez_compressed(os.path.join(OUT, "se2_utils.npz"), **ut)

    # ---------------- Tracker.track glue (KLT + FMT mocked) ----------------
    tg = {}
    K = 120
    p, n, inl = unique_clique_pairs(K, 51, 0.25)
    rng = np.random.default_rng(52)
    status = (rng.random(K) > 0.15).astype(np.uint8).reshape(-1, 1)
    good = status.flatten().astype(bool)

    def fake_klt(srcImg, tgtImg, pts):
        return n[good], p[good], n[~good], p[~good], status.copy()

    r_tr.getTrackedPointsKLT = fake_klt
    r_tr.getRotationUsingFMT = lambda a, b: (0.0, 1.0, 0.0)
    tr = r_tr.Tracker("x", ["a", "b"], {}, {"rejectOutliers": True, "useFMT": False})
    with quiet():
        g_old, g_new, ang_, cs = tr.track(None, None, None, None, p, 1)
        R, h = tr.getTransform(g_old, g_new, pixel=False)
    tg.update(prev=p, new=n, klt_status=status, good_old=g_old, good_new=g_new, corrStatus=cs,
              R=R, h=h)
    np.savez_compressed(os.path.join(OUT, "tracker_glue.npz"), **tg)

    # ---------------- Keyframe glue ----------------
    kf = {}
    r_map.RADAR_CART_CENTER = np.array([1012., 1012.])   # avoid the cv2 warp in updateInfo
    rng = np.random.default_rng(61)
    feats = (rng.random((50, 2)) * 2024 - 1012) * 0.0864
    pose = np.array([12.0, -7.5, 0.4])
    vel = np.array([3.0, -0.2, 0.1])
    tiny_polar = (synthetic_polar_u8(0)[:, :2025].astype(np.float32) / 255.)
    k = r_map.Keyframe(pose, feats, tiny_polar, vel)
    cs = (rng.random(50) > 0.3).astype(np.uint8).reshape(-1, 1)
    k.pruneFeaturePoints(cs)
    kf.update(pose=pose, feats=feats, vel=vel, corrStatus=cs,
              undist=k.featurePointsLocalUndistorted, pruned_global=k.getPrunedFeaturesGlobalPosition(),
              n_cloud=np.int32(k.pointCloud.shape[0]))
    np.savez_compressed(os.path.join(OUT, "keyframe_glue.npz"), **kf)

    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
