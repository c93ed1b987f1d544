#!/usr/bin/env python3
"""Fixture generator.  RUNS ONLY IN THE BUILD CONTAINER (it imports the read-only reference).

tiny_hybrid.npz = the poses of the REFERENCE'S OWN loop - RawROAMSystem.run (RawROAMSystem.py:86-298) with its own Tracker glue,
outlier rejection (networkx), Keyframe / Map bookkeeping, MotionDistortionSolver (scipy least_squares) and Trajectory - on its own
data/tiny, with only the three functions it takes from wheels that are absent here replaced by the oracle's restatements:
    parseData.convertPolarImageToCartesian (cv2.warpPolar)      -> oracle.convertPolarImageToCartesian
    getTransformKLT.getTrackedPointsKLT (cv2.calcOpticalFlowPyrLK) -> oracle.getTrackedPointsKLT
    getFeatures.appendNewFeatures (skimage blob_doh + ANMS)     -> oracle.getFeatures + oracle.append_dedupe
(FMT's rotation is printed and unused: stubbed; plotting off.)  Stored: arrays only - the per-frame poses for HEAD's keyframe
thresholds and for "a keyframe on every frame" (Mapping.TRANS_THRESHOLD_SQ = 0), each without and with the one swapped near-tie of
frame 2's detector responses (tests/test_oracle_tiny_traj.py).  What it pins: oracle.OdometryPipeline's LOOP (a15 glue, a9, a11-a14)
against the reference's code, to float64 round-off instead of the three printed decimals."""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
REF = "/root/reference"


def run(swap, every_frame):
    import oracle
    from test_oracle_tiny_traj import _detect, _detect_with_swap
    from PIL import Image
    import parseData, getFeatures, getTransformKLT, FMT, Tracker, Mapping, RawROAMSystem as RRS          # noqa: E401
    frame = [0]
    detect = _detect_with_swap({2: 521}, frame) if swap else _detect

    def get_polar(paths, index):
        return parseData.extractDataFromRadarImage(np.array(Image.open(paths[index]).convert("L"), dtype=np.uint8))[0]

    def to_cart(polar, *a, **k):
        return oracle.convertPolarImageToCartesian(np.ascontiguousarray(polar, np.float32))

    def append_new(src, old):
        return oracle.append_dedupe(old, detect(src)), None

    for mod in (parseData, RRS, Mapping, Tracker, getTransformKLT, getFeatures):
        if hasattr(mod, "convertPolarImageToCartesian"): mod.convertPolarImageToCartesian = to_cart
        if hasattr(mod, "getPolarImageFromImgPaths"): mod.getPolarImageFromImgPaths = get_polar
        if hasattr(mod, "appendNewFeatures"): mod.appendNewFeatures = append_new
        if hasattr(mod, "getTrackedPointsKLT"): mod.getTrackedPointsKLT = oracle.getTrackedPointsKLT
        if hasattr(mod, "getRotationUsingFMT"): mod.getRotationUsingFMT = lambda *a, **k: (0.0, 1.0, 0.0)
    if not hasattr(Tracker.Tracker, "_orig_track"):
        Tracker.Tracker._orig_track = Tracker.Tracker.track

    def track(self, prevImg, currImg, prevPolar, currPolar, blobCoord, seqInd):
        frame[0] = seqInd
        return Tracker.Tracker._orig_track(self, prevImg, currImg, prevPolar, currPolar, blobCoord, seqInd)
    Tracker.Tracker.track = track
    RRS.RawROAMSystem.plot = lambda self, *a, **k: None
    Mapping.TRANS_THRESHOLD_SQ = 0.0 if every_frame else 4.0
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.symlink(os.path.join(REF, "data"), os.path.join(d, "data"))
        os.chdir(d)                                       # the constructor makes ./img/... : never inside the reference
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                s = RRS.RawROAMSystem("tiny", paramFlags={"rejectOutliers": True, "useANMS": True, "useFMT": False, "correctMotionDistortion": True})
                s.run(0, -1)
        finally:
            os.chdir(cwd)
    return np.array(s.estTraj.poses, np.float64)


def main():
    import make_goldens
    make_goldens._install_stubs()
    sys.path.insert(0, REF)
    out = {}
    for swap in (False, True):
        for every in (False, True):
            out["poses_%s_%s" % ("swap" if swap else "plain", "every_frame" if every else "head")] = run(swap, every)
    np.savez_compressed(os.path.join(HERE, "tiny_hybrid.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, np.round(v[6], 4))


if __name__ == "__main__":
    main()
