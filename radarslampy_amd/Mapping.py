"""Keyframe / Map containers (reference Mapping.py:13-174): host-side bookkeeping around the
device stages (polar peaks on every updateInfo, undistortion of the keyframe features)."""
import numpy as np

from .getPointCloud import getPointCloudPolarInd
from .motionDistortion import MotionDistortionSolver

ROT_THRESHOLD = 0.2
TRANS_THRESHOLD = 2.0
TRANS_THRESHOLD_SQ = TRANS_THRESHOLD ** 2
RADAR_CART_CENTER = np.array([1012., 1012.])


def _se2_apply(pose, pts):
    x, y, th = pose
    c, s = np.cos(th), np.sin(th)
    return pts @ np.array([[c, s], [-s, c]]) + np.array([x, y])


class Keyframe():
    """pose [x,y,th] (m, m, rad), features in sensor-centred metres, the scan's polar peaks, velocity."""

    def __init__(self, globalPose, featurePointsLocal, radarPolarImg, velocity) -> None:
        self.updateInfo(globalPose, featurePointsLocal, radarPolarImg, velocity)

    def updateInfo(self, globalPose, featurePointsLocal, radarPolarImg, velocity) -> None:
        self.pose, self.velocity = globalPose, velocity
        self.radarPolarImg = radarPolarImg
        self.featurePointsLocal = self.prunedFeaturePoints = featurePointsLocal
        self.pointCloud = getPointCloudPolarInd(radarPolarImg)                                   # Mapping.py:62
        und = MotionDistortionSolver.undistort(velocity, featurePointsLocal)[:, :2]               # Mapping.py:65
        self.featurePointsLocalUndistorted = self.prunedUndistortedLocals = und

    def getPrunedFeaturesGlobalPosition(self) -> np.ndarray:
        return _se2_apply(self.pose, self.prunedUndistortedLocals)

    def pruneFeaturePoints(self, corrStatus: np.ndarray) -> None:
        keep = np.asarray(corrStatus).reshape(-1) != 0
        self.prunedFeaturePoints = self.prunedFeaturePoints[keep]
        self.prunedUndistortedLocals = self.prunedUndistortedLocals[keep]


class Map():
    def __init__(self, sequenceName=None, estTraj=None, imgPathArr=(), filePaths=None) -> None:
        self.sequenceName, self.estTraj = sequenceName, estTraj
        self.imgPathArr, self.filePaths = imgPathArr, filePaths
        self.mapPoints, self.keyframes = [], []

    def isGoodKeyframe(self, keyframe: Keyframe) -> bool:
        """rotation >= 0.2 rad or squared translation >= 4 m^2 w.r.t. the last keyframe (Mapping.py:149-174)"""
        last, cand = np.asarray(self.keyframes[-1].pose), np.asarray(keyframe.pose)
        if abs(last[2] - cand[2]) >= ROT_THRESHOLD:
            return True
        return bool(np.sum((last[:2] - cand[:2]) ** 2) >= TRANS_THRESHOLD_SQ)

    def addKeyframe(self, keyframe: Keyframe) -> None:
        self.keyframes.append(keyframe)


class DeviceMap():
    """the keyframes of one engine lane, resident in HBM (roam_engine_map_*).  Read-only view with the
    reference's attribute names; `keyframes[-1]` is the live keyframe the tracker is pruning."""

    class _KF():
        def __init__(self, d):
            self.pose, self.velocity = d["pose"], d["velocity"]
            self.prunedUndistortedLocals = d["prunedUndistortedLocals"]
            self.scan = d["scan"]

        def getPrunedFeaturesGlobalPosition(self) -> np.ndarray:
            return _se2_apply(self.pose, self.prunedUndistortedLocals)

    def __init__(self, engine, lane: int) -> None:
        self.engine, self.lane = engine, int(lane)

    def __len__(self) -> int:
        return self.engine.map_count(self.lane)

    @property
    def keyframes(self):
        return [DeviceMap._KF(d) for d in self.engine.map_keyframes(self.lane)]

    def isGoodKeyframe(self, pose) -> bool:
        """rotation >= 0.2 rad or squared translation >= 4 m^2 w.r.t. the live keyframe (Mapping.py:149-174)"""
        last = self.engine.map_keyframe(self.lane, len(self) - 1)["pose"]
        pose = np.asarray(pose)
        return bool(abs(last[2] - pose[2]) >= ROT_THRESHOLD or np.sum((last[:2] - pose[:2]) ** 2) >= TRANS_THRESHOLD_SQ)
