"""Keyframe / Map containers (reference Mapping.py:13-174): host-side bookkeeping around the
device stages (polar peaks every updateInfo, undistortion of the keyframe features)."""
import numpy as np

from .getPointCloud import getPointCloudPolarInd
from .motionDistortion import MotionDistortionSolver
from .utils import getRotationMatrix

ROT_THRESHOLD = 0.2
TRANS_THRESHOLD = 2.0
TRANS_THRESHOLD_SQ = TRANS_THRESHOLD * TRANS_THRESHOLD
RADAR_CART_CENTER = np.array([1012., 1012.])


class Keyframe():
    def __init__(self, globalPose, featurePointsLocal, radarPolarImg, velocity) -> None:
        self.updateInfo(globalPose, featurePointsLocal, radarPolarImg, velocity)

    def updateInfo(self, globalPose, featurePointsLocal, radarPolarImg, velocity) -> None:
        self.pose = globalPose
        self.radarPolarImg = radarPolarImg
        self.featurePointsLocal = featurePointsLocal
        self.prunedFeaturePoints = self.featurePointsLocal
        self.pointCloud = getPointCloudPolarInd(radarPolarImg)
        self.velocity = velocity
        self.featurePointsLocalUndistorted = MotionDistortionSolver.undistort(velocity, featurePointsLocal)[:, :2]
        self.prunedUndistortedLocals = self.featurePointsLocalUndistorted

    def getPrunedFeaturesGlobalPosition(self) -> np.ndarray:
        x, y, th = self.pose
        R = getRotationMatrix(th)
        t = np.array([x, y]).reshape(2, 1)
        return (R @ (self.prunedUndistortedLocals.T) + t).T

    def pruneFeaturePoints(self, corrStatus: np.ndarray) -> None:
        keep = corrStatus.flatten().astype(bool)
        self.prunedFeaturePoints = self.prunedFeaturePoints[keep]
        self.prunedUndistortedLocals = self.prunedUndistortedLocals[keep]


class Map():
    def __init__(self, sequenceName=None, estTraj=None, imgPathArr=(), filePaths=None) -> None:
        self.sequenceName = sequenceName
        self.imgPathArr = imgPathArr
        self.filePaths = filePaths
        self.estTraj = estTraj
        self.mapPoints = []
        self.keyframes = []

    def isGoodKeyframe(self, keyframe: Keyframe) -> bool:
        srcPose, targetPose = self.keyframes[-1].pose, keyframe.pose
        if np.abs(srcPose[2] - targetPose[2]) >= ROT_THRESHOLD:
            return True
        return bool(((srcPose[0:2] - targetPose[0:2]) ** 2).sum() >= TRANS_THRESHOLD_SQ)

    def addKeyframe(self, keyframe: Keyframe) -> None:
        self.keyframes.append(keyframe)
