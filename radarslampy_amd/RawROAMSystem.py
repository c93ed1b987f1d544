"""Non-plotting counterpart of the reference driver (reference RawROAMSystem.py:20-333): same class
name, constructor and run() semantics, the loop body of :162-298 on top of the drop-in modules, with
the paramFlags actually honoured (`rejectOutliers`, `correctMotionDistortion`).  §8f-f3 "next" row:
plotting, video export and the CLI are out of scope."""
import os

import numpy as np

from .getFeatures import N_FEATURES_BEFORE_RETRACK, appendNewFeatures
from .Mapping import Keyframe, Map
from .motionDistortion import MotionDistortionSolver
from .parseData import (RANGE_RESOLUTION_CART_M, convertPolarImageToCartesian, getPolarImageFromImgPaths,
                        getRadarImgPaths)
from .Tracker import Tracker
from .trajectoryPlotting import Trajectory, computePosesRMSE, getGroundTruthTrajectory
from .utils import convertPoseToTransform, convertRandHtoDeltas, radarImgPathToTimestamp

RADAR_CART_CENTER = np.array([1012, 1012])


class RawROAMSystem():
    def __init__(self, sequenceName: str, paramFlags: dict = None, hasGroundTruth: bool = True, dataRoot: str = "data") -> None:
        self.sequenceName = sequenceName
        self.paramFlags = dict(paramFlags or {})
        self.hasGroundTruth = hasGroundTruth
        dataPath = os.path.join(dataRoot, sequenceName, "radar")
        timestampPath = os.path.join(dataRoot, sequenceName, "radar.timestamps")
        assert os.path.exists(dataPath), "Failed to find radar data for sequence " + sequenceName
        assert os.path.exists(timestampPath), "Failed to find radar timestamp information for sequence " + sequenceName
        self.dataRoot = dataRoot
        self.imgPathArr = getRadarImgPaths(dataPath, timestampPath)
        self.sequenceSize = len(self.imgPathArr)
        self.filePaths = {"data": dataPath, "timestamp": timestampPath}
        self.gtTraj = None
        self.estTraj = None
        self.tracker = Tracker(self.sequenceName, self.imgPathArr, self.filePaths, self.paramFlags)
        self.map = Map(self.sequenceName, self.estTraj, self.imgPathArr, self.filePaths)
        self.frameLog = []

    def run(self, startSeqInd: int = 0, endSeqInd: int = -1, initPose=None) -> None:
        imgPathArr, tracker = self.imgPathArr, self.tracker
        assert 0 <= startSeqInd < self.sequenceSize
        if endSeqInd < 0:
            endSeqInd = self.sequenceSize - 1
        assert endSeqInd < self.sequenceSize and startSeqInd <= endSeqInd
        initTimestamp = radarImgPathToTimestamp(imgPathArr[startSeqInd])
        gtPath = os.path.join(self.dataRoot, self.sequenceName, "gt", "radar_odometry.csv")
        if self.hasGroundTruth and os.path.exists(gtPath):
            self.gtTraj = getGroundTruthTrajectory(gtPath)
            if initPose is None:
                initPose = self.gtTraj.getPoseAtTimes(initTimestamp)
        if initPose is None:
            initPose = np.zeros(3)
        initPose = np.asarray(initPose, dtype=np.float64)
        self.estTraj = Trajectory([initTimestamp], [initPose])
        do_md = self.paramFlags.get("correctMotionDistortion", True)

        MDS = MotionDistortionSolver(np.diag([4, 4]), np.diag([1, 1, (5 * np.pi / 180) ** 2]))
        prev_pose = convertPoseToTransform(initPose)
        prevImgPolar = getPolarImageFromImgPaths(imgPathArr, startSeqInd)
        prevImgCart = convertPolarImageToCartesian(prevImgPolar)
        blobCoord, _ = appendNewFeatures(prevImgCart, np.empty((0, 2)))
        metricCoord = (blobCoord - RADAR_CART_CENTER) * RANGE_RESOLUTION_CART_M
        zero_velocity = np.zeros((3,))
        old_kf = Keyframe(initPose, metricCoord, prevImgPolar, zero_velocity)
        self.map.addKeyframe(old_kf)
        possible_kf = Keyframe(initPose, metricCoord, prevImgPolar, zero_velocity)
        latestPose = initPose

        for seqInd in range(startSeqInd + 1, endSeqInd + 1):
            currImgPolar = getPolarImageFromImgPaths(imgPathArr, seqInd)
            currImgCart = convertPolarImageToCartesian(currImgPolar)
            good_old, good_new, rotAngleRad, corrStatus = tracker.track(prevImgCart, currImgCart, prevImgPolar, currImgPolar,
                                                                        blobCoord, seqInd)
            old_kf.pruneFeaturePoints(corrStatus)
            R, h = tracker.getTransform(good_old, good_new, pixel=False)
            centered_new = (good_new - RADAR_CART_CENTER) * RANGE_RESOLUTION_CART_M
            timestamp = radarImgPathToTimestamp(imgPathArr[seqInd])
            if do_md:
                p_w = old_kf.getPrunedFeaturesGlobalPosition()
                T_wj = prev_pose @ np.block([[R, h], [np.zeros((2,)), 1]])
                MDS.update_problem(prev_pose, p_w, centered_new, T_wj)
                sol = MDS.optimize_library()
                pose_vector, velocity = sol[3:], sol[:3]
                self.estTraj.appendAbsoluteTransform(timestamp, pose_vector)
            else:                                      # updateTrajectory (RawROAMSystem.py:301-317)
                self.estTraj.appendRelativeDeltas(timestamp, convertRandHtoDeltas(R, h))
                pose_vector, velocity = self.estTraj.poses[-1].copy(), np.zeros(3)
            latestPose = pose_vector
            possible_kf.updateInfo(latestPose, centered_new, currImgPolar, velocity)
            nFeatures = good_new.shape[0]
            retrack = (nFeatures <= N_FEATURES_BEFORE_RETRACK)
            newkf = retrack or self.map.isGoodKeyframe(possible_kf)
            if newkf:
                self.map.addKeyframe(possible_kf)
                old_kf = possible_kf
                if retrack:
                    good_new, _ = appendNewFeatures(currImgCart, good_new)
                    centered_new = (good_new - RADAR_CART_CENTER) * RANGE_RESOLUTION_CART_M
                    old_kf.updateInfo(latestPose, centered_new, currImgPolar, velocity)
                possible_kf = Keyframe(latestPose, centered_new, currImgPolar, velocity)
            self.frameLog.append(dict(seqInd=seqInd, n_tracked=len(blobCoord), n_inliers=nFeatures, new_keyframe=bool(newkf),
                                      retrack=bool(retrack), pose=np.array(latestPose, dtype=np.float64)))
            blobCoord = good_new.copy()
            prevImgCart = currImgCart
            prev_pose = convertPoseToTransform(latestPose)

    def rmse(self):
        """position RMSE of the estimate against the ground truth at the estimate's timestamps"""
        assert self.gtTraj is not None and self.estTraj is not None
        gt = self.gtTraj.getPoseAtTimes(self.estTraj.timestamps)
        return computePosesRMSE(gt, self.estTraj.poses)
