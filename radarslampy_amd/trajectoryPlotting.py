"""Trajectory container, ground-truth integration and RMSE with the reference's names
(reference trajectoryPlotting.py:11-122,183-213).  §8f-f3 "next" row; plotting is out of scope."""
import csv

import numpy as np

from .utils import convertPoseToTransform, convertTransformToPose, normalize_angles


class Trajectory():
    def __init__(self, timestamps, poses):
        self.timestamps = np.array(timestamps)
        self.poses = np.array(poses, dtype=np.float64)
        self.pose_transform = convertPoseToTransform(self.poses[-1])

    def appendRelativeDeltas(self, time, d_xyth):
        dx, dy, dth = d_xyth
        self.timestamps = np.append(self.timestamps, time)
        x, y, th = self.poses[-1]
        x += dx * np.cos(th) - dy * np.sin(th)
        y += dx * np.sin(th) + dy * np.cos(th)
        th += dth
        self.poses = np.vstack((self.poses, [x, y, th]))

    def appendRelativeTransform(self, time, R, h):
        self.timestamps = np.append(self.timestamps, time)
        A = np.block([[R, h], [np.zeros((1, 2)), 1]])
        self.pose_transform = A @ self.pose_transform
        self.poses = np.vstack((self.poses, convertTransformToPose(self.pose_transform)))

    def appendAbsoluteTransform(self, time, pose):
        self.timestamps = np.append(self.timestamps, time)
        self.poses = np.vstack((self.poses, pose))

    def getPoseAtTimes(self, times):
        """cubic interpolation with nearest-sample fallback (trajectoryPlotting.py:73-97)"""
        import scipy.interpolate
        scalar = np.isscalar(times)
        tq = np.atleast_1d(times)
        try:
            f = [scipy.interpolate.interp1d(self.timestamps, self.poses[:, i], kind='cubic', bounds_error=False) for i in range(3)]
            poses = np.vstack([fi(tq) for fi in f]).T
        except Exception:
            poses = np.zeros((len(tq), 3))
            for i, t in enumerate(tq):
                poses[i, :] = self.poses[np.argmin(np.abs(self.timestamps - t))]
        return poses[0, :] if scalar else poses


def computePosesRMSE(gtPoses, estPoses):
    euclidean_err = np.linalg.norm(gtPoses[:, :-1] - estPoses[:, :-1], axis=-1)
    return np.sqrt(np.mean(euclidean_err ** 2))


def getGroundTruthTrajectory(gtPath):
    """radar_odometry.csv -> Trajectory (destination_radar_timestamp col 9, x col 2, y col 3, yaw col 7)"""
    with open(gtPath) as gt_file:
        gt_reader = csv.reader(gt_file)
        _ = next(gt_file)
        ts, poses, d_xyths = [], [], {}
        x, y, th = 0, 0, 0
        for row in gt_reader:
            timestamp = int(row[9])
            dx, dy, dth = float(row[2]), float(row[3]), float(row[7])
            x += dx * np.cos(th) + dy * -np.sin(th)
            y += dx * np.sin(th) + dy * np.cos(th)
            th = normalize_angles(th + dth)
            ts.append(timestamp)
            poses.append([x, y, th])
            d_xyths[timestamp] = [dx, dy, dth]
    traj = Trajectory(np.array(ts), np.array(poses))
    traj.gt_deltas = d_xyths
    return traj
