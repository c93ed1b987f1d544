"""Trajectory bookkeeping for the streaming driver: pose log, SE(2) integration of relative motions, ground-truth
loading and the position RMSE.  The public names follow the reference's trajectoryPlotting.py (Trajectory with its
append* methods and getPoseAtTimes, computePosesRMSE, getGroundTruthTrajectory - reference :11-122,183-213) so that
code written against it keeps working; the implementation is this project's own: poses live in a growable block
(amortised O(1) append), whole delta sequences are integrated in one vectorised pass (`integrate_deltas`), the
ground-truth file is read as a numeric table.  Plotting is out of scope."""
import numpy as np

GT_COLUMNS = dict(stamp=9, dx=2, dy=3, dyaw=7)      # radar_odometry.csv: destination_radar_timestamp, x, y, yaw


def _wrap(a):
    return (np.asarray(a) + np.pi) % (2 * np.pi) - np.pi


def integrate_deltas(deltas, pose0=(0.0, 0.0, 0.0), wrap=True):
    """Dead-reckon a whole sequence of body-frame motions (n, 3) [dx, dy, dth] from pose0 -> (n, 3) poses after each
    motion: heading = running sum of the turns (wrapped into [-pi, pi) when `wrap`), position = running sum of the
    motions rotated by the heading BEFORE each turn."""
    d = np.asarray(deltas, np.float64).reshape(-1, 3)
    head_after = pose0[2] + np.cumsum(d[:, 2])
    head_before = np.concatenate(([pose0[2]], head_after[:-1]))
    c, s = np.cos(head_before), np.sin(head_before)
    out = np.empty_like(d)
    out[:, 0] = pose0[0] + np.cumsum(d[:, 0] * c - d[:, 1] * s)
    out[:, 1] = pose0[1] + np.cumsum(d[:, 0] * s + d[:, 1] * c)
    out[:, 2] = _wrap(head_after) if wrap else head_after
    return out


class Trajectory:
    """time-stamped SE(2) poses [x, y, th]"""

    def __init__(self, timestamps, poses):
        t = np.atleast_1d(np.asarray(timestamps))
        p = np.asarray(poses, np.float64).reshape(-1, 3)
        assert len(t) == len(p) and len(p) >= 1
        self._n = len(p)
        cap = max(64, 2 * self._n)
        self._t = np.empty(cap, t.dtype)
        self._p = np.empty((cap, 3))
        self._t[:self._n], self._p[:self._n] = t, p
        self._T = None                       # left-composed transform of appendRelativeTransform (lazy)

    # ---- views
    @property
    def timestamps(self):
        return self._t[:self._n]

    @property
    def poses(self):
        return self._p[:self._n]

    def __len__(self):
        return self._n

    def _push(self, time, pose):
        if self._n == len(self._t):
            self._t = np.concatenate((self._t, np.empty_like(self._t)))
            self._p = np.concatenate((self._p, np.empty_like(self._p)))
        self._t[self._n], self._p[self._n] = time, pose
        self._n += 1

    # ---- the reference's append flavours
    def appendAbsoluteTransform(self, time, pose):
        self._push(time, np.asarray(pose, np.float64).reshape(3))

    def appendRelativeDeltas(self, time, d_xyth):
        """one body-frame motion on top of the latest pose (heading not wrapped, like the reference)"""
        self._push(time, integrate_deltas([d_xyth], self._p[self._n - 1], wrap=False)[0])

    def appendRelativeTransform(self, time, R, h):
        """left-compose [[R, h], [0, 1]] onto the running transform and log its pose"""
        if self._T is None:
            x, y, th = self._p[self._n - 1]
            self._T = np.array([[np.cos(th), -np.sin(th), x], [np.sin(th), np.cos(th), y], [0.0, 0.0, 1.0]])
        A = np.eye(3)
        A[:2, :2], A[:2, 2] = R, np.asarray(h, np.float64).reshape(2)
        self._T = A @ self._T
        self._push(time, (self._T[0, 2], self._T[1, 2], np.arctan2(self._T[1, 0], self._T[0, 0])))

    def extend_absolute(self, times, poses):
        for t, p in zip(times, np.asarray(poses, np.float64).reshape(-1, 3)):
            self._push(t, p)

    # ---- queries
    def getPoseAtTimes(self, times):
        """poses at arbitrary times: cubic spline per component through the samples, nearest sample when a spline cannot
        be built (fewer than four samples) - the behaviour of the reference's interp1d(kind='cubic') with its fallback"""
        scalar = np.isscalar(times)
        tq = np.atleast_1d(np.asarray(times, np.float64))
        t = self.timestamps.astype(np.float64)
        if self._n >= 4 and np.all(np.diff(t) > 0):
            from scipy.interpolate import make_interp_spline
            out = make_interp_spline(t, self.poses, k=3)(tq)
            out[(tq < t[0]) | (tq > t[-1])] = np.nan          # bounds_error=False: no extrapolation
        else:
            out = self.poses[np.abs(t[None, :] - tq[:, None]).argmin(axis=1)]
        return out[0] if scalar else out


def computePosesRMSE(gtPoses, estPoses):
    """root mean square of the planar position error (headings are not part of it)"""
    d = np.asarray(gtPoses, np.float64)[:, :2] - np.asarray(estPoses, np.float64)[:, :2]
    return float(np.sqrt(np.mean(np.einsum("ij,ij->i", d, d))))


def load_gt_deltas(gtPath):
    """radar_odometry.csv -> (timestamps int64 (n,), deltas (n, 3) [dx, dy, dyaw]); one header line"""
    c = GT_COLUMNS
    tab = np.loadtxt(gtPath, delimiter=",", skiprows=1, usecols=(c["stamp"], c["dx"], c["dy"], c["dyaw"]), dtype=np.float64, ndmin=2)
    stamps = np.loadtxt(gtPath, delimiter=",", skiprows=1, usecols=(c["stamp"],), dtype=np.int64, ndmin=1)
    return stamps, tab[:, 1:]


def getGroundTruthTrajectory(gtPath):
    """ground-truth Trajectory of an Oxford sequence: its relative motions dead-reckoned from the origin; `.gt_deltas`
    maps a destination radar timestamp to its [dx, dy, dyaw]"""
    stamps, deltas = load_gt_deltas(gtPath)
    traj = Trajectory(stamps, integrate_deltas(deltas))
    traj.gt_deltas = {int(s): d.tolist() for s, d in zip(stamps, deltas)}
    return traj
