"""Motion-distortion solver (reference motionDistortion.py:36-325, live 3-argument constructor
:70-78) on the MI355X (kabsch_mds.hip: MINPACK-lmdif restated, one workgroup per problem)."""
import numpy as np

from . import _ffi

RADAR_SCAN_FREQUENCY = 4
VERBOSE = False


class MotionDistortionSolver():
    def __init__(self, sigma_p, sigma_v, frequency=RADAR_SCAN_FREQUENCY):
        self.total_scan_time = 1 / frequency
        self.sigma_p = np.diag(sigma_p)
        self.sigma_v = np.diag(sigma_v)

    def update_problem(self, T_wj0, p_w, p_jt, T_wj, debug=False):
        assert (p_w.shape == p_jt.shape)
        self.T_wj0 = np.asarray(T_wj0, dtype=np.float64)
        self.T_wj0_inv = np.linalg.inv(self.T_wj0)
        self.p_w = np.asarray(p_w, dtype=np.float64)
        self.p_jt = np.asarray(p_jt, dtype=np.float64)
        self.T_wj_initial = np.asarray(T_wj, dtype=np.float64)
        self.debug = debug
        self.v_j_initial = self.infer_velocity(self.T_wj0_inv @ self.T_wj_initial)
        self.dT = MotionDistortionSolver.compute_time_deltas(self.total_scan_time, self.p_jt)
        sigma_vector = np.concatenate((np.tile(self.sigma_p, self.p_jt.shape[0]), self.sigma_v))
        self.info_vector = 1 / sigma_vector

    def infer_velocity(self, transform):
        return np.array([transform[0, 2], transform[1, 2], np.arctan2(transform[1, 0], transform[0, 0])]) / self.total_scan_time

    @staticmethod
    def compute_time_deltas(period, points):
        _, dT = _ffi.default_context().mds_undistort(np.zeros(3), np.asarray(points, dtype=np.float64), period)
        return dT

    @staticmethod
    def undistort(v_j, points, period=1 / RADAR_SCAN_FREQUENCY, times=None):
        """-> (N,3) homogeneous undistorted points (motionDistortion.py:126-153)."""
        if times is not None:
            raise NotImplementedError("explicit times are not used on the reference's live path")
        assert period > 0
        pts = np.asarray(points, dtype=np.float64)
        xy, _ = _ffi.default_context().mds_undistort(np.asarray(v_j, dtype=np.float64), pts, period)
        return np.column_stack((xy, np.ones(len(xy))))

    def optimize_library(self):
        """MINPACK-lmdif solve from x0 = [v_j_initial, pose of T_wj_initial] -> (6,) [v(3), pose(3)]."""
        sigma5 = np.concatenate((self.sigma_p, self.sigma_v)).astype(np.float64)
        sol, nfev, info, _, _ = _ffi.default_context().mds_solve(self.T_wj0, self.p_w, self.p_jt, self.T_wj_initial,
                                                                  sigma5, self.total_scan_time)
        self.nfev, self.status = nfev, info
        return sol
