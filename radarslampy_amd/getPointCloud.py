"""Polar peak extraction (reference getPointCloud.py:11-54) on the MI355X (peaks.hip)."""
import numpy as np

from . import _ffi


def getPointCloudPolarInd(polarImage: np.ndarray, peakDistance: float = None, peakProminence: float = None) -> np.ndarray:
    """-> (K, 2) int64 rows [thetaInd, rInd], azimuth-major, range ascending."""
    if peakDistance is not None or peakProminence is not None:
        raise NotImplementedError("the reference never passes distance/prominence (Mapping.py:62)")
    return _ffi.default_context().peaks_polar_f32(polarImage).astype(np.int64)


def getPointCloudFromRecord(record_u8: np.ndarray, payload_off: int = 11, clip: int = 2025) -> np.ndarray:
    """Fused decode + peaks straight from the raw u8 record (no f32 polar image)."""
    return _ffi.default_context().peaks_record_u8(record_u8, payload_off, clip).astype(np.int64)
