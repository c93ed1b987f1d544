"""Fourier-Mellin rotation prior with the reference's names (reference FMT.py:10-90); the computation runs on the MI355X
(csrc/fmt.hip).  SURVEY §8f-f4."""
from . import _ffi
from .parseData import RANGE_RESOLUTION_CART_M

FMT_DOWNSAMPLE_FACTOR = 10      # FMT.py:10
FMT_RANGE_CLIP_M = 87.5         # FMT.py:11


def getRotationUsingFMT(srcPolarImg, targetPolarImg, downsampleFactor: int = FMT_DOWNSAMPLE_FACTOR, maxRangeClipM=FMT_RANGE_CLIP_M):
    """-> (angleRad with R(angleRad) @ src = target, scaling factor, response); polar (not log-polar) float32 images"""
    assert srcPolarImg.shape == targetPolarImg.shape, "Images need to have the same shape!"
    clip = int(maxRangeClipM / RANGE_RESOLUTION_CART_M) if maxRangeClipM > 0 else 0
    return _ffi.default_context().fmt_rotation(srcPolarImg, targetPolarImg, clip_px=clip, downsample=int(downsampleFactor))
