"""radarslampy_amd — MI355X-native (gfx950, hand-written HIP behind a ctypes C-ABI) front end
for the per-scan hot path of Samleo8/RadarSLAMPy: polar peaks, polar->Cartesian warp,
pyramidal KLT, consistency-graph outlier rejection, 2-D Kabsch, motion-distortion LM.

The modules mirror the reference's module names so that `from radarslampy_amd.Tracker import
Tracker` (etc.) is a drop-in for the reference import.  There is no CPU fallback: without
libroam_hip.so and a gfx950 device every compute call raises `_ffi.RoamError`."""
__version__ = "0.1.0"
