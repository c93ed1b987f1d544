"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/roam_abi.h
declares; host-side logic (SE(2) helpers, dedupe, record decode, synthetic generator) works
without a GPU; compute entry points fail loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "roam_abi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(roam_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from radarslampy_amd import _ffi
    assert os.path.exists(_ffi.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"missing symbol {s}"
    assert set(syms) == set(_ffi.ABI_SYMBOLS), set(syms) ^ set(_ffi.ABI_SYMBOLS)
    _ffi.load_library()


def test_version_string():
    from radarslampy_amd import _ffi
    lib = _ffi.load_library()
    assert b"gfx950" in lib.roam_version()


def test_no_device_fails_loudly():
    """On a box without a GPU every compute path must raise - there is no CPU fallback."""
    from radarslampy_amd import _ffi
    lib = _ffi.load_library()
    h = ctypes.c_void_p()
    rc = lib.roam_create(0, ctypes.byref(h))
    if rc == _ffi.ROAM_OK:          # running on the GPU box: nothing to assert here
        lib.roam_destroy(h)
        pytest.skip("GPU present")
    assert rc == _ffi.ROAM_E_NODEVICE
    with pytest.raises(_ffi.RoamError):
        _ffi.Context(0)
    from radarslampy_amd.getPointCloud import getPointCloudPolarInd
    with pytest.raises(_ffi.RoamError):
        getPointCloudPolarInd(np.zeros((4, 8), np.float32))


def test_null_context_is_rejected():
    from radarslampy_amd import _ffi
    lib = _ffi.load_library()
    n = ctypes.c_int64(0)
    assert lib.roam_peaks_polar_f32(None, None, 1, 1, None, 0, ctypes.byref(n)) == _ffi.ROAM_E_ARG
    assert lib.roam_destroy(None) == _ffi.ROAM_E_ARG
    assert lib.roam_engine_step(None, None) == _ffi.ROAM_E_ARG


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "radarslampy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "liboracle" not in src, f


def test_se2_helpers_match_goldens(golden):
    from radarslampy_amd import utils
    g = golden("se2_utils")
    assert np.allclose(utils.convertPoseToTransform(g["poses"]), g["T"], atol=1e-15)
    assert np.allclose(utils.convertTransformToPose(g["T"]), g["poses_back"], atol=1e-15)
    assert np.allclose(utils.normalize_angles(g["ang"]), g["ang_norm"], atol=1e-15)
    assert np.allclose(np.stack([utils.invert_transform(T) for T in g["T"]]), g["Tinv"], atol=1e-12)
    assert np.array_equal(utils.homogenize(g["poses"][:, :2]), g["homog"])
    R = utils.getRotationMatrix(0.3)
    assert np.allclose(utils.convertRandHtoDeltas(R, np.array([[1.5], [-0.25]])), g["deltas"], atol=1e-15)


def test_record_decode_matches_golden(golden):
    from radarslampy_amd import parseData
    g = golden("record_format")
    rec = np.zeros((400, 3779), np.uint8)
    rec[:, :11] = g["meta_u8"]
    rec[:, 11:11 + 64] = g["payload_head_u8"]
    data, az, rres, ares, valid, ts = parseData.extractDataFromRadarImage(rec)
    assert data.shape == (400, 2025) and data.dtype == np.float32
    assert np.array_equal(data[:, :64], g["polar_head"]) and np.array_equal(az, g["azimuths"])
    assert np.array_equal(valid, g["valid"]) and np.array_equal(ts, g["timestamps"])
    assert rres == float(g["range_resolution"])


def test_dedupe_append():
    from radarslampy_amd.getFeatures import dedupe_append
    import oracle
    rng = np.random.default_rng(1)
    old = rng.integers(0, 50, (40, 2)).astype(np.float32)
    new = rng.integers(0, 50, (60, 2)).astype(np.float64)
    assert np.array_equal(dedupe_append(old, new), oracle.append_dedupe(old, new))


def test_synthetic_record_layout():
    from radarslampy_amd import synth, parseData
    recs, poses, feat = synth.make_sequence(3, 2)
    assert recs[0].shape == (400, 3779) and recs[0].dtype == np.uint8
    data, az, _, _, valid, ts = parseData.extractDataFromRadarImage(recs[1])
    assert valid.all() and np.all(np.diff(ts[:, 0]) > 0)
    assert abs(float(az[1, 0] - az[0, 0]) - 14 / 5600 * 2 * np.pi) < 1e-6
    assert 5 < data.mean() * 255 < 20 and feat.shape[0] > 100
    recs2, _, _ = synth.make_sequence(3, 2)
    assert np.array_equal(recs[1], recs2[1])        # seeded, reproducible


def test_u8_decode_identity():
    """The kernels decode power codes as (float)((double)k * (1.0/255.0)); this equals the
    reference's float32 division k/255 (parseData.py:40) for every one of the 256 codes."""
    k = np.arange(256)
    ref = k.astype(np.float32) / np.float32(255.)
    got = (k.astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
    assert np.array_equal(got, ref)
    # the warp's float32-only form (round 5, csrc/warp.hip code_to_f32): fma(k, head, k * tail) with head + tail = 1 / 255 - evaluated
    # here in exact rational arithmetic with one rounding per float32 operation
    from fractions import Fraction

    def rnd32(fr):
        f = np.float32(float(fr))
        cands = [np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))]
        return min(cands, key=lambda t: (abs(Fraction(float(t)) - fr), int(np.float32(t).view(np.uint32)) & 1))

    head, tail = float.fromhex("0x1.010102p-8"), float.fromhex("-0x1.fdfdfep-33")
    assert np.float32(head) == head and np.float32(tail) == tail
    for kk in range(256):
        low = rnd32(Fraction(kk) * Fraction(tail))
        assert rnd32(Fraction(kk) * Fraction(head) + Fraction(float(low))) == ref[kk], kk


def test_trajectory_and_gt_loader(tmp_path):
    from radarslampy_amd.trajectoryPlotting import Trajectory, getGroundTruthTrajectory, computePosesRMSE
    p = tmp_path / "radar_odometry.csv"
    rows = ["source_timestamp,destination_timestamp,x,y,z,roll,pitch,yaw,source_radar_timestamp,destination_radar_timestamp"]
    for i in range(6):
        rows.append(f"{i},{i},1.0,0.0,0,0,0,{0.1 if i % 2 else 0.0},{100 + i},{100 + i}")
    p.write_text("\n".join(rows) + "\n")
    gt = getGroundTruthTrajectory(str(p))
    assert gt.poses.shape == (6, 3) and abs(gt.poses[0, 0] - 1.0) < 1e-12 and gt.gt_deltas[101] == [1.0, 0.0, 0.1]
    tr = Trajectory([0], [np.zeros(3)])
    tr.appendRelativeDeltas(1, [1.0, 0.0, np.pi / 2])
    tr.appendRelativeDeltas(2, [1.0, 0.0, 0.0])
    assert np.allclose(tr.poses[-1], [1.0, 1.0, np.pi / 2])
    tr.appendAbsoluteTransform(3, np.array([5.0, 5.0, 0.0]))
    assert tr.poses.shape == (4, 3)
    assert abs(computePosesRMSE(np.zeros((3, 3)), np.array([[3, 4, 0.0]] * 3)) - 5.0) < 1e-12
    assert np.allclose(gt.getPoseAtTimes(102), gt.poses[2], atol=1e-9)


def test_native_blob_bookkeeping_matches_oracle_and_live_scipy():
    """roam_prune_blobs / roam_argsort_np122 are host code inside libroam_hip.so (no GPU): the product's restatement of
    scikit-image's pair order and NumPy 1.22's tie order against the oracle's independent C restatement, and against the
    skimage construct executed on the live scipy + CPython"""
    import math
    import numpy as np
    import oracle
    from radarslampy_amd import getFeatures as gf
    from scipy import spatial
    from test_oracle_reference_dump import _overlap
    rng = np.random.default_rng(8)
    for t in range(30):
        n = int(rng.integers(1, 900))
        c = rng.integers(20, 2000, size=(max(1, n // (3 if t % 2 else 12)), 2))
        b = np.column_stack([np.clip(c[rng.integers(0, len(c), n)] + rng.integers(-18, 19, size=(n, 2)), 0, 2023).astype(float),
                             rng.choice([5.005, 10.0], size=n)])
        got = gf._prune_blobs(b, 0.5)
        assert np.array_equal(got, oracle.prune_blobs(b, 0.5)), t
        if t < 8:
            want = b.copy()
            for i, j in np.array(list(spatial.cKDTree(want[:, :2]).query_pairs(2 * want[:, 2].max() * math.sqrt(2)))).reshape(-1, 2):
                b1, b2 = want[i], want[j]
                if _overlap(b1, b2) > 0.5:
                    if b1[2] > b2[2]:
                        b2[2] = 0
                    else:
                        b1[2] = 0
            assert np.array_equal(got, want[want[:, 2] > 0]), t
        v = rng.choice([5.005, 10.0], size=n) if t % 3 else rng.standard_normal(n)
        assert np.array_equal(gf.argsort_numpy122(v), oracle.argsort_numpy122(v)), t
    v = np.concatenate([np.arange(3000), np.arange(3000)[::-1]]).astype(float)       # organ pipe: deep recursion
    assert np.array_equal(gf.argsort_numpy122(v), oracle.argsort_numpy122(v))
    assert len(gf._prune_blobs(np.zeros((0, 3)), 0.5)) == 0 and len(gf.argsort_numpy122(np.zeros(0))) == 0
