"""GPU (-m gpu): the fused detection kernel (csrc/retrack_fused.inc: integral image + Hessian determinants + 3 x 3 x 3 maxima in one
kernel, the float64 integral image never in HBM; getFeatures.py:39-51) against the engine's default two-kernel form
(rt_integral_kernel + rt_det_strip_kernel), through roam_engine_debug_detect: the integral image bit for bit, the candidate lists
(position, layer, determinant) of every detection bit for bit - on the reference's real data/tiny scans, on synthetic Oxford-size
scans, and on image sizes that do not fill the bands / blocks.  And end to end: an engine created with ROAM_FUSED_DETECT=1 detects
the same features as the oracle.  (The fused form is an opt-in variant: it is measured slower)"""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _compare(eng, P, s_slot):
    n1, rc1, v1, S1 = eng.debug_detect(True, P, s_slot=s_slot)          # fused first: the scratch image holds nothing of the other form
    n0, rc0, v0, S0 = eng.debug_detect(False, P, s_slot=s_slot)
    assert np.array_equal(S0, S1), int(np.count_nonzero(S0 != S1))
    assert np.array_equal(n0, n1) and n0.min() > 0 and n0.max() <= rc0.shape[1], (n0[:8], n1[:8])
    for i in range(P):
        assert np.array_equal(rc0[i, :n0[i]], rc1[i, :n0[i]]) and np.array_equal(v0[i, :n0[i]], v1[i, :n0[i]]), i
    return n0


def test_fused_equals_two_kernels_on_the_reference_real_scans():
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    pay = np.load(os.path.join(HERE, "golden", "tiny_track.npz"))["payload"]
    T, rows, clip = pay.shape
    P = 220                                                                   # 20 detections of each of the 11 real scans
    ctx = _ffi.Context(0)
    eng = Engine(P, T, ctx=ctx, rows=rows, stride=clip, payload_off=0, clip=clip, retrack_on_device=True, retrack_slots=P)
    for t in range(T):
        eng.upload_scan(t, np.ascontiguousarray(pay[t]))
    eng.init_lanes_detect(0, [b % T for b in range(P)], np.zeros((P, 3)))
    eng.step([(b + 1) % T for b in range(P)])
    n = _compare(eng, P, s_slot=3)
    assert len(set(n[:T].tolist())) > 3                                       # eleven different scans, not one
    eng.close()
    ctx.close()


@pytest.mark.parametrize("clip", [300, 497, 1000])
def test_fused_equals_two_kernels_on_image_sizes_that_do_not_fill_bands_and_blocks(clip):
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    rng = np.random.default_rng(clip)
    pay = (rng.random((400, clip)) * 12).astype(np.uint8)
    for _ in range(80):
        a, r = int(rng.integers(0, 400)), int(rng.integers(20, clip - 5))
        pay[max(0, a - 3):a + 3, max(0, r - 3):r + 3] = rng.integers(150, 255)
    P = 208
    ctx = _ffi.Context(0)
    eng = Engine(P, 2, ctx=ctx, rows=400, stride=clip, payload_off=0, clip=clip, retrack_on_device=True, retrack_slots=P)
    eng.upload_scan(0, pay)
    eng.upload_scan(1, np.ascontiguousarray(pay[::-1]))
    eng.init_lanes_detect(0, [b % 2 for b in range(P)], np.zeros((P, 3)))
    eng.step([(b + 1) % 2 for b in range(P)])
    _compare(eng, P, s_slot=1)
    eng.close()
    ctx.close()


def test_fused_equals_two_kernels_on_synthetic_oxford_records():
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, _ = synth.make_sequence(31, 3, n_movers=6, distortion=True)
    P = 512                                                                   # two workgroups per CU
    ctx = _ffi.Context(0)
    eng = Engine(P, 3, ctx=ctx, retrack_on_device=True, retrack_slots=P)
    for t in range(3):
        eng.upload_scan(t, recs[t])
    eng.init_lanes_detect(0, [b % 3 for b in range(P)], np.zeros((P, 3)))
    eng.step([(b + 1) % 3 for b in range(P)])
    _compare(eng, P, s_slot=2)
    eng.close()
    ctx.close()


def test_engine_with_the_fused_detector_matches_the_oracle(monkeypatch):
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    monkeypatch.setenv("ROAM_FUSED_DETECT", "1")
    recs, poses, _ = synth.make_sequence(21, 2, n_movers=8)
    P = 208
    ctx = _ffi.Context(0)
    eng = Engine(P, 2, ctx=ctx, retrack_on_device=True, retrack_slots=P)
    for t in range(2):
        eng.upload_scan(t, recs[t])
    eng.init_lanes_detect(0, [b % 2 for b in range(P)], np.tile(poses[0], (P, 1)))
    for b in (0, 1, 101, 207):
        cart = oracle.convertPolarImageToCartesian(recs[b % 2][:, 11:11 + 2025].astype(np.float32) / np.float32(255.))
        want = oracle.append_dedupe(np.empty((0, 2)), oracle.getFeatures(cart)[0])
        assert 150 <= len(want) <= 260 and np.array_equal(eng.lane_features(b), want), b
    eng.close()
    ctx.close()
