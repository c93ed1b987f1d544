"""GPU (-m gpu): the engine's HOT kernels on the reference's REAL scans.

The one-lane real-scan tests (test_gpu_tiny_traj.py, test_gpu_retrack.py::test_device_retrack_on_the_reference_real_scans) hold
fewer than 200 detections per chunk and therefore take the small-chunk detection kernels.  Here the 11 real data/tiny payloads
(tests/golden/tiny_track.npz) are played by 2 x 256 lanes - 256 forward from the ground-truth start pose (the run the reference
printed into img/roam_mapping/tiny_traj, getFeatures.py:22-95 / parseData.py:100-135 / outlierRejection.py:63-75) and 256
backward - with retrack_slots = 512, so that EVERY detection chunk holds >= 200 detections and the engine's batch path runs on
real data: the one-sweep / fused detection kernels, rt_blobs_kernel<true>, the batched SSC, warp_gather_kernel + the pyramid
kernels at a few hundred lanes; and 512 problems per launch is where the engine starts the longest clique / blob-bookkeeping problems
first (cq_order_kernel).  Per lane and per step, against oracle.OdometryPipeline: features bit for bit, counts, poses at
1e-4 m / 1e-5 rad; frames 1-3 of the forward lanes against the numbers the reference itself printed; the warped + quantised image
and its three pyramid levels byte for byte."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PRINT = 1.1e-3
HALF = 256


def _detect(cart):
    return oracle.getFeatures(cart)[0]


def test_batch_kernels_on_the_reference_real_scans():
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    traj = np.load(os.path.join(HERE, "golden", "tiny_traj.npz"))
    pay = np.load(os.path.join(HERE, "golden", "tiny_track.npz"))["payload"]
    T, rows, clip = pay.shape
    B = 2 * HALF
    order = [list(range(T)), list(range(T - 1, -1, -1))]                   # forward / backward through the same 11 scans
    pose0 = [np.array(traj["gt_pose"][0], np.float64), np.zeros(3)]
    ctx = _ffi.Context(0)
    eng = Engine(B, T, ctx=ctx, rows=rows, stride=clip, payload_off=0, clip=clip, retrack_on_device=True, retrack_slots=B)
    for t in range(T):
        eng.upload_scan(t, np.ascontiguousarray(pay[t]))
    grp = np.arange(B) // HALF                                              # lane -> 0 forward, 1 backward
    eng.init_lanes_detect(0, [order[g][0] for g in grp], np.array([pose0[g] for g in grp]))
    pipes = []
    for g in range(2):
        rec0 = np.ascontiguousarray(pay[order[g][0]])
        cart0 = oracle.convertPolarImageToCartesian(rec0.astype(np.float32) / np.float32(255.))
        feat0 = oracle.append_dedupe(np.empty((0, 2)), _detect(cart0))
        assert 150 <= len(feat0) <= 260
        pipes.append(oracle.OdometryPipeline(rec0, feat0, pose0[g], detect=_detect, payload_off=0, clip=clip))
    for b in range(B):                                                       # 512 first detections in one chunk
        assert np.array_equal(eng.lane_features(b), pipes[grp[b]].blobCoord), b
    check_img = (0, 1, HALF - 1, HALF, B - 1)
    n_rt = [0, 0]
    for k in range(1, T):
        eng.step([order[g][k] for g in grp])
        res = eng.results()
        want = [pipes[g].step(np.ascontiguousarray(pay[order[g][k]])) for g in range(2)]
        # every chunk of this step is either empty or holds >= 256 detections (the batch detection path)
        for name in ("doh_integral", "doh_det_maxima"):
            m = eng.kernel_chunk_ms(name, 1)
            assert m.shape[0] == 1 and (m[0] > 0.0).any(), (k, name, m)
        for b in range(B):
            g, got, w = grp[b], res[b], want[grp[b]]
            tag = (k, b)
            assert (got["n_tracked"], got["n_good"], got["n_inliers"], got["n_peaks"]) == \
                   (w["n_tracked"], w["n_good"], w["n_inliers"], w["n_peaks"]), tag
            assert got["clique_proven"] and got["detect_overflow"] == 0, tag
            assert got["retrack"] == bool(w["retrack"]) and got["retracked_on_device"] == bool(w["retrack"]), tag
            assert np.abs(got["pose"][:2] - w["pose"][:2]).max() <= 1e-4 and abs(got["pose"][2] - w["pose"][2]) <= 1e-5, (tag, got["pose"], w["pose"])
            assert np.array_equal(eng.lane_features(b), pipes[g].blobCoord), tag
            if g == 0 and k <= 3:                                            # the reference's own prints, frames 1-3
                printed = np.array([got["pose"][0], got["pose"][1], np.rad2deg(got["pose"][2])])
                assert np.abs(printed - traj["roam_mapping_est_pose"][k - 1]).max() <= PRINT, (tag, printed)
        for g in range(2):
            n_rt[g] += bool(want[g]["retrack"])
        for b in check_img:                                                  # warp_gather_kernel + pyramid on real scans, byte for byte
            for lvl in range(4):
                assert np.array_equal(eng.lane_image(b, lvl), pipes[grp[b]].prevPyr[lvl]), (k, b, lvl)
        if k in (1, 5, T - 1):
            for b in check_img:
                assert np.array_equal(eng.lane_peaks(b), want[grp[b]]["peaks"]), (k, b)
    assert n_rt[0] == 5 and n_rt[1] >= 2, n_rt                               # forward: frames 1, 2, 4, 7, 9 (the reference's pictures)
    eng.close()
    ctx.close()
