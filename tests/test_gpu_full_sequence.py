"""GPU (-m gpu): BASELINE configs 3 and 4 at FULL length - all 8 866 frames (9.02 km) of the reference's full_seq_1 ground-truth
trajectory (tests/golden/full_seq_1_gt_deltas.npz; scans rendered on the fly by a pool of host processes from an unbounded
reflector world with 10 % movers and scintillation, with intra-scan distortion when motion distortion is on) streamed through
one lane by the RawROAMSystem driver's core: pinned staging ring, asynchronous uploads, one roam_engine_step per pair,
device-side retracks, poses from the result ring (reference RawROAMSystem.py:162-298, updateTrajectory :301-317).
Slow but run: ~30 s per mode on a box with >= 64 cores (the rendering is the cost); on small hosts the sequence is cut."""
import json
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("md", [False, True])
def test_full_seq_1_streaming(md):
    from radarslampy_amd import synth
    from radarslampy_amd.RawROAMSystem import stream_records
    from radarslampy_amd.trajectoryPlotting import computePosesRMSE
    cores = os.cpu_count() or 4
    n = 8866 if cores >= 32 else 1200
    deltas = np.load(os.path.join(HERE, "golden", "full_seq_1_gt_deltas.npz"))["deltas"][:n - 1]
    gt = synth.poses_from_deltas(deltas)
    jobs = synth.stream_jobs(synth.StreamWorld(2, mover_fraction=0.10), gt, distortion=md, scintillation=0.4)
    with mp.get_context("spawn").Pool(min(96, max(2, cores - 2))) as pool:
        recs = pool.imap(synth._render_job, jobs, chunksize=8)              # rendered ahead of the consumer, in order
        poses, log = stream_records(recs, len(gt), gt[0], {"rejectOutliers": True, "correctMotionDistortion": md})
    dist = float(np.hypot(deltas[:, 0], deltas[:, 1]).sum())
    rmse = computePosesRMSE(gt[1:], poses)
    err = np.hypot(*(poses[:, :2] - gt[1:, :2]).T)
    hd = float(np.sqrt(np.mean(((poses[:, 2] - gt[1:, 2] + np.pi) % (2 * np.pi) - np.pi) ** 2)))
    n_rt, n_kf = int(sum(e["retrack"] for e in log)), int(sum(e["new_keyframe"] for e in log))
    print(f"full_seq_1 md={md}: {len(gt)} frames, {dist:.1f} m, position RMSE {rmse:.3f} m, final error {err[-1]:.3f} m, heading RMSE {hd:.5f} rad, "
          f"retracks {n_rt}, keyframes {n_kf}")
    assert len(poses) == n - 1 and np.isfinite(poses).all()
    assert all(e["n_tracked"] > 0 for e in log[1:])                          # the lane never ran dry: every retrack refilled it
    if n == 8866:
        # dead-reckoned over 9 km: the drift stays below 1.5 % of the distance driven with the motion-distortion solve (the paper
        # reports 41.8 m on the real data) and below 3 % with plain Kabsch dead reckoning
        assert rmse < (0.015 if md else 0.03) * dist and hd < 0.2, (rmse, hd)
        # the run is deterministic (seeded rendering, exact kernels): THE committed figures (tests/golden/full_seq_1_figures.json), so
        # that "this kernel change was exact" is a test - a determinism check, not a parity reference.  Round 4 moved them once, on purpose: the clique tie-break is now the
        # reference's (rounds 2-3, lexicographic tie-break: 56.295 m / 178.9 m)
        want = json.load(open(os.path.join(HERE, "golden", "full_seq_1_figures.json")))["md_on" if md else "md_off"]
        assert abs(rmse - want["position_rmse_m"]) <= 1e-3 and abs(hd - want["heading_rmse_rad"]) <= 1e-5, (rmse, hd, want)
        assert (n_rt, n_kf) == (want["retracks"], want["keyframes"]), (n_rt, n_kf, want)
