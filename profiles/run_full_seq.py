#!/usr/bin/env python3
"""BASELINE configs 3 / 4 end to end: the 8 866 frames of full_seq_1 (synthetic scans rendered along the sequence's real
ground-truth motions, tests/golden/full_seq_1_gt_deltas.npz) streamed through the engine by the RawROAMSystem driver's
core (stream_records: pinned staging ring, asynchronous uploads, one roam_engine_step per pair, device-side retracks, poses
from the result ring).  Prints the position RMSE against the rendered ground truth and the streaming rate.
usage: python profiles/run_full_seq.py [frames] [md 0|1] [procs]"""
import json, multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from radarslampy_amd import synth
from radarslampy_amd.RawROAMSystem import stream_records
from radarslampy_amd.trajectoryPlotting import computePosesRMSE

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8866
    md = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else min(96, os.cpu_count() or 8)
    deltas = np.load(os.path.join(ROOT, "tests", "golden", "full_seq_1_gt_deltas.npz"))["deltas"][:n - 1]
    gt = synth.poses_from_deltas(deltas)
    jobs = synth.stream_jobs(synth.StreamWorld(2, mover_fraction=0.10), gt, distortion=md, scintillation=0.4)
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        recs = pool.imap(synth._render_job, jobs, chunksize=8)          # rendered ahead of the consumer, in order
        t1 = time.perf_counter()
        poses, log = stream_records(recs, len(gt), gt[0], {"rejectOutliers": True, "correctMotionDistortion": md})
        t2 = time.perf_counter()
    err = np.hypot(*(poses[:, :2] - gt[1:, :2]).T)
    out = dict(frames=len(gt), motion_distortion=md, distance_m=round(float(np.hypot(deltas[:, 0], deltas[:, 1]).sum()), 1),
               position_rmse_m=round(computePosesRMSE(gt[1:], poses), 3), final_position_error_m=round(float(err[-1]), 3),
               max_position_error_m=round(float(err.max()), 3),
               heading_rmse_rad=round(float(np.sqrt(np.mean(((poses[:, 2] - gt[1:, 2] + np.pi) % (2 * np.pi) - np.pi) ** 2))), 5),
               retracks=int(sum(e["retrack"] for e in log)), keyframes=int(sum(e["new_keyframe"] for e in log)),
               mean_tracked=round(float(np.mean([e["n_tracked"] for e in log])), 1), mean_inliers=round(float(np.mean([e["n_inliers"] for e in log])), 1),
               wall_s_incl_rendering=round(t2 - t1, 1), pairs_per_s_one_lane_incl_rendering_and_h2d=round((len(gt) - 1) / (t2 - t1), 1), render_procs=procs)
    print(json.dumps(out))
