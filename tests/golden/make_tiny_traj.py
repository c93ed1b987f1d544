#!/usr/bin/env python3
"""Fixture generator for tests/golden/tiny_traj.npz.  RUNS ONLY IN THE BUILD CONTAINER (reads /root/reference).

The reference repository holds the END-TO-END OUTPUT of its own runs on `data/tiny` as pictures: every frame's
trajectory plot carries the numbers RawROAMSystem.plotTraj printed into it (RawROAMSystem.py:434-447, utils.f_arr:
three decimals, angles in degrees) and the running RMSE in the title (trajectoryPlotting.py:176):

  img/roam_mapping/tiny_traj/00NN.jpg   HEAD's pipeline (plot labels "Previous Features" / "Map Points" =
                                        getTransformKLT.py:64, Mapping.py:203): motion-distortion poses - THE pin for
                                        the whole §8 path, ties of the maximum clique included
  img/roam/tiny_traj/00NN.jpg           an earlier revision (legend "Image 0 Features"): dead reckoning from the Kabsch fit,
                                        different feature bookkeeping - kept as data, HEAD does not reproduce it
  img/dead_reckoning/tiny_traj/00NN.jpg the legacy driver (getTransformKLT.py:384-541 at an earlier revision:
                                        calculateTransformDxDth + a RANDOM RANSAC rejection), six decimals, radians -
                                        kept as data, not reproducible by construction (frame 1 is a different plot)

The numbers below were transcribed by eye from the pictures.  This script does not trust the transcription: for the two
sets that share HEAD's layout it renders the four-line text block with matplotlib (the reference used matplotlib too:
same DejaVu Sans, same 'small' size, same rasteriser), locates it in the picture by normalised cross-correlation
(block NCC 0.97-0.99 against JPEG noise and the grid lines that cross the text) and then, for EVERY digit of every
number, renders the nine alternatives and requires the transcribed digit to correlate best inside the pixels that
differ (the RMSE of the title, a different font, is transcribed only).  The GT deltas and timestamps in the pictures are cross-checked against data/tiny/gt/radar_odometry.csv and the
PNG names.  Stored: the arrays, and the text-block crops of the pictures as evidence (uint8 luminance).
"""
import csv
import glob
import os

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

# frame: (EST pose x, y, th_deg), (GT deltas), (EST deltas), RMSE
ROAM_MAPPING = {
    1: ((151.478, 7.945, -0.593), (2.286, -0.011, -0.582), (2.098, 0.171, -0.840), 0.19),
    2: ((153.638, 7.906, -1.050), (2.078, -0.022, -0.623), (2.161, -0.017, -0.458), 0.19),
    3: ((155.920, 7.910, -1.506), (1.930, -0.019, -0.443), (2.281, 0.046, -0.456), 0.24),
    4: ((157.923, 7.920, -1.881), (1.879, -0.012, -0.353), (2.002, 0.063, -0.375), 0.30),
    5: ((159.911, 7.871, -2.150), (1.934, -0.000, -0.261), (1.988, 0.015, -0.270), 0.35),
    6: ((161.739, 7.780, -2.289), (1.862, -0.002, -0.118), (1.831, -0.021, -0.139), 0.37),
    7: ((163.477, 7.653, -2.475), (1.786, 0.005, 0.067), (1.741, -0.058, -0.186), 0.38),
    8: ((165.211, 7.552, -2.661), (1.674, 0.008, 0.500), (1.737, -0.026, -0.186), 0.38),
    9: ((167.047, 7.320, -2.363), (1.577, 0.006, 0.519), (1.845, -0.146, 0.299), 0.42),
    10: ((168.428, 7.202, -1.908), (1.543, 0.008, 0.174), (1.384, -0.060, 0.455), 0.43),
}
ROAM = {
    1: ((150.740, 8.645, -0.358), (2.286, -0.011, -0.582), (1.364, 0.874, -0.605), 0.90),
    2: ((152.046, 9.380, -0.896), (2.078, -0.022, -0.623), (1.301, 0.743, -0.539), 1.55),
    3: ((153.318, 10.165, -1.464), (1.930, -0.019, -0.443), (1.259, 0.805, -0.568), 2.18),
    4: ((154.793, 10.589, -1.793), (1.879, -0.012, -0.353), (1.463, 0.462, -0.330), 2.66),
    5: ((156.497, 10.742, -1.979), (1.934, -0.000, -0.261), (1.699, 0.206, -0.185), 3.01),
    6: ((158.036, 11.037, -2.219), (1.862, -0.002, -0.118), (1.528, 0.348, -0.241), 3.34),
    7: ((159.433, 11.381, -2.495), (1.786, 0.005, 0.067), (1.382, 0.398, -0.276), 3.66),
    8: ((160.852, 11.613, -2.689), (1.674, 0.008, 0.500), (1.408, 0.294, -0.194), 3.96),
    9: ((162.907, 10.955, -2.340), (1.577, 0.006, 0.519), (2.084, -0.561, 0.349), 4.07),
    10: ((165.031, 10.241, -1.946), (1.543, 0.008, 0.174), (2.151, -0.627, 0.394), 4.08),
}
# legacy driver: (Est pose x, y, th_rad), (GT deltas dx, dy, dth_deg), (Est deltas dx, dy, dth_rad), RMSE; frames 2..10
DEAD_RECKONING = {
    2: ((153.484186, 7.777705, -0.000172), (2.078366, -0.022247, -0.623149), (2.025994, 0.000000, -0.002282), 0.19),
    3: ((155.500468, 7.777359, -0.002122), (1.930374, -0.019123, -0.443011), (2.016282, 0.000000, -0.001950), 0.20),
    4: ((157.407956, 7.773312, -0.003441), (1.878835, -0.012445, -0.352770), (1.907493, 0.000000, -0.001319), 0.20),
    5: ((159.270327, 7.766904, -0.004564), (1.934158, -0.000435, -0.261211), (1.862382, 0.000000, -0.001124), 0.22),
    6: ((161.074279, 7.758670, -0.005472), (1.861726, -0.001693, -0.117915), (1.803970, 0.000000, -0.000908), 0.25),
    7: ((162.789562, 7.749283, -0.006694), (1.785589, 0.004586, 0.066921), (1.715309, 0.000000, -0.001222), 0.29),
    8: ((164.461446, 7.738091, -0.007176), (1.674029, 0.008494, 0.500479), (1.671922, 0.000000, -0.000482), 0.32),
    9: ((166.113754, 7.726234, -0.006756), (1.576616, 0.006244, 0.518985), (1.652350, 0.000000, 0.000420), 0.33),
    10: ((167.712400, 7.715433, -0.005562), (1.543221, 0.008295, 0.174408), (1.598683, 0.000000, 0.001195), 0.35),
}


def f_arr(xs):
    s = [f"{x:.3f}" for x in xs]
    s[-1] += "\N{DEGREE SIGN}"
    return "[" + ",".join(s) + "]"


def block_text(ts, row):
    pose, gtd, estd, _ = row
    return f"Timestamp: {ts}\nEST Pose: {f_arr(pose)}\nGT Deltas: {f_arr(gtd)}\nEST Deltas: {f_arr(estd)}"


class Renderer:
    def __init__(self):
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        self.fig = plt.figure(figsize=(4, 1), dpi=100)

    def __call__(self, text):
        self.fig.clear()
        self.fig.text(10 / 400, 80 / 100, text, ha="left", va="top", fontsize="small")
        self.fig.canvas.draw()
        return np.asarray(self.fig.canvas.buffer_rgba())[..., :3].mean(-1).astype(np.float32)


def ncc(a, b):
    a = a - a.mean()
    b = b - b.mean()
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum() + 1e-9))


def locate(img, tmpl):
    """(NCC, y, x) of the best match of tmpl (the tight text box) in img: normalised cross-correlation at every offset"""
    from scipy.signal import fftconvolve
    t0 = tmpl - tmpl.mean()
    n = tmpl.size
    ones = np.ones_like(tmpl)
    num = fftconvolve(img, t0[::-1, ::-1], mode="valid")
    s1 = fftconvolve(img, ones, mode="valid")
    s2 = fftconvolve(img * img, ones, mode="valid")
    var = np.maximum(s2 - s1 * s1 / n, 1e-6)
    score = num / np.sqrt(var * (t0 * t0).sum())
    y, x = np.unravel_index(int(np.argmax(score)), score.shape)
    return float(score[y, x]), int(y), int(x)


def verify(render, img, text):
    """-> (block NCC, crop, digits checked); raises if any other digit fits a position better than the transcribed one"""
    full = render(text)
    ys, xs = np.nonzero(full < 200)
    y0, y1, x0, x1 = ys.min() - 1, ys.max() + 2, xs.min() - 1, xs.max() + 2
    tmpl = full[y0:y1, x0:x1]
    c, oy, ox = locate(img, tmpl)
    assert c > 0.90, c
    crop = img[oy:oy + tmpl.shape[0], ox:ox + tmpl.shape[1]]
    n = 0
    for i, ch in enumerate(text):
        if not ch.isdigit():
            continue
        for alt in "0123456789":
            if alt == ch:
                continue
            other = render(text[:i] + alt + text[i + 1:])[y0:y1, x0:x1]
            dy, dx = np.nonzero(np.abs(other - tmpl) > 8)
            if len(dy) == 0:
                continue
            a0, a1, b0, b1 = max(0, dy.min() - 1), dy.max() + 2, max(0, dx.min() - 1), dx.max() + 2
            good, bad = ncc(tmpl[a0:a1, b0:b1], crop[a0:a1, b0:b1]), ncc(other[a0:a1, b0:b1], crop[a0:a1, b0:b1])
            assert good > bad, (text, i, ch, alt, good, bad)
        n += 1
    return c, crop.astype(np.uint8), n


def ground_truth():
    rows = list(csv.reader(open(os.path.join(REF, "data", "tiny", "gt", "radar_odometry.csv"))))[1:]
    return {int(r[9]): (float(r[2]), float(r[3]), float(np.rad2deg(float(r[7])))) for r in rows}


def ground_truth_poses(ts):
    """trajectoryPlotting.getGroundTruthTrajectory (:183-213) + Trajectory.getPoseAtTimes (:72-101, cubic interp1d) at the scan
    timestamps: row 0 = the start pose RawROAMSystem.run takes (RawROAMSystem.py:123-126), the rest = what the RMSE is taken against"""
    from scipy.interpolate import interp1d
    rows = list(csv.reader(open(os.path.join(REF, "data", "tiny", "gt", "radar_odometry.csv"))))[1:]
    x = y = th = 0.0
    T, P = [], []
    for r in rows:
        dx, dy, dth = float(r[2]), float(r[3]), float(r[7])
        x += dx * np.cos(th) + dy * -np.sin(th)
        y += dx * np.sin(th) + dy * np.cos(th)
        th = (th + dth + np.pi) % (2 * np.pi) - np.pi
        T.append(int(r[9])); P.append([x, y, th])
    T, P = np.array(T), np.array(P)
    return np.column_stack([interp1d(T, P[:, k], kind="cubic", bounds_error=False)(np.array(ts)) for k in range(3)])


def main():
    from PIL import Image
    paths = sorted(glob.glob(os.path.join(REF, "data", "tiny", "radar", "*.png")))
    ts = [int(os.path.basename(p)[:-4]) for p in paths]
    assert len(ts) == 11
    gt = ground_truth()
    out = dict(timestamps=np.array(ts, np.int64), gt_pose=ground_truth_poses(ts))
    render = Renderer()
    for name, table in (("roam_mapping", ROAM_MAPPING), ("roam", ROAM)):
        pose, gtd, estd, rmse, nccs = [], [], [], [], []
        for f in range(1, 11):
            row = table[f]
            img = np.array(Image.open(os.path.join(REF, "img", name, "tiny_traj", f"{f:04d}.jpg")).convert("L")).astype(np.float32)
            c, crop, nd = verify(render, img, block_text(ts[f], row))
            print(name, f, "block NCC %.3f" % c, nd, "digits verified")
            for a, b in zip(row[1], gt[ts[f]]):                       # what the picture says about the ground truth = the csv
                assert abs(a - b) <= 5.01e-4, (name, f, row[1], gt[ts[f]])
            out[f"{name}_crop_{f:02d}"] = crop
            pose.append(row[0]); gtd.append(row[1]); estd.append(row[2]); rmse.append(row[3]); nccs.append(c)
        out[f"{name}_est_pose"] = np.array(pose)                       # x [m], y [m], theta [deg], 3 decimals
        out[f"{name}_gt_deltas"] = np.array(gtd)
        out[f"{name}_est_deltas"] = np.array(estd)
        out[f"{name}_rmse"] = np.array(rmse)                           # 2 decimals (title)
        out[f"{name}_block_ncc"] = np.array(nccs)
    fr = sorted(DEAD_RECKONING)
    for f in fr:
        for a, b in zip(DEAD_RECKONING[f][1], gt[ts[f]]):
            assert abs(a - b) <= 5.01e-7 * max(1, 1), (f, DEAD_RECKONING[f][1], gt[ts[f]])
        im = Image.open(os.path.join(REF, "img", "dead_reckoning", "tiny_traj", f"{f:04d}.jpg")).convert("L")
        out[f"dead_reckoning_crop_{f:02d}"] = np.array(im.crop((110, 12, 530, 118)))
    out["dead_reckoning_frames"] = np.array(fr)
    out["dead_reckoning_est_pose"] = np.array([DEAD_RECKONING[f][0] for f in fr])       # x, y [m], theta [RAD], 6 decimals
    out["dead_reckoning_gt_deltas"] = np.array([DEAD_RECKONING[f][1] for f in fr])      # dx, dy [m], dtheta [deg]
    out["dead_reckoning_est_deltas"] = np.array([DEAD_RECKONING[f][2] for f in fr])     # dx, dy [m], dtheta [RAD]
    out["dead_reckoning_rmse"] = np.array([DEAD_RECKONING[f][3] for f in fr])
    path = os.path.join(OUT, "tiny_traj.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e3, "kB")


if __name__ == "__main__":
    main()
