#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (reads the reference's pictures).  Decodes the correspondence markers the reference drew into
img/roam_mapping/tiny_traj/00NN.jpg - yellow '.' = good_old, the inlier set of the pair in the previous image's pixels
(getTransformKLT.py:56-62 through Tracker.plot) - and compares them with the inlier sets of oracle.OdometryPipeline (frame-2 swap,
tests/test_oracle_tiny_traj.py): per frame, the yellow blobs no oracle point explains and the oracle points without a marker; for
frame 6 every one of the four tied maximum cliques.  Result (round 5): frames 2, 3, 5, 6, 7 - nothing unexplained, nothing unmarked,
and at frame 6 only the FIRST clique in networkx order fits (each of the others leaves one 21-pixel blob unexplained and one point
unmarked): the reference's frame-6 inlier set IS ours.  Its different pose comes from the keyframe policy of the run that made the
pictures.  Display transform: fitted on frames 2, 3, 5 (x_disp = a x + bx, y_disp = a y + by)."""
import os
import sys

import numpy as np
from PIL import Image
from scipy import ndimage as ndi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle                                                   # noqa: E402
from test_oracle_tiny_traj import _detect_with_swap             # noqa: E402

PICS = "/root/reference/img/roam_mapping/tiny_traj/%04d.jpg"


def yellow_mask(k):
    im = np.asarray(Image.open(PICS % k).convert("RGB")).astype(float)
    m = (im[..., 0] > 190) & (im[..., 1] > 190) & (im[..., 2] < 140)
    m[:, 520:] = False                                          # the trajectory panel
    m[:100, 305:] = False                                       # the legend
    return m


def render(pts, a, bx, by, rad, shape):
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    out = np.zeros(shape, bool)
    for x, y in pts:
        out |= (xx - (a * x + bx)) ** 2 + (yy - (a * y + by)) ** 2 <= rad * rad
    return out


def iou(m, pts, a, bx, by, rad=2.3):
    r = render(pts, a, bx, by, rad, m.shape)
    return (r & m).sum() / max(1, (r | m).sum())


def compare(m, pts, a, bx, by, rad=2.4):
    un = m & ~render(pts, a, bx, by, rad + 0.8, m.shape)
    lab, n = ndi.label(un)
    blobs = []
    for i in range(1, n + 1):
        if (lab == i).sum() >= 4:
            c = ndi.center_of_mass(lab == i)
            blobs.append((int((lab == i).sum()), round((c[1] - bx) / a), round((c[0] - by) / a)))
    unmarked = []
    for j, (x, y) in enumerate(pts):
        cx, cy = a * x + bx, a * y + by
        yy, xx = np.mgrid[int(cy) - 3:int(cy) + 5, int(cx) - 3:int(cx) + 5]
        ok = (yy >= 0) & (yy < m.shape[0]) & (xx >= 0) & (xx < m.shape[1])          # (a point at the picture's edge)
        yy, xx = np.clip(yy, 0, m.shape[0] - 1), np.clip(xx, 0, m.shape[1] - 1)
        d = ((xx - cx) ** 2 + (yy - cy) ** 2 <= rad * rad) & ok
        if d.sum() == 0:
            continue
        if (m[yy, xx] & d).sum() / d.sum() < 0.45:
            unmarked.append((j, round(float(x), 1), round(float(y), 1)))
    return blobs, unmarked


def main():
    traj = np.load(os.path.join(ROOT, "tests", "golden", "tiny_traj.npz"))
    pay = np.load(os.path.join(ROOT, "tests", "golden", "tiny_track.npz"))["payload"]
    frame, rec = [0], {}
    detect = _detect_with_swap({2: 521}, frame)
    real = oracle.rejectOutliers

    def rej(prev, new):
        out = real(prev, new)
        rec[frame[0]] = dict(prev=prev.copy(), mask=out[2].copy(), masks=oracle.max_cliques_nx_all(oracle.consistency_graph(prev, new)))
        return out
    oracle.rejectOutliers = rej
    cart0 = oracle.convertPolarImageToCartesian(pay[0].astype(np.float32) / np.float32(255.))
    pipe = oracle.OdometryPipeline(np.ascontiguousarray(pay[0]), oracle.append_dedupe(np.empty((0, 2)), detect(cart0)), traj["gt_pose"][0],
                                   detect=detect, payload_off=0, clip=pay.shape[2], keyframe_trans_m=0.0)
    for t in range(1, 11):
        frame[0] = t
        pipe.step(np.ascontiguousarray(pay[t]))
    oracle.rejectOutliers = real
    masks = {k: yellow_mask(k) for k in (2, 3, 5, 6, 7, 8, 9, 10)}
    best = None
    for a in np.linspace(0.1992, 0.2004, 13):
        for bx in np.linspace(85.5, 86.75, 11):
            for by in np.linspace(36.75, 38.25, 13):
                s = np.mean([iou(masks[k], rec[k]["prev"][rec[k]["mask"]], a, bx, by) for k in (2, 3, 5)])
                if best is None or s > best[0]:
                    best = (s, a, bx, by)
    _, a, bx, by = best
    print("display transform: a %.5f bx %.3f by %.3f (mean IoU %.3f on frames 2, 3, 5)" % (a, bx, by, best[0]))
    for k in (2, 3, 5, 6, 7, 8, 9, 10):
        pts = rec[k]["prev"][rec[k]["mask"]]
        print("frame %d: %d inliers, IoU %.3f, unexplained yellow blobs / unmarked points:" % (k, len(pts), iou(masks[k], pts, a, bx, by)), *compare(masks[k], pts, a, bx, by))
    for f in (6, 8, 9, 10):                                      # (round 6: frames 8-10 added)
        for i, m in enumerate(rec[f]["masks"]):
            print("frame %d, tied clique %d of %d (leaves out %s):" % (f, i, len(rec[f]["masks"]), np.where(~m)[0].tolist()),
                  *compare(masks[f], rec[f]["prev"][m], a, bx, by))


if __name__ == "__main__":
    main()
