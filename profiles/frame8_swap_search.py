#!/usr/bin/env python3
"""BUILD CONTAINER / CPU only.  Frames 8-10 of data/tiny (round-5 verdict, item 4): under the pictures' keyframe policy and the swapped
near-tie of frame 2, frames 1-7 print the reference's numbers and frame 8 is 2.6 units of the last digit off.  The correspondence
markers (profiles/frame6_markers.py) say the reference's frame-8 inlier set is the FIRST clique in networkx order - ours - with ONE
feature exchanged ((1676, 1014) for one near (1704, 989)) that no tied clique contains: the feature sets differ after the re-detection
of frame 7.  This script looks for the cause the way frame 2's was found: exchange one near-tie (responses < 1e-3 apart) of the
response-ordered DoH maxima of frame 7 (or of frame 4, the retrack before it) and see whether frames 8 and 9 then print the reference's poses."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle                                                   # noqa: E402
from test_oracle_tiny_traj import _deltas, _printed             # noqa: E402

W = 2024


def main():
    traj = np.load(os.path.join(ROOT, "tests", "golden", "tiny_traj.npz"))
    pay = np.load(os.path.join(ROOT, "tests", "golden", "tiny_track.npz"))["payload"]
    frame, swaps, ties = [0], {2: 521}, {}

    def detect(cart):
        f = frame[0]
        sig = np.linspace(0.01, 10, 3)
        rcs, val, _ = oracle.doh_maxima(np.asarray(cart, np.float64), sig, .0005)
        idx = np.argsort(-val, kind="stable")
        v = val[idx]
        ties[f] = [i for i in range(len(v) - 1) if abs(v[i] - v[i + 1]) / v[i] < 1e-3]
        if f in swaps:
            i = swaps[f]
            idx[[i, i + 1]] = idx[[i + 1, i]]
        bl = rcs[idx].astype(np.float64)
        bl[:, 2] = sig[rcs[idx][:, 2]]
        sel = oracle.adaptiveNMS((W, W), oracle.prune_blobs(bl, 0.5))
        return np.fliplr(sel[:, :2])

    cart0 = oracle.convertPolarImageToCartesian(pay[0].astype(np.float32) / np.float32(255.))
    pipe = oracle.OdometryPipeline(np.ascontiguousarray(pay[0]), oracle.append_dedupe(np.empty((0, 2)), detect(cart0)), traj["gt_pose"][0],
                                   detect=detect, payload_off=0, clip=pay.shape[2], keyframe_trans_m=0.0)
    est = [traj["gt_pose"][0]]
    snaps = {}
    for t in range(1, 10):
        if t in (4, 7):
            snaps[t] = (copy.deepcopy(pipe), list(est))
        frame[0] = t
        out = pipe.step(np.ascontiguousarray(pay[t]))
        est.append(out["pose"].copy())
        d = np.abs(_printed(out["pose"]) - traj["roam_mapping_est_pose"][t - 1])
        print("baseline frame %d: |pose - print| = %s  retrack %s" % (t, np.round(d, 4), bool(out["retrack"])), flush=True)
    print("near-ties: frame 4: %d, frame 7: %d" % (len(ties.get(4, [])), len(ties.get(7, []))), flush=True)
    hits = []
    for f in (7, 4):
        for i in ties.get(f, []):
            swaps.clear(); swaps.update({2: 521, f: i})
            p, e = copy.deepcopy(snaps[f][0]), list(snaps[f][1])
            p.detect = detect
            worst = {}
            for t in range(f, 10):
                frame[0] = t
                out = p.step(np.ascontiguousarray(pay[t]))
                e.append(out["pose"].copy())
                worst[t] = float(np.abs(_printed(out["pose"]) - traj["roam_mapping_est_pose"][t - 1]).max())
            ok = all(worst[t] <= 1.1e-3 for t in range(f, 9))
            print("swap frame %d index %3d: max |pose - print| frames %s%s" % (f, i, {t: round(w, 4) for t, w in worst.items()}, "   <-- frames up to 8 print the reference" if ok else ""), flush=True)
            if ok:
                hits.append((f, i, worst))
    print("hits:", hits)


if __name__ == "__main__":
    main()
