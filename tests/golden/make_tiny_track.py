#!/usr/bin/env python3
"""Fixture generator for tests/golden/tiny_track.npz.  RUNS ONLY IN THE BUILD CONTAINER (reads /root/reference).

The reference repository holds OUTPUTS of its own cv2 / scikit-image / SciPy / NumPy stack on its
11 real `data/tiny` scans.  They pin the three stages whose libraries are absent here (cv2.warpPolar,
cv2.calcOpticalFlowPyrLK, skimage.feature.blob_doh) and the order-dependent bookkeeping around them:

  img/dead_reckoning/tiny_10.npz   blobCoord (257, 2) f32: the feature set the legacy driver
                                   (reference getTransformKLT.py:384-541, an earlier revision of it) saved after
                                   tracking frames 0 -> 10: blob_doh detections of frames 0, 3 and 9 (no ANMS at
                                   that revision) pushed through 10 / 7 / 1 pyramidal-LK steps.
  img/blob/tiny/00NN.jpg           written by getFeatures.py:121-186: the uint8 Cartesian image with a red circle
                                   (thickness 1, radius int(sigma)) on EVERY blob_doh blob and a green one
                                   (thickness 3) on every blob adaptiveNMS kept, for each of the 11 frames.

Only arrays are stored (no reference source text):
  payload        (11, 400, 2025) u8   clipped power payload of the 11 records (columns 11..2035 of the PNGs)
  blobCoord_ref  (257, 2) f32          tiny_10.npz["blobCoord"]
  ovl_idx_NN / ovl_green_NN / ovl_red_NN   sparse overlay of frame NN decoded from the JPEG:
                                        flat pixel index (int32) and u8 "greenness" G - max(R, B) and "redness"
                                        R - max(G, B) (clipped at 0) wherever either is >= 24
  lum_NN (for NN in 00, 10), lum_box    512 x 512 crop of the JPEG luminance (R+G+B)/3 rounded to u8
"""
import glob
import os

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
CLIP = 2025
LUM_BOX = (756, 1268, 756, 1268)         # y0, y1, x0, x1 (central crop)


def main():
    paths = sorted(glob.glob(os.path.join(REF, "data", "tiny", "radar", "*.png")))
    assert len(paths) == 11
    recs = np.array([np.array(Image.open(p)) for p in paths])
    assert recs.shape == (11, 400, 3779) and recs.dtype == np.uint8
    out = dict(payload=np.ascontiguousarray(recs[:, :, 11:11 + CLIP]),
               blobCoord_ref=np.load(os.path.join(REF, "img", "dead_reckoning", "tiny_10.npz"))["blobCoord"],
               lum_box=np.array(LUM_BOX, np.int32))
    for f in range(11):
        im = np.array(Image.open(os.path.join(REF, "img", "blob", "tiny", f"{f:04d}.jpg"))).astype(np.int32)
        assert im.shape == (2024, 2024, 3)
        R, G, B = im[..., 0], im[..., 1], im[..., 2]
        green = np.clip(G - np.maximum(R, B), 0, 255)
        red = np.clip(R - np.maximum(G, B), 0, 255)
        idx = np.flatnonzero((green >= 24) | (red >= 24)).astype(np.int32)
        out[f"ovl_idx_{f:02d}"] = idx
        out[f"ovl_green_{f:02d}"] = green.ravel()[idx].astype(np.uint8)
        out[f"ovl_red_{f:02d}"] = red.ravel()[idx].astype(np.uint8)
        if f in (0, 10):
            y0, y1, x0, x1 = LUM_BOX
            out[f"lum_{f:02d}"] = np.rint((R + G + B)[y0:y1, x0:x1] / 3.0).astype(np.uint8)
    path = os.path.join(OUT, "tiny_track.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
