"""diagnostics on the GPU box: the fused detection kernel (retrack_fused.inc) against the two-kernel form on the same detections -
integral image of one slot bit for bit, candidate lists of every slot.  usage: python profiles/fused_check.py [small|full|real] [slots]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radarslampy_amd import _ffi, synth                      # noqa: E402
from radarslampy_amd.engine import Engine                    # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "small"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 208
ctx = _ffi.Context(0)
if mode == "small":
    clip = int(os.environ.get("CLIP", "300"))
    rng = np.random.default_rng(7)
    pay = (rng.random((400, clip)) * 40).astype(np.uint8)
    for _ in range(60):
        a, r = rng.integers(0, 400), rng.integers(20, clip - 5)
        pay[max(0, a - 3):a + 3, max(0, r - 3):r + 3] = rng.integers(150, 255)
    eng = Engine(P, 2, ctx=ctx, rows=400, stride=clip, payload_off=0, clip=clip, retrack_on_device=True, retrack_slots=P)
    eng.upload_scan(0, pay)
    eng.upload_scan(1, np.ascontiguousarray(pay[::-1]))
    eng.init_lanes_detect(0, [b % 2 for b in range(P)], np.zeros((P, 3)))
    eng.step([b % 2 for b in range(P)])
elif mode == "real":
    pay = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tiny_track.npz"))["payload"]
    T, rows, clip = pay.shape
    eng = Engine(P, T, ctx=ctx, rows=rows, stride=clip, payload_off=0, clip=clip, retrack_on_device=True, retrack_slots=P)
    for t in range(T):
        eng.upload_scan(t, np.ascontiguousarray(pay[t]))
    eng.init_lanes_detect(0, [b % T for b in range(P)], np.zeros((P, 3)))
    eng.step([(b + 1) % T for b in range(P)])
else:
    recs, poses, feat = synth.make_sequence(31, 3, n_movers=6, distortion=True)
    eng = Engine(P, 3, ctx=ctx, retrack_on_device=True, retrack_slots=P)
    for t in range(3):
        eng.upload_scan(t, recs[t])
    eng.init_lanes_detect(0, [b % 3 for b in range(P)], np.zeros((P, 3)))
    eng.step([(b + 1) % 3 for b in range(P)])
eng.synchronize()
print("engine up; features lane 0:", len(eng.lane_features(0)), flush=True)
s_slot = int(os.environ.get("SSLOT", "1"))
t1 = time.time(); n1, rc1, v1, S1 = eng.debug_detect(True, P, s_slot=s_slot); t2 = time.time()     # (fused first: S holds nothing of the other form yet)
t0 = time.time(); n0, rc0, v0, S0 = eng.debug_detect(False, P, s_slot=s_slot); t1, t2, t0 = time.time(), time.time() - t1 + t2 - t2, t0

W = S0.shape[0]
bad = np.argwhere(S0 != S1)
print("integral image W=%d: %d of %d elements differ" % (W, len(bad), S0.size))
if len(bad):
    print(" first differing (row, col):", bad[:8].tolist())
    rr, cc = bad[0]
    print(" S0, S1 there:", S0[rr, cc], S1[rr, cc], " rows with a difference:", np.unique(bad[:, 0])[:20], " cols:", np.unique(bad[:, 1])[:20])
    print(" count per band:", np.bincount((bad[:, 0] - 17).clip(0) // 62)[:40])
print("candidate counts equal:", np.array_equal(n0, n1), n0[:6], n1[:6])
nbad = 0
for i in range(P):
    k = min(n0[i], rc0.shape[1])
    if n0[i] != n1[i] or not np.array_equal(rc0[i, :k], rc1[i, :k]) or not np.array_equal(v0[i, :k], v1[i, :k]):
        nbad += 1
        if nbad <= 3:
            a = set(zip(rc0[i, :n0[i]].tolist(), v0[i, :n0[i]].tolist())); b = set(zip(rc1[i, :n1[i]].tolist(), v1[i, :n1[i]].tolist()))
            only0 = sorted(a - b)[:6]; only1 = sorted(b - a)[:6]
            print(" slot", i, "n", n0[i], n1[i], "only two-kernel:", [(x >> 16, (x >> 2) & 0x3fff, x & 3, v) for x, v in only0],
                  "only fused:", [(x >> 16, (x >> 2) & 0x3fff, x & 3, v) for x, v in only1])
print("slots whose candidate lists differ: %d of %d" % (nbad, P))
if os.environ.get("TIME"):
    for name in ["doh_integral", "doh_det_maxima", "doh_fused"] + ["doh_fused:%s" % m for m in os.environ.get("ABL", "").split(",") if m]:
        ms, by = eng.time_kernel(name, 5)
        print("%s: %.3f ms per %d detections" % (name, ms, P))
eng.close(); ctx.close()
