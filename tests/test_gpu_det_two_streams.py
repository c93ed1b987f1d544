"""GPU (-m gpu): the two-stream form of the device-side detection (launch_retrack: the determinants of a chunk on a second stream beside
the next chunk's integral images, the images alternating between the two halves of the scratch) against the one-stream form - same
engine, every lane re-detecting in every step, several chunks per step so that both halves are reused.  Engines of >= 2 048 lanes take
the two-stream form by default (chunks of 1 024); here it is switched through ROAM_DET_SIDE on a smaller engine."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(chunk):
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    os.environ["ROAM_DET_SIDE"] = str(chunk)                        # read by roam_engine_create
    try:
        ctx = _ffi.Context(0)
        B, T = 600, 4
        seqs = [synth.make_sequence(300 + d, T, n_movers=20, distortion=True, scintillation=0.5) for d in range(2)]
        eng = Engine(B, 2 * T, ctx=ctx, retrack_on_device=True, retrack_slots=512)
        for d in range(2):
            for t in range(T):
                eng.upload_scan(d * T + t, seqs[d][0][t])
        for b in range(B):
            eng.init_lane(b, (b % 2) * T, seqs[b % 2][2][:50 + (b % 4) * 25], seqs[b % 2][1][0])
        assert eng.detect_chunk() == (chunk if chunk else 512)
        out = []
        for t in range(1, T):
            eng.set_retrack(2)
            eng.step([(b % 2) * T + t for b in range(B)])
            res = eng.results()
            h = zlib.crc32(np.array([r["pose"] for r in res]).tobytes())
            h = zlib.crc32(np.array([[r["n_tracked"], r["n_good"], r["n_inliers"], r["n_after_retrack"]] for r in res]).tobytes(), h)
            for b in (0, 1, 127, 128, 255, 256, 383, 511, 512, 599):
                h = zlib.crc32(eng.lane_features(b).tobytes(), h)
            out.append(h)
            ms = eng.kernel_chunk_ms("doh_integral", 1)
            assert ms.shape[1] == -(-B // eng.detect_chunk()) and (ms[0] > 0).all()
        eng.close()
        ctx.close()
        return out
    finally:
        os.environ.pop("ROAM_DET_SIDE", None)


def test_two_stream_detection_equals_one_stream():
    one = run(0)
    assert run(128) == one          # five chunks per step: both halves of the scratch are reused twice
    assert run(256) == one          # three chunks
