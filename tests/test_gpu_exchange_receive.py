"""GPU (-m gpu): the RECEIVE half of the multi-rank keyframe exchange (BASELINE config 5) on one GPU.

No N > 1 hardware is available to a test, and the CPU stub of tests/test_distributed_cpu.py shares no code with csrc/engine.hip, so the
kernel the exchange runs after its ncclAllGather (kfx_append_kernel: Map.addKeyframe in rank order, Mapping.py:176-180) is fed here with
receive buffers as an 8-rank all-gather would leave them - valid records, empty ones, records over the caps, several valid ones in one
gather, a remote map of exactly `world` slots that wraps - and the remote map is compared with a NumPy model of the ring."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORLD = 8


def make_record(lay, rng, rank, n, P, scan):
    from radarslampy_amd import _ffi
    rec = np.zeros(lay["rec_bytes"], np.uint8)
    hdr = _ffi.KeyframeHdr()
    pose, vel = rng.normal(size=3), rng.normal(size=3)
    hdr.pose[:], hdr.velocity[:] = pose, vel
    hdr.n_features, hdr.n_peaks, hdr.scan, hdr.lane = n, P, scan, rank
    rec[:C.sizeof(hdr)] = np.frombuffer(bytes(hdr), np.uint8)
    loc = rng.normal(size=(max(n, 0), 2)) if n <= _ffi.MAX_FEATURES else np.zeros((0, 2))
    pk = rng.integers(0, 3000, (P, 2), dtype=np.int32) if 0 <= P <= lay["max_peaks"] else np.zeros((0, 2), np.int32)
    if len(loc):
        rec[lay["locals_off"]:lay["locals_off"] + loc.size * 8] = np.frombuffer(loc.tobytes(), np.uint8)
    if len(pk):
        rec[lay["peaks_off"]:lay["peaks_off"] + pk.size * 4] = np.frombuffer(pk.tobytes(), np.uint8)
    return rec, dict(pose=pose, velocity=vel, locals=loc, peaks=pk, scan=scan, lane=rank, root=rank)


def test_append_kernel_on_fabricated_gathers():
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    rng = np.random.default_rng(8)
    ctx = _ffi.Context(0)
    eng = Engine(1, 2, ctx=ctx)
    with pytest.raises(_ffi.RoamError):
        eng.debug_keyframe_append(None, WORLD)                       # no remote map yet
    eng.remote_map_reserve(WORLD)                                    # exactly `world` slots: the smallest legal ring
    with pytest.raises(_ffi.RoamError):
        eng.debug_keyframe_append(None, WORLD + 1)                   # fewer slots than ranks would tear a gather
    lay = eng.debug_keyframe_append(None, WORLD)
    assert lay["rec_bytes"] == lay["peaks_off"] + 8 * lay["max_peaks"] and lay["locals_off"] == 64
    assert eng.remote_map_count() == (0, 0)
    K, PK = _ffi.MAX_FEATURES, lay["max_peaks"]
    model = []                                                       # every keyframe ever appended, in order
    # (n_features, n_peaks) per rank; None = the rank made no keyframe in this step
    gathers = [
        [(5, 7), None, (K + 1, 3), (4, PK + 1), (3, -2), (K, 11), (0, 0), None],        # valid, empty, over the caps (dropped), full, zero-size
        [None] * WORLD,                                                                  # a step without any keyframe
        [(17, 40), (1, 1), (200, 3000), (9, 0), (33, 5), (2, 2), (64, 64), (8, 1)],       # eight valid in one gather: the ring wraps
        [None, None, None, None, None, None, None, (12, 6)],                             # only the last rank
        [(7, PK), (6, 2), None, None, (K + 5, 1), None, (5, 5), None],
    ]
    for g, spec in enumerate(gathers):
        recv = np.zeros((WORLD, lay["rec_bytes"]), np.uint8)
        for r, s in enumerate(spec):
            n, P = (-1, 0) if s is None else s
            rec, want = make_record(lay, rng, r, n, P, scan=100 * g + r)
            recv[r] = rec
            if s is not None and 0 <= n <= K and 0 <= P <= PK:
                model.append(want)
        eng.debug_keyframe_append(recv, WORLD)
        received, resident = eng.remote_map_count()
        assert received == len(model) and resident == min(len(model), WORLD), (g, received, resident)
        tail = model[-resident:] if resident else []
        for i, want in enumerate(tail):                              # index 0 = the oldest keyframe still resident
            got = eng.remote_map_get(i)
            tag = (g, i)
            assert got["root"] == want["root"] and got["lane"] == want["lane"] and got["scan"] == want["scan"], tag
            assert np.array_equal(got["pose"], want["pose"]) and np.array_equal(got["velocity"], want["velocity"]), tag
            assert np.array_equal(got["prunedUndistortedLocals"], want["locals"]), tag
            assert np.array_equal(got["peaks"], want["peaks"]), tag
    assert len(model) == 3 + 8 + 1 + 3
    # a two-rank job on the same engine (fewer ranks than the buffer was sized for): still rank order
    recv = np.zeros((2, lay["rec_bytes"]), np.uint8)
    for r in range(2):
        recv[r], want = make_record(lay, rng, r, 10 + r, 4, scan=900 + r)
        model.append(want)
    eng.debug_keyframe_append(recv, 2)
    assert eng.remote_map_count() == (len(model), WORLD)
    assert [eng.remote_map_get(WORLD - 2 + r)["scan"] for r in range(2)] == [900, 901]
    eng.close()
    ctx.close()
