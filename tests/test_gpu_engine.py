"""GPU (-m gpu): the batched device-resident engine against the oracle's CPU restatement of
the RawROAMSystem.run loop body, lane by lane, over several consecutive scan pairs."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

POS_TOL = 1e-4   # m
ANG_TOL = 1e-5   # rad


@pytest.fixture(scope="module")
def sequences():
    from radarslampy_amd import synth
    return [synth.make_sequence(seed, 4, n_movers=(12 if seed == 2 else 0), distortion=(seed == 3)) for seed in (1, 2, 3)]


@pytest.mark.parametrize("md,reject", [(True, True), (False, True), (True, False)])
def test_engine_matches_oracle_pipeline(sequences, md, reject):
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    ctx = _ffi.Context(0)
    B, T = len(sequences), 4
    eng = Engine(B, B * T, ctx=ctx, reject_outliers=reject, motion_distortion=md)
    pipes = []
    for b, (recs, poses, feat) in enumerate(sequences):
        for t in range(T):
            eng.upload_scan(b * T + t, recs[t])
        eng.init_lane(b, b * T, feat, poses[0])
        pipes.append(oracle.OdometryPipeline(recs[0], feat, poses[0], reject_outliers=reject, motion_distortion=md))
    for t in range(1, T):
        eng.step([b * T + t for b in range(B)])
        res = eng.results()
        for b, (recs, poses, feat) in enumerate(sequences):
            want = pipes[b].step(recs[t])
            got = res[b]
            tag = (md, reject, t, b)
            assert got["n_tracked"] == want["n_tracked"], tag
            assert got["n_good"] == want["n_good"], tag
            assert got["n_inliers"] == want["n_inliers"], tag
            assert got["n_peaks"] == want["n_peaks"], tag
            assert got["clique_proven"], tag
            assert np.array_equal(eng.lane_features(b), pipes[b].blobCoord), tag
            if md and reject:
                for lvl in range(4):      # map+gather warp and the pyramid are bit-exact
                    assert np.array_equal(eng.lane_image(b, lvl), pipes[b].prevPyr[lvl]), (tag, lvl)
            assert np.array_equal(eng.lane_peaks(b), want["peaks"]), tag
            assert np.abs(got["h"] - want["h"]).max() <= POS_TOL, tag
            assert abs(np.arctan2(got["R"][1, 0], got["R"][0, 0]) - np.arctan2(want["R"][1, 0], want["R"][0, 0])) <= ANG_TOL, tag
            assert np.abs(got["pose"][:2] - want["pose"][:2]).max() <= POS_TOL, (tag, got["pose"], want["pose"])
            assert abs(got["pose"][2] - want["pose"][2]) <= ANG_TOL, (tag, got["pose"], want["pose"])
            assert got["new_keyframe"] == bool(want["new_keyframe"]), tag
            assert got["retrack"] == bool(want["retrack"]), tag
    st = eng.stage_times()
    assert set(st) >= {"ingest_peaks", "warp_quantise", "pyramid", "klt", "max_clique", "mds_lm"}
    eng.close()
    ctx.close()


def test_engine_lm_forms_mixed_in_one_batch():
    """the motion-distortion solve has two kernel forms (one wavefront per problem up to 254 points, a workgroup above: kabsch_mds.hip);
    in the engine the first lists the problems it leaves to the second - lanes of both kinds in one batch, over several steps (the two
    lists alternate), against the oracle pipeline lane by lane"""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, feat = synth.make_sequence(11, 4, n_static=520, n_movers=0)
    sizes = [len(feat), 120, min(len(feat), 330), 60, min(len(feat), 280), 200]
    assert len(feat) >= 300
    ctx = _ffi.Context(0)
    B, T = len(sizes), 4
    eng = Engine(B, T, ctx=ctx, reject_outliers=True, motion_distortion=True)
    for t in range(T):
        eng.upload_scan(t, recs[t])
    pipes = []
    for b, n in enumerate(sizes):
        eng.init_lane(b, 0, feat[:n], poses[0])
        pipes.append(oracle.OdometryPipeline(recs[0], feat[:n], poses[0], reject_outliers=True, motion_distortion=True))
    seen = set()
    for t in range(1, T):
        eng.step([t] * B)
        res = eng.results()
        for b in range(B):
            want, got = pipes[b].step(recs[t]), res[b]
            tag = (t, b, sizes[b])
            assert got["n_inliers"] == want["n_inliers"] and got["n_good"] == want["n_good"], tag
            seen.add(got["n_inliers"] + 2 > 256)
            assert np.abs(got["pose"][:2] - want["pose"][:2]).max() <= POS_TOL, (tag, got["pose"], want["pose"])
            assert abs(got["pose"][2] - want["pose"][2]) <= ANG_TOL, (tag, got["pose"], want["pose"])
            assert got["lm_info"] in (1, 2, 3, 4), (tag, got["lm_info"])
    assert seen == {True, False}, "both kernel forms must have been exercised"
    eng.close()
    ctx.close()


def test_engine_without_stage_events_gives_the_same_results():
    """roam_engine_set_stage_events(0) (what the single-sequence driver does): the step's timestamp events are not recorded - same
    results record for record, stage_times / kernel_avg refuse with a state error instead of reading stale events"""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, _ = synth.make_sequence(31, 4, n_movers=8)
    out = []
    for ev in (True, False):
        ctx = _ffi.Context(0)
        eng = Engine(2, 4, ctx=ctx, retrack_on_device=True, stage_events=ev)
        for t in range(4):
            eng.upload_scan(t, recs[t])
        for b in range(2):
            eng.init_lane_detect(b, 0, poses[0])
        rows = []
        for t in range(1, 4):
            eng.step([t, t])
            rows.append([(r["n_tracked"], r["n_good"], r["n_inliers"], tuple(r["pose"])) for r in eng.results()])
        if ev:
            assert "klt" in eng.stage_times()
        else:
            with pytest.raises(_ffi.RoamError):
                eng.stage_times()
            with pytest.raises(_ffi.RoamError):
                eng.kernel_avg("warp_quantise", 1)
        out.append(rows)
        eng.close()
        ctx.close()
    assert out[0] == out[1]


def test_engine_argument_errors():
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    ctx = _ffi.Context(0)
    eng = Engine(1, 2, ctx=ctx)
    with pytest.raises(_ffi.RoamError):
        eng.step([5])                       # scan index outside the pool
    with pytest.raises(_ffi.RoamError):
        eng.init_lane(3, 0, np.zeros((4, 2), np.float32), np.zeros(3))
    eng.close()
    ctx.close()


def test_engine_retrack_matches_oracle(sequences):
    """Few initial features -> the retrack branch (RawROAMSystem.py:250-271): DoH + SSC-ANMS on the
    resident scan, dedupe-append, keyframe refresh; then tracking continues from the new set."""
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    recs, poses, feat = sequences[0]
    T = len(recs)
    ctx = _ffi.Context(0)
    eng = Engine(1, T, ctx=ctx)
    for t in range(T):
        eng.upload_scan(t, recs[t])
    feat0 = feat[:64]
    eng.init_lane(0, 0, feat0, poses[0])
    pipe = oracle.OdometryPipeline(recs[0], feat0, poses[0], detect=lambda cart: oracle.getFeatures(cart)[0])
    saw_retrack = False
    for t in range(1, T):
        eng.step([t])
        got = eng.results()[0]
        want = pipe.step(recs[t])
        assert got["retrack"] == bool(want["retrack"]) and got["n_inliers"] == want["n_inliers"], t
        if got["retrack"]:
            saw_retrack = True
            eng.retrack_lane(0, t)
        assert np.array_equal(eng.lane_features(0), pipe.blobCoord), t
        assert np.abs(got["pose"][:2] - want["pose"][:2]).max() <= POS_TOL and abs(got["pose"][2] - want["pose"][2]) <= ANG_TOL, t
    assert saw_retrack and len(pipe.blobCoord) > 150
    eng.close()
    ctx.close()


def test_engine_degenerate_lanes(sequences):
    """Lanes with 0, 1 and 3 features next to a healthy lane: nothing hangs, the starved lanes keep
    their pose and raise the retrack flag, the healthy lane is unaffected."""
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    recs, poses, feat = sequences[0]
    ctx = _ffi.Context(0)
    eng = Engine(4, 2, ctx=ctx)
    eng.upload_scan(0, recs[0])
    eng.upload_scan(1, recs[1])
    sets = [feat[:0], feat[:1], feat[:3], feat]
    for b, f in enumerate(sets):
        eng.init_lane(b, 0, f, poses[0])
    eng.step([1, 1, 1, 1])
    res = eng.results()
    ref = oracle.OdometryPipeline(recs[0], feat, poses[0])
    want = ref.step(recs[1])
    assert res[3]["n_inliers"] == want["n_inliers"]
    assert np.abs(res[3]["pose"][:2] - want["pose"][:2]).max() <= POS_TOL
    for b in (0, 1):
        assert res[b]["n_inliers"] <= 1 and res[b]["retrack"]
        assert np.allclose(res[b]["pose"], poses[0])
    assert res[2]["retrack"] and res[2]["n_inliers"] <= 3 and np.all(np.isfinite(res[2]["pose"]))
    assert res[0]["n_peaks"] == res[3]["n_peaks"] == want["n_peaks"]
    eng.close()
    ctx.close()


def test_async_pinned_upload_matches_sync_upload(sequences):
    """f2: records streamed from pinned host memory on the copy stream (double-buffered, fenced) give the
    same per-step results as records uploaded synchronously beforehand."""
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    recs, poses, feat = sequences[0]
    T = len(recs)
    ctx = _ffi.Context(0)
    ref = Engine(2, T, ctx=ctx)
    for t in range(T):
        ref.upload_scan(t, recs[t])
    for b in range(2):
        ref.init_lane(b, 0, feat, poses[0])
    want = []
    for t in range(1, T):
        ref.step([t, t])
        want.append(ref.results())
    ref.close()
    eng = Engine(2, 4, ctx=ctx)                         # two halves of 2 slots
    pinned = ctx.host_alloc((T, 400, 3779))
    for t in range(T):
        pinned[t] = recs[t]
    eng.upload_scans_async(0, pinned[0], n=2, stride=0)
    eng.synchronize()
    for b in range(2):
        eng.init_lane(b, b, feat, poses[0])
    eng.upload_scans_async(2, pinned[1], n=2, stride=0)     # scans of step 0 -> half 1
    for i, t in enumerate(range(1, T)):
        half = (i + 1) % 2
        eng.fence()
        eng.step(np.array([0, 1], np.int32) + 2 * half)
        if t + 1 < T:
            eng.upload_scans_async(2 * (i % 2), pinned[t + 1], n=2, stride=0)
        got = eng.results()
        for b in range(2):
            assert np.array_equal(got[b]["pose"], want[i][b]["pose"]), (t, b)
            assert got[b]["n_inliers"] == want[i][b]["n_inliers"] and got[b]["n_peaks"] == want[i][b]["n_peaks"]
    eng.synchronize()
    ctx.host_free(pinned)
    eng.close()
    ctx.close()


def test_engine_lane_batches_are_independent(sequences):
    """71 lanes (two full 32-scan warp groups + a ragged third, odd peak row groups) fed from three distinct
    sequences in an interleaved order: every lane must reproduce - bit for bit - the result, features, peak list
    and pyramid of the first lane that runs the same sequence, at every step."""
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    ctx = _ffi.Context(0)
    S, T, B = len(sequences), 3, 71
    eng = Engine(B, S * T, ctx=ctx)
    for s, (recs, poses, feat) in enumerate(sequences):
        for t in range(T):
            eng.upload_scan(s * T + t, recs[t])
    seq_of = [(5 * b + b // 7) % S for b in range(B)]
    for b in range(B):
        recs, poses, feat = sequences[seq_of[b]]
        eng.init_lane(b, seq_of[b] * T, feat, poses[0])
    first = {s: seq_of.index(s) for s in range(S)}
    for t in range(1, T):
        eng.step([seq_of[b] * T + t for b in range(B)])
        res = eng.results()
        for b in range(B):
            r = first[seq_of[b]]
            if b == r:
                continue
            for key in ("n_tracked", "n_good", "n_inliers", "n_peaks", "lm_nfev"):
                if key in res[r]:
                    assert res[b][key] == res[r][key], (t, b, key)
            assert np.array_equal(res[b]["pose"], res[r]["pose"]), (t, b)
            assert np.array_equal(eng.lane_features(b), eng.lane_features(r)), (t, b)
        for b in (1, 31, 32, 33, 63, 64, 70):
            r = first[seq_of[b]]
            assert np.array_equal(eng.lane_peaks(b), eng.lane_peaks(r)), (t, b)
            for lvl in range(4):
                assert np.array_equal(eng.lane_image(b, lvl), eng.lane_image(r, lvl)), (t, b, lvl)


def test_engine_keyframe_map_matches_oracle(sequences):
    """8f-f1: every keyframe the tracker replaces (pose criterion inside the step, or retrack through the host)
    must appear in the device-resident map with the state the oracle's Keyframe object had when it was replaced
    (pose, creation velocity, pruned undistorted locals); the last entry is the live keyframe."""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    # 0.8-1.6 m per scan: the 2 m keyframe criterion fires every second or third frame
    recs, poses, feat = synth.make_sequence(11, 8, n_movers=0, distortion=True)
    T = len(recs)
    ctx = _ffi.Context(0)
    eng = Engine(2, T, ctx=ctx)
    eng.map_reserve(8)
    for t in range(T):
        eng.upload_scan(t, recs[t])
    feat0 = feat[:90]
    pipes = []
    for b in range(2):
        eng.init_lane(b, 0, feat0, poses[0])
        pipes.append(oracle.OdometryPipeline(recs[0], feat0, poses[0], detect=lambda cart: oracle.getFeatures(cart)[0]))
    assert eng.map_count(0) == 1 and eng.map_keyframe(0, 0)["scan"] == 0
    frozen = [[], []]

    def same(kf, okf, tag):
        assert np.abs(kf["pose"][:2] - okf.pose[:2]).max() <= POS_TOL and abs(kf["pose"][2] - okf.pose[2]) <= ANG_TOL, tag
        assert np.abs(kf["velocity"] - np.asarray(okf.velocity)).max() <= 1e-4, tag
        assert kf["prunedUndistortedLocals"].shape == okf.prunedUndistortedLocals.shape, tag
        assert np.abs(kf["prunedUndistortedLocals"] - okf.prunedUndistortedLocals).max() <= 1e-4, tag

    n_new = 0
    for t in range(1, T):
        before = [p.old_kf for p in pipes]
        eng.step([t, t])
        res = eng.results()
        for b in range(2):
            want = pipes[b].step(recs[t])
            assert res[b]["retrack"] == bool(want["retrack"]) and res[b]["new_keyframe"] == bool(want["new_keyframe"]), (t, b)
            if want["new_keyframe"]:
                frozen[b].append(before[b])
                n_new += 1
            if res[b]["retrack"]:
                eng.retrack_lane(b, t)
            assert eng.map_count(b) == len(frozen[b]) + 1, (t, b)
            same(eng.map_keyframe(b, len(frozen[b])), pipes[b].old_kf, (t, b, "live"))
    assert n_new >= 2
    for b in range(2):
        kfs = eng.map_keyframes(b)
        assert len(kfs) == len(frozen[b]) + 1
        for i, okf in enumerate(frozen[b]):
            same(kfs[i], okf, (b, i))
        assert [k["scan"] for k in kfs] == sorted(k["scan"] for k in kfs)
    eng.close()
    ctx.close()


def test_engine_pipelined_steps_match_synchronised_steps(sequences):
    """The front end of step N+1 (peaks, warp, pyramid) may overlap the back end of step N when steps are enqueued
    back to back.  Six steps without any host synchronisation in between must leave exactly the state that the same
    six steps leave when the host reads the results after every step (poses, features, peaks, all pyramid levels)."""
    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine
    recs, poses, feat = synth.make_sequence(21, 7, n_movers=6, distortion=True)
    T, B = len(recs), 40
    ctx = _ffi.Context(0)

    def run(sync_each_step):
        eng = Engine(B, T, ctx=ctx)
        for t in range(T):
            eng.upload_scan(t, recs[t])
        for b in range(B):
            eng.init_lane(b, 0, feat[:200 + 3 * b], poses[0])
        for t in range(1, T):
            eng.step([t] * B)
            if sync_each_step:
                eng.results()
        res = eng.results()
        out = dict(pose=np.array([r["pose"] for r in res]), vel=np.array([r["velocity"] for r in res]),
                   counts=np.array([[r["n_tracked"], r["n_good"], r["n_inliers"], r["n_peaks"], r["lm_nfev"]] for r in res]),
                   feats=[eng.lane_features(b) for b in (0, 7, 39)], peaks=[eng.lane_peaks(b) for b in (0, 7, 39)],
                   imgs=[eng.lane_image(b, lvl) for b in (0, 39) for lvl in range(4)])
        eng.close()
        return out

    a, b = run(True), run(False)
    assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["vel"], b["vel"]) and np.array_equal(a["counts"], b["counts"])
    for k in ("feats", "peaks", "imgs"):
        for x, y in zip(a[k], b[k]):
            assert np.array_equal(x, y), k
    ctx.close()


def test_rccl_keyframe_broadcast_world1(sequences):
    """8e: the native RCCL path (roam_comm_* / roam_bcast_keyframe) on one GPU: communicator of size 1, all-reduce,
    barrier, and the keyframe payload packed on the device, broadcast and read back == the engine's own accessors"""
    import tempfile
    from radarslampy_amd import _ffi
    from radarslampy_amd import distributed as D
    from radarslampy_amd.engine import Engine
    ctx = _ffi.Context(0)
    B, T = len(sequences), 4
    eng = Engine(B, B * T, ctx=ctx)
    for b, (recs, poses, feat) in enumerate(sequences):
        for t in range(T):
            eng.upload_scan(b * T + t, recs[t])
        eng.init_lane(b, b * T, feat, poses[0])
    eng.step([b * T + 1 for b in range(B)])
    eng.step([b * T + 2 for b in range(B)])
    comm = D.RcclComm(ctx, D.FileRendezvous(tempfile.mkdtemp(prefix="roam_rdv_"), 0, 1))
    try:
        assert comm.info() == (0, 1)
        assert comm.allreduce_max(0.125) == 0.125
        comm.barrier()
        eng.remote_map_reserve(3)                        # the consumer: Map.addKeyframe on every rank, in HBM
        sent = []
        for lane in (0, B - 1):
            got = comm.bcast_keyframe(eng, 0, lane)
            want = eng.live_keyframe(lane)
            assert got["lane"] == lane and got["scan"] == want["scan"]
            for k in ("pose", "velocity", "prunedUndistortedLocals", "peaks"):
                assert np.array_equal(got[k], want[k]), k
            assert len(got["peaks"]) > 1000 and len(got["prunedUndistortedLocals"]) > 20
            sent.append(got)
        for lane in (0, 0):
            sent.append(comm.bcast_keyframe(eng, 0, lane))
        # four keyframes went through a ring of three: the oldest is gone, the others are byte for byte what was broadcast
        assert eng.remote_map_count() == (4, 3)
        for i in range(3):
            kf = eng.remote_map_get(i)
            assert kf["root"] == 0 and kf["lane"] == sent[1 + i]["lane"] and kf["scan"] == sent[1 + i]["scan"]
            for k in ("pose", "velocity", "prunedUndistortedLocals", "peaks"):
                assert np.array_equal(kf[k], sent[1 + i][k]), (i, k)
        with pytest.raises(_ffi.RoamError):
            comm.bcast_keyframe(eng, 0, B)                # a lane that does not exist is refused before any collective starts
    finally:
        comm.close()
    eng.close()
    ctx.close()


def test_pageable_host_memory_is_refused_by_the_async_upload(sequences):
    """roam_engine_upload_scans_async reads the records from the GPU: a pageable numpy buffer must come back as an argument
    error, not as a device fault"""
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    ctx = _ffi.Context(0)
    eng = Engine(1, 2, ctx=ctx)
    rec = np.ascontiguousarray(sequences[0][0][0])
    with pytest.raises(_ffi.RoamError) as ei:
        eng.upload_scans_async(0, rec, n=1)
    assert ei.value.code == _ffi.ROAM_E_ARG and "pinned" in str(ei.value)
    pinned = ctx.host_alloc(rec.shape)
    pinned[...] = rec
    eng.upload_scans_async(0, pinned, n=1)
    eng.synchronize()
    eng.close()
    ctx.close()


def test_rccl_keyframe_exchange_in_the_loop_world1(sequences):
    """BASELINE config 5's loop on one GPU (communicator of size 1): roam_keyframe_exchange after EVERY step, never a host
    synchronisation in between.  The remote map then holds exactly the keyframes the steps made (result flag bit 1), in step
    order, each equal to the lane's live keyframe right after its step; the odometry is untouched by the exchange."""
    import tempfile
    from radarslampy_amd import _ffi
    from radarslampy_amd import distributed as D
    from radarslampy_amd.engine import Engine
    recs, poses, feat = sequences[0]
    T = len(recs)

    def run(exchange):
        ctx = _ffi.Context(0)
        eng = Engine(1, T, ctx=ctx, retrack_on_device=True)
        for t in range(T):
            eng.upload_scan(t, recs[t])
        eng.init_lane(0, 0, feat[:90], poses[0])
        comm = D.RcclComm(ctx, D.FileRendezvous(tempfile.mkdtemp(prefix="roam_rdv_"), 0, 1)) if exchange else None
        out, live = [], []
        polls, mid_first = [], []
        try:
            if exchange:
                eng.remote_map_reserve(16)
            order = list(range(1, T)) + list(range(T - 2, -1, -1))
            for k, t in enumerate(order):
                eng.step([t])
                if exchange:
                    eng.keyframe_exchange(0)
                    if k in (len(order) // 3, 2 * len(order) // 3):       # the map is the loop's consumer API: polled WHILE the loop runs
                        n_mid, res_mid = eng.remote_map_count()
                        polls.append((k, n_mid))
                        if res_mid:
                            mid_first.append(eng.remote_map_get(0))
                if exchange is False:                     # the reference run: read the keyframe a step leaves behind (blocking)
                    r = eng.results()[0]
                    live.append(eng.live_keyframe(0) if r["new_keyframe"] else None)
            for k in range(len(order) - min(8, len(order)), len(order)):
                out.append(eng.results(k)[0])
            got = None
            if exchange:
                n_rec, n_res = eng.remote_map_count()
                got = [eng.remote_map_get(i) for i in range(n_res)]
                assert n_rec == n_res
                # the mid-run polls saw a growing prefix of the final map (kfx_settle used to refuse the second read)
                assert [n for _, n in polls] == sorted(n for _, n in polls) and polls[-1][1] <= n_rec, polls
                assert polls[-1][1] > 0 and n_rec > polls[0][1], (polls, n_rec)
                for kf in mid_first:
                    assert np.array_equal(kf["prunedUndistortedLocals"], got[0]["prunedUndistortedLocals"]) and kf["scan"] == got[0]["scan"]
        finally:
            if comm is not None:
                comm.close()
            eng.close()
            ctx.close()
        return out, live, got

    ref, live, _ = run(False)
    out, _, got = run(True)
    assert [r["pose"].tobytes() for r in out] == [r["pose"].tobytes() for r in ref]
    want = [kf for kf in live if kf is not None]
    assert len(want) >= 1 and len(got) == len(want), (len(want), len(got))
    for g, w in zip(got, want):
        assert g["root"] == 0 and g["lane"] == 0 and g["scan"] == w["scan"]
        for key in ("pose", "velocity", "prunedUndistortedLocals", "peaks"):
            assert np.array_equal(g[key], w[key][:32768] if key == "peaks" else w[key]), key
