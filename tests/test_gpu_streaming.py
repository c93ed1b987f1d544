"""GPU (-m gpu): BASELINE configs 3 / 4 at length - 500 consecutive scan pairs of one lane along the first 500 ground-truth
motions of the reference's full_seq_1 (tests/golden/full_seq_1_gt_deltas.npz; rendered from an unbounded reflector world
with movers, scintillation and intra-scan distortion), motionDistortion OFF and ON + outlier rejection:
  * every 50th pair the engine's step is re-derived by the oracle from the engine's own state of the pair before
    (features, live keyframe, pose) -> counts, features and pose agree (1e-4 m / 1e-5 rad);
  * steps enqueued back to back (three-stage pipeline, result ring read two steps late, asynchronous uploads) leave
    bit-identical poses, features and keyframes to steps synchronised one by one;
  * keyframe / retrack cycles recover (feature count never stays at zero), the ring buffers wrap hundreds of times, and the
    dead-reckoned position stays within 5 % of the distance travelled (slow movers inside the 0.5 m consistency threshold bias every estimate - the reference's algorithm, not the arithmetic)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
N = 500
HERE = os.path.dirname(os.path.abspath(__file__))


_SEQ = {}


def sequence(distorted):
    """config 3 renders the scans without intra-scan distortion, config 4 with it"""
    if distorted not in _SEQ:
        from radarslampy_amd import synth
        deltas = np.load(os.path.join(HERE, "golden", "full_seq_1_gt_deltas.npz"))["deltas"][:N]
        poses = synth.poses_from_deltas(deltas)
        world = synth.StreamWorld(11, mover_fraction=0.15)
        jobs = synth.stream_jobs(world, poses, distortion=distorted, scintillation=0.4)
        with mp.get_context("spawn").Pool(min(48, os.cpu_count() or 4)) as pool:
            _SEQ[distorted] = (pool.map(synth._render_job, jobs, chunksize=4), poses)
    return _SEQ[distorted]


def _run(recs, pose0, md, pipelined, probe_every=0, ahead=3):
    """-> poses (N, 3), final (features, keyframe), probes {step: (features, keyframe, pose) BEFORE that step}"""
    from radarslampy_amd import _ffi
    from radarslampy_amd.engine import Engine
    ctx = _ffi.Context(0)
    RING = 8
    eng = Engine(1, RING, ctx=ctx, motion_distortion=md, reject_outliers=True, retrack_on_device=True)
    pinned = ctx.host_alloc((RING, 400, 3779))
    n = len(recs)
    poses, flags, probes = np.empty((n - 1, 3)), np.zeros(n - 1, int), {}

    def up(k):
        pinned[k % RING] = recs[k]
        eng.upload_scans_async(k % RING, pinned[k % RING], n=1)

    for k in range(max(ahead, 1) + 1):
        up(k)
    eng.synchronize()
    eng.init_lane_detect(0, 0, pose0)
    for k in range(1, n):
        if probe_every and k % probe_every == 0:
            r = eng.results()[0] if k > 1 else dict(pose=np.asarray(pose0, float))
            probes[k] = (eng.lane_features(0), eng.live_keyframe(0), r["pose"].copy())
        eng.step([k % RING])
        eng.fence()
        if k + ahead + 1 < n and ahead == 0:
            up(k + 1)                                        # the NEXT step's own scan, behind the fence: that step has to wait for it
            pinned[(k + 4) % RING] = recs[min(k + 4, n - 1)]
            eng.upload_scans_async((k + 4) % RING, pinned[(k + 4) % RING], n=1)      # ... and not for this one, of a slot it does not read
        elif k + 3 < n and ahead:
            up(k + 3)
        if pipelined:
            if k - 3 >= 0:
                a = eng.results_array(k - 3)
                poses[k - 3], flags[k - 3] = a["pose"][0], a["flags"][0]
        else:
            a = eng.results_array()
            poses[k - 1], flags[k - 1] = a["pose"][0], a["flags"][0]
    for s in range(max(0, n - 4), n - 1):
        a = eng.results_array(s)
        poses[s], flags[s] = a["pose"][0], a["flags"][0]
    final = (eng.lane_features(0), eng.live_keyframe(0))
    ctx.host_free(pinned)
    eng.close()
    ctx.close()
    return poses, flags, final, probes


def test_a_step_waits_for_the_uploads_of_its_own_scans():
    """roam_engine_step waits for the newest asynchronous upload among the pool slots IT reads (not for every upload enqueued so far: that
    one may sit behind a fence).  Frame k + 1 uploaded right after step k - behind the fence, with a later upload of another slot queued
    after it - gives the results of the run that uploads three frames ahead, record for record."""
    recs, poses = sequence(True)
    recs = recs[:60]
    a = _run(recs, poses[0], True, True)
    b = _run(recs, poses[0], True, True, ahead=0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2][0], b[2][0])


@pytest.mark.parametrize("md", [False, True])
def test_500_consecutive_pairs(md):
    recs, gt = sequence(md)
    sync_poses, flags, sync_final, probes = _run(recs, gt[0], md, pipelined=False, probe_every=50)
    pipe_poses, pflags, pipe_final, _ = _run(recs, gt[0], md, pipelined=True)
    # pipelined == synchronised, bit for bit
    assert sync_poses.tobytes() == pipe_poses.tobytes() and np.array_equal(flags, pflags)
    assert np.array_equal(sync_final[0], pipe_final[0])
    for k in ("pose", "velocity", "prunedUndistortedLocals", "peaks"):
        assert np.array_equal(sync_final[1][k], pipe_final[1][k]), k
    # the engine's step re-derived by the oracle from the engine's state, every 50th pair
    for s, (feat, kf, pose) in sorted(probes.items()):
        p = oracle.OdometryPipeline(recs[s - 1], feat, pose, motion_distortion=md, detect=lambda c: oracle.getFeatures(c)[0])
        p.old_kf.pose, p.old_kf.velocity, p.old_kf.prunedUndistortedLocals = kf["pose"], kf["velocity"], kf["prunedUndistortedLocals"]
        want = p.step(recs[s])
        got = sync_poses[s - 1]
        assert bool(flags[s - 1] & 4) == bool(want["retrack"]) and bool(flags[s - 1] & 2) == bool(want["new_keyframe"]), s
        assert np.abs(got[:2] - want["pose"][:2]).max() <= 1e-4 and abs(got[2] - want["pose"][2]) <= 1e-5, (s, got, want["pose"])
    # retrack / keyframe cycles happened and recovered; nothing overflowed
    n_rt = int(np.count_nonzero(flags & 8))
    assert n_rt >= 20 and np.all(((flags >> 8) & 15) == 0), n_rt
    assert np.count_nonzero(flags & 2) >= 100
    # dead-reckoned drift over the ~500 m driven
    dist = np.hypot(*np.diff(gt[:, :2], axis=0).T).sum()
    err = np.hypot(*(sync_poses[:, :2] - gt[1:, :2]).T)
    print(f"md={md}: position RMSE {np.sqrt(np.mean(err ** 2)):.3f} m over {dist:.1f} m, final error {err[-1]:.3f} m, retracks {n_rt}")
    assert np.sqrt(np.mean(err ** 2)) < 0.05 * dist, (np.sqrt(np.mean(err ** 2)), dist)
