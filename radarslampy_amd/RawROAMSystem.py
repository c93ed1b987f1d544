"""Streaming odometry driver on the device-resident engine.

Same entry points as the reference's RawROAMSystem (constructor arguments, run(startSeqInd, endSeqInd), the paramFlags
`rejectOutliers` / `correctMotionDistortion`; reference RawROAMSystem.py:20-160), but not its loop: frames are decoded on
the host (PNG inflate), copied into a pinned staging ring, uploaded on the copy stream while earlier frames are being
processed (roam_engine_upload_scans_async), and every scan pair is ONE roam_engine_step - tracking, outlier rejection,
pose solve, keyframe bookkeeping and the feature re-detection all stay on the GPU.  Poses come back through the engine's
per-step result ring a few frames behind the enqueue front, so the pipeline never drains.  Several sequences can run as
lanes of one engine (`run_many`).  Plotting, video export and the command line are out of scope."""
import os

import numpy as np

from . import _ffi
from .engine import Engine
from .parseData import NativeRecordReader, RecordDecodePool, getRadarImgPaths, prefetchRadarRecords
from .trajectoryPlotting import Trajectory, computePosesRMSE, getGroundTruthTrajectory
from .utils import radarImgPathToTimestamp

RING = 8            # resident scans per lane: frames k-1 .. k+LOOKAHEAD and a margin for the 3-deep step pipeline
LOOKAHEAD = 3       # uploads run this many frames ahead of the step that consumes them
LAG = 2             # results are read this many steps behind the enqueue front


class RawROAMSystem:
    def __init__(self, sequenceName: str, paramFlags: dict = None, hasGroundTruth: bool = True, dataRoot: str = "data",
                 ctx: _ffi.Context = None, decodeWorkers: int = 0) -> None:
        self.decodeWorkers = decodeWorkers              # host processes inflating PNGs ahead of the engine (0: min(32, cores / 2))
        self.sequenceName, self.paramFlags, self.hasGroundTruth, self.dataRoot = sequenceName, dict(paramFlags or {}), hasGroundTruth, dataRoot
        seq = os.path.join(dataRoot, sequenceName)
        self.filePaths = {"data": os.path.join(seq, "radar"), "timestamp": os.path.join(seq, "radar.timestamps")}
        for what, p in self.filePaths.items():
            if not os.path.exists(p):
                raise FileNotFoundError(f"sequence {sequenceName}: no radar {what} at {p}")
        self.imgPathArr = getRadarImgPaths(self.filePaths["data"], self.filePaths["timestamp"])
        self.sequenceSize = len(self.imgPathArr)
        self.ctx = ctx
        self.gtTraj = self.estTraj = None
        self.frameLog = []

    # ------------------------------------------------------------------ one sequence
    def run(self, startSeqInd: int = 0, endSeqInd: int = -1, initPose=None) -> None:
        n = self.sequenceSize
        if endSeqInd < 0:
            endSeqInd = n - 1
        if not (0 <= startSeqInd <= endSeqInd < n):
            raise IndexError(f"frames {startSeqInd}..{endSeqInd} of a {n}-frame sequence")
        frames = list(range(startSeqInd, endSeqInd + 1))
        stamps = [radarImgPathToTimestamp(self.imgPathArr[i]) for i in frames]
        gtPath = os.path.join(self.dataRoot, self.sequenceName, "gt", "radar_odometry.csv")
        if self.hasGroundTruth and os.path.exists(gtPath):
            self.gtTraj = getGroundTruthTrajectory(gtPath)
            if initPose is None:
                initPose = self.gtTraj.getPoseAtTimes(stamps[0])
        initPose = np.zeros(3) if initPose is None else np.asarray(initPose, np.float64)
        # PNG inflate + un-filter run ahead of the loop on the library's pool of host threads (parseData.NativeRecordReader ->
        # roam_png_pool_*), straight into a ring of pinned slots the records are uploaded from: the feeding thread copies nothing
        paths = [self.imgPathArr[i] for i in frames]
        ctx = self.ctx or _ffi.Context(int(os.environ.get("ROAM_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
        try:
            workers = self.decodeWorkers if self.decodeWorkers > 0 else default_decode_workers()
            with NativeRecordReader(min(workers, max(1, len(paths))), ctx=ctx, hold=RING) as rd:
                poses, log = stream_records(rd.records(paths), len(frames), initPose, self.paramFlags, ctx, records_pinned=True)
        finally:
            if self.ctx is None:
                ctx.close()
        self.estTraj = Trajectory([stamps[0]], [initPose])
        self.estTraj.extend_absolute(stamps[1:], poses)
        for k, e in enumerate(log):
            e["seqInd"] = frames[k + 1]
        self.frameLog = log

    def rmse(self):
        """position RMSE of the estimate against the ground truth at the estimate's timestamps"""
        if self.gtTraj is None or self.estTraj is None:
            raise RuntimeError("run() a sequence with ground truth first")
        return computePosesRMSE(self.gtTraj.getPoseAtTimes(self.estTraj.timestamps), self.estTraj.poses)


def default_decode_workers():
    """host threads inflating PNGs for one sequence: half the cores this rank may count on, at most 32"""
    cores = os.cpu_count() or 2
    per_node = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    return max(1, min(32, cores // 2 // per_node))


def stream_records(records, n_frames, init_pose, paramFlags=None, ctx=None, rows=400, stride=3779, payload_off=11, clip=2025,
                   synchronous=False, on_engine=None, after_step=None, before_close=None, records_pinned=False, timing=None):
    """records: iterator of n_frames (rows, stride) u8 Oxford records of ONE sequence.  -> (poses (n_frames-1, 3), per-pair log).
    Frame 0 seeds the lane (features detected on the device); frame k is uploaded LOOKAHEAD frames before step k needs it.
    synchronous=True awaits every pose before the next frame is stepped (the latency of one pair instead of the pipeline's
    rate; same poses) and also returns the per-pair seconds from the step call to the pose on the host.
    records_pinned: every record is a view of pinned memory that stays untouched for RING more frames (NativeRecordReader(hold=RING)):
    it is uploaded from where it lies instead of through this function's staging ring.
    timing: a dict that receives `loop_s` - the seconds from the first step's enqueue to the last pose on the host (the engine's creation,
    the first uploads and the first frame's detection before it, the tear-down after it: what a sequence of any length pays once).
    Hooks (the multi-GPU keyframe exchange of BASELINE config 5 lives in them): on_engine(eng) once after the engine exists,
    after_step(eng, k) after every enqueued step, before_close(eng) when all poses are in."""
    flags = dict(paramFlags or {})
    own = ctx is None
    ctx = ctx or _ffi.Context(int(os.environ.get("ROAM_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    eng = Engine(1, RING, ctx=ctx, rows=rows, stride=stride, payload_off=payload_off, clip=clip,
                 reject_outliers=flags.get("rejectOutliers", True), motion_distortion=flags.get("correctMotionDistortion", True),
                 retrack_on_device=True, stage_events=False)
    if on_engine is not None:
        on_engine(eng)
    pinned = None if records_pinned else ctx.host_alloc((RING, rows, stride))
    it = iter(records)
    uploaded = 0
    # Records that are not pinned go through a staging ring, and the 1.5 MB copy of a frame is a quarter of what this thread spends per
    # frame (enqueue ~140 us, copy ~100 us: the single-sequence rate is the rate of THIS thread).  A feeder thread does the copies ahead
    # (NumPy releases the GIL for them).  Ring slot f % RING is free for frame f once step f - RING has been collected: the copy stream
    # is in order, so the upload of frame f - RING - the slot's only reader - is behind that step.
    import threading
    cv = threading.Condition()
    st = dict(staged=0, collected=-1, done=False, stop=False, err=None)

    def feeder():
        f = 0
        try:
            for rec in it:
                with cv:
                    cv.wait_for(lambda: st["stop"] or f < RING or st["collected"] >= f - RING)
                    if st["stop"]:
                        return
                pinned[f % RING] = rec
                f += 1
                with cv:
                    st["staged"] = f
                    cv.notify_all()
        except BaseException as e:                                            # raised again by the consumer, at its turn
            with cv:
                st["err"] = e
        finally:
            with cv:
                st["done"] = True
                cv.notify_all()

    th = None
    if not records_pinned:
        th = threading.Thread(target=feeder, name="roam-stage", daemon=True)
        th.start()

    def upload_next():
        nonlocal uploaded
        slot = uploaded % RING
        if records_pinned:
            rec = next(it, None)
            if rec is None:
                return False
            eng.upload_scans_async(slot, rec, n=1)
        else:
            with cv:
                cv.wait_for(lambda: st["staged"] > uploaded or st["done"])
                if st["staged"] <= uploaded:
                    if st["err"] is not None:
                        raise st["err"]
                    return False
            eng.upload_scans_async(slot, pinned[slot], n=1)
        uploaded += 1
        return True

    poses = np.empty((max(0, n_frames - 1), 3))
    log = []

    def collect(step):
        r = eng.results(step)[0]
        if th is not None:
            with cv:                                                          # step `step` consumed frame step + 1 (and every upload before it)
                st["collected"] = step + 1
                cv.notify_all()
        poses[step] = r["pose"]
        log.append(dict(n_tracked=r["n_tracked"], n_good=r["n_good"], n_inliers=r["n_inliers"], new_keyframe=r["new_keyframe"],
                        retrack=r["retrack"], n_after_retrack=r["n_after_retrack"], pose=r["pose"].copy(), velocity=r["velocity"].copy()))

    try:
        for _ in range(min(n_frames, 1 + LOOKAHEAD)):
            upload_next()
        eng.synchronize()
        eng.init_lane_detect(0, 0, init_pose)
        lat = []
        import time as _time
        t_loop = _time.perf_counter()
        for k in range(1, n_frames):
            if synchronous:
                import time
                t0 = time.perf_counter()
            eng.step([k % RING])
            if after_step is not None:
                after_step(eng, k)
            # slot (k + LOOKAHEAD) % RING last held frame k + LOOKAHEAD - RING <= k - 5: its steps are behind the fence
            eng.fence()
            upload_next()
            if synchronous:
                collect(k - 1)
                lat.append(time.perf_counter() - t0)
            elif k - 1 - LAG >= 0:
                collect(k - 1 - LAG)
        if not synchronous:
            for s in range(max(0, n_frames - 1 - LAG), n_frames - 1):
                collect(s)
        if timing is not None:
            timing["loop_s"] = _time.perf_counter() - t_loop
        if before_close is not None:
            before_close(eng)
    finally:
        if th is not None:
            with cv:
                st["stop"] = True
                cv.notify_all()
            th.join()
        eng.close()
        if pinned is not None:
            ctx.host_free(pinned)
        if own:
            ctx.close()
    return (poses, log, lat) if synchronous else (poses, log)
