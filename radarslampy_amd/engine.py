"""Batched device-resident scan-pair engine (Python face of the roam_engine_* C-ABI).

B independent sequences advance in lock-step: one `step()` = one scan pair per lane,
everything between the raw u8 record in HBM and the SE(2) pose on the device.  This is the
MI355X-first shape of RawROAMSystem.run's loop body (reference RawROAMSystem.py:162-298):
the reference's single sequential loop becomes the batch dimension that fills 256 CUs."""
import ctypes as C

import numpy as np

from . import _ffi


class Engine:
    def __init__(self, lanes: int, pool_scans: int, ctx: _ffi.Context = None, rows=400, stride=3779, payload_off=11,
                 clip=2025, peaks_cap=65536, reject_outliers=True, motion_distortion=True, clique_node_limit=0,
                 sigma5=(4.0, 4.0, 1.0, 1.0, (5 * np.pi / 180) ** 2), retrack_on_device=False, retrack_slots=0,
                 keyframe_trans_m=0.0, keyframe_rot_rad=0.0, stage_events=True):
        """keyframe_trans_m / keyframe_rot_rad: Map.isGoodKeyframe's thresholds (Mapping.py:13-15); 0 = the reference's 2.0 m / 0.2 rad.
        stage_events=False: no timestamp events in the step (stage_times / the detection figures of kernel_avg are then unavailable):
        what the single-sequence driver does - they are ~50 us of its pair"""
        self.ctx = ctx or _ffi.default_context()
        self.lib = self.ctx.lib
        cfg = _ffi.EngineCfg(lanes, rows, stride, payload_off, clip, pool_scans, peaks_cap, int(reject_outliers),
                             int(motion_distortion), int(clique_node_limit), (C.c_double * 5)(*sigma5), int(retrack_on_device),
                             int(retrack_slots), float(keyframe_trans_m), float(keyframe_rot_rad))
        self.cfg = cfg
        self.lanes, self.pool_scans = lanes, pool_scans
        self.rows, self.stride = rows, stride
        self.ctx.check(self.lib.roam_engine_create(self.ctx.h, C.byref(cfg)))
        if not stage_events:
            self.ctx.check(self.lib.roam_engine_set_stage_events(self.ctx.h, 0))
        self._res = (_ffi.LaneResult * lanes)()

    def close(self):
        if self.ctx is not None and getattr(self.ctx, "h", None):
            self.lib.roam_engine_destroy(self.ctx.h)
        self.ctx = None

    def upload_scan(self, pool_idx: int, rec: np.ndarray):
        rec = np.ascontiguousarray(rec, np.uint8)
        assert rec.shape == (self.rows, self.stride), rec.shape
        self.ctx.check(self.lib.roam_engine_upload_scan(self.ctx.h, int(pool_idx), _ffi._ptr(rec)))

    def upload_scans_async(self, pool_idx0: int, pinned_records: np.ndarray, n: int = None, stride: int = None):
        """asynchronous upload of n records from PINNED host memory (Context.host_alloc) on the copy stream"""
        rec_bytes = self.rows * self.stride
        n = pinned_records.size // rec_bytes if n is None else n
        stride = rec_bytes if stride is None else stride
        self.ctx.check(self.lib.roam_engine_upload_scans_async(self.ctx.h, int(pool_idx0), int(n), _ffi._ptr(pinned_records), int(stride)))

    def fence(self):
        self.ctx.check(self.lib.roam_engine_fence(self.ctx.h))

    def copy_scan(self, dst_idx: int, src_idx: int):
        self.ctx.check(self.lib.roam_engine_copy_scan(self.ctx.h, int(dst_idx), int(src_idx)))

    def init_lane(self, lane: int, pool_idx: int, pts: np.ndarray, pose):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        pose = np.ascontiguousarray(pose, np.float64)
        self.ctx.check(self.lib.roam_engine_init_lane(self.ctx.h, int(lane), int(pool_idx), _ffi._ptr(pts), pts.shape[0],
                                                      _ffi._ptr(pose)))

    def init_lane_detect(self, lane: int, pool_idx: int, pose):
        """init_lane with the first features detected on the device (needs retrack_on_device=True)"""
        pose = np.ascontiguousarray(pose, np.float64)
        self.ctx.check(self.lib.roam_engine_init_lane_detect(self.ctx.h, int(lane), int(pool_idx), _ffi._ptr(pose)))

    def init_lanes_detect(self, lane0: int, pool_idx, poses):
        """init_lane_detect for the lanes lane0 .. lane0 + n - 1 in ONE device pass (pool_idx (n,), poses (n, 3))"""
        pool_idx = np.ascontiguousarray(pool_idx, np.int32).ravel()
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 3)
        assert len(pool_idx) == len(poses)
        self.ctx.check(self.lib.roam_engine_init_lanes_detect(self.ctx.h, int(lane0), len(pool_idx), _ffi._ptr(pool_idx), _ffi._ptr(poses)))

    # ---- 8e consumer: the global map of received keyframes (Mapping.Map.addKeyframe on every rank)
    def remote_map_reserve(self, keyframes: int = 64):
        self.ctx.check(self.lib.roam_remote_map_reserve(self.ctx.h, int(keyframes)))

    def keyframe_exchange(self, lane: int = 0):
        """collective, non-blocking (roam_keyframe_exchange): call on every rank after each step"""
        self.ctx.check(self.lib.roam_keyframe_exchange(self.ctx.h, int(lane)))

    def debug_keyframe_append(self, recv: np.ndarray = None, world: int = 1):
        """test / debug (roam_debug_keyframe_append): `recv` = (world, rec_bytes) u8 as an all-gather would leave it -> this rank's remote
        map, through the exchange's own append kernel.  Returns the record layout dict(rec_bytes, locals_off, peaks_off, max_peaks)."""
        rb, lo, po, mp = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        if recv is not None:
            recv = np.ascontiguousarray(recv, np.uint8)
        self.ctx.check(self.lib.roam_debug_keyframe_append(self.ctx.h, _ffi._ptr(recv) if recv is not None else None, int(world),
                                                           C.byref(rb), C.byref(lo), C.byref(po), C.byref(mp)))
        return dict(rec_bytes=rb.value, locals_off=lo.value, peaks_off=po.value, max_peaks=mp.value)

    def remote_map_count(self):
        """(keyframes received so far, keyframes resident in the ring)"""
        rec, res = C.c_int64(0), C.c_int32(0)
        self.ctx.check(self.lib.roam_remote_map_count(self.ctx.h, C.byref(rec), C.byref(res)))
        return rec.value, res.value

    def remote_map_get(self, index: int) -> dict:
        """received keyframe `index` (0 = the oldest resident): the dict RcclComm.bcast_keyframe returns + the sending rank"""
        hdr, root = _ffi.KeyframeHdr(), C.c_int32(-1)
        loc = np.empty((_ffi.MAX_FEATURES, 2), np.float64)
        pk = np.empty((self.cfg.peaks_cap, 2), np.int32)
        self.ctx.check(self.lib.roam_remote_map_get(self.ctx.h, int(index), C.byref(hdr), C.byref(root), _ffi._ptr(loc), loc.shape[0],
                                                    _ffi._ptr(pk), pk.shape[0]))
        return dict(pose=np.array(hdr.pose[:]), velocity=np.array(hdr.velocity[:]), prunedUndistortedLocals=loc[:hdr.n_features].copy(),
                    peaks=pk[:hdr.n_peaks].copy(), scan=hdr.scan, lane=hdr.lane, root=root.value)

    def set_features(self, lane: int, pts: np.ndarray):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        self.ctx.check(self.lib.roam_engine_set_features(self.ctx.h, int(lane), _ffi._ptr(pts), pts.shape[0]))

    # ---- 8f-f1: device-resident keyframe map (Mapping.Map.keyframes)
    def map_reserve(self, keyframes_per_lane: int = 16):
        """keep every keyframe of every lane in HBM (call once, before the first init_lane)"""
        self.ctx.check(self.lib.roam_engine_map_reserve(self.ctx.h, int(keyframes_per_lane)))

    def map_count(self, lane: int) -> int:
        n = C.c_int32(0)
        self.ctx.check(self.lib.roam_engine_map_count(self.ctx.h, int(lane), C.byref(n)))
        return n.value

    def map_keyframe(self, lane: int, index: int) -> dict:
        """keyframe `index` of the lane's map (0 = oldest, map_count-1 = live): pose, velocity at creation,
        prunedUndistortedLocals (n, 2) in metres, pool scan it was created on"""
        pose, vel = np.empty(3), np.empty(3)
        loc = np.empty((_ffi.MAX_FEATURES, 2))
        n, sc = C.c_int32(0), C.c_int32(0)
        self.ctx.check(self.lib.roam_engine_map_get(self.ctx.h, int(lane), int(index), _ffi._ptr(pose), _ffi._ptr(vel),
                                                    _ffi._ptr(loc), loc.shape[0], C.byref(n), C.byref(sc)))
        return dict(pose=pose, velocity=vel, prunedUndistortedLocals=loc[:n.value].copy(), scan=sc.value)

    def live_keyframe(self, lane: int) -> dict:
        """the keyframe the tracker is pruning right now (Mapping.Map.keyframes[-1]) + the latest polar point cloud"""
        kf = self.map_keyframe(lane, self.map_count(lane) - 1)
        kf.update(peaks=self.lane_peaks(lane), lane=lane)
        return kf

    def map_keyframes(self, lane: int):
        return [self.map_keyframe(lane, i) for i in range(self.map_count(lane))]

    def step(self, scan_idx):
        idx = np.ascontiguousarray(scan_idx, np.int32)
        assert idx.shape == (self.lanes,)
        self.ctx.check(self.lib.roam_engine_step(self.ctx.h, _ffi._ptr(idx)))

    def synchronize(self):
        self.ctx.check(self.lib.roam_synchronize(self.ctx.h))

    def set_retrack(self, mode: int):
        """0 = suspended, 1 = lanes that ran out of features (default), 2 = every lane every step (measurement)"""
        self.ctx.check(self.lib.roam_engine_set_retrack(self.ctx.h, int(mode)))

    def steps_enqueued(self) -> int:
        n = C.c_int64(0)
        self.ctx.check(self.lib.roam_engine_steps_enqueued(self.ctx.h, C.byref(n)))
        return n.value

    def results_array(self, step: int = None):
        """the same records as one structured numpy array (pose, velocity, counts, flags): cheap for thousands of lanes"""
        if step is None:
            self.ctx.check(self.lib.roam_engine_results(self.ctx.h, self._res, self.lanes))
        else:
            self.ctx.check(self.lib.roam_engine_step_results(self.ctx.h, int(step), self._res, self.lanes))
        return np.ctypeslib.as_array(self._res).copy()

    def results(self, step: int = None):
        """per-lane records of the last step, or of step `step` (0-based; only that step is waited for - the ring keeps
        the last 8 steps, so poses / flags can be consumed while later steps are still running)"""
        if step is None:
            self.ctx.check(self.lib.roam_engine_results(self.ctx.h, self._res, self.lanes))
        else:
            self.ctx.check(self.lib.roam_engine_step_results(self.ctx.h, int(step), self._res, self.lanes))
        out = []
        for r in self._res:
            out.append(dict(pose=np.array(r.pose[:]), velocity=np.array(r.velocity[:]),
                            R=np.array(r.kabsch_R[:]).reshape(2, 2), h=np.array(r.kabsch_h[:]).reshape(2, 1),
                            n_tracked=r.n_tracked, n_good=r.n_good, n_inliers=r.n_inliers, n_peaks=r.n_peaks,
                            lm_nfev=r.lm_nfev, lm_info=r.lm_info, clique_proven=bool(r.flags & 1),
                            new_keyframe=bool(r.flags & 2), retrack=bool(r.flags & 4), retracked_on_device=bool(r.flags & 8),
                            detect_overflow=(r.flags >> 8) & 15, n_after_retrack=r.n_after_retrack))
        return out

    def lane_features(self, lane: int):
        pts = np.empty((_ffi.MAX_FEATURES, 2), np.float32)
        K = C.c_int32(0)
        self.ctx.check(self.lib.roam_engine_lane_features(self.ctx.h, int(lane), _ffi._ptr(pts), _ffi.MAX_FEATURES, C.byref(K)))
        return pts[:K.value].copy()

    def lane_peaks(self, lane: int):
        cap = self.cfg.peaks_cap
        out = np.empty((cap, 2), np.int32)
        n = C.c_int64(0)
        self.ctx.check(self.lib.roam_engine_lane_peaks(self.ctx.h, int(lane), _ffi._ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def detect_features(self, pool_idx: int):
        """getFeatures (getFeatures.py:74-95) on a resident scan: DoH maxima on the device, then the
        blob_doh bookkeeping, SSC-ANMS (device) and the [x, y] flip."""
        from . import getFeatures as gf
        p = gf.DEFAULT_FEATURE_PARAMS
        sig = np.linspace(p["min_sigma"], p["max_sigma"], p["num_sigma"])
        cap = 1 << 18
        rcs = np.empty((cap, 3), np.int32)
        val = np.empty(cap, np.float64)
        n = C.c_int32(0)
        self.ctx.check(self.lib.roam_engine_doh_maxima(self.ctx.h, int(pool_idx), _ffi._ptr(sig), len(sig), float(p["threshold"]),
                                                       _ffi._ptr(rcs), _ffi._ptr(val), cap, C.byref(n)))
        blobs = gf.blobs_from_maxima(rcs[:n.value], val[:n.value], sig)
        W = 2 * (self.cfg.clip // 2)
        blobs = gf.adaptiveNMS(np.empty((W, W), np.bool_), blobs) if len(blobs) else blobs
        return np.fliplr(blobs[:, :2])

    def retrack_lane(self, lane: int, pool_idx: int):
        """appendNewFeatures(currImgCart, good_new) + keyframe refresh (RawROAMSystem.py:264-270)."""
        from . import getFeatures as gf
        new = self.detect_features(pool_idx)
        pts = gf.dedupe_append(self.lane_features(lane), new)
        if len(pts) > _ffi.MAX_FEATURES:
            pts = pts[:_ffi.MAX_FEATURES]
        self.set_features(lane, pts)
        return pts

    def lane_image(self, lane: int, level: int = 0):
        W = 2 * (self.cfg.clip // 2)
        for _ in range(level):
            W = (W + 1) // 2
        out = np.empty((W, W), np.uint8)
        self.ctx.check(self.lib.roam_engine_lane_image(self.ctx.h, int(lane), int(level), _ffi._ptr(out), out.size))
        return out

    def stage_times(self):
        ms = (C.c_float * 16)()
        names = (C.c_char_p * 16)()
        n = C.c_int32(0)
        self.ctx.check(self.lib.roam_engine_stage_times(self.ctx.h, ms, names, 16, C.byref(n)))
        return {names[i].decode(): float(ms[i]) for i in range(n.value)}

    def kernel_avg(self, name: str, last_steps: int):
        """(average in-step launch time in ms, steps averaged) of a front-end kernel over the last steps"""
        ms, n = C.c_float(0), C.c_int32(0)
        self.ctx.check(self.lib.roam_engine_kernel_avg(self.ctx.h, name.encode(), int(last_steps), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def detect_chunk(self) -> int:
        """detections per launch of a detection kernel inside a step (roam_engine_detect_chunk): the chunks of kernel_chunk_ms"""
        n = C.c_int32(0)
        self.ctx.check(self.lib.roam_engine_detect_chunk(self.ctx.h, C.byref(n)))
        return n.value

    def kernel_chunk_ms(self, name: str, last_steps: int):
        """(steps, chunks) float32 array: launch duration in ms of every detection chunk of the last steps, oldest first
        (-1: the step ran without device-side detection) - roam_engine_kernel_chunk_ms"""
        buf = np.zeros(int(last_steps) * 16, dtype=np.float32)
        nc, ns = C.c_int32(0), C.c_int32(0)
        self.ctx.check(self.lib.roam_engine_kernel_chunk_ms(self.ctx.h, name.encode(), int(last_steps), buf.ctypes.data_as(C.POINTER(C.c_float)),
                                                            buf.size, C.byref(nc), C.byref(ns)))
        return buf[: ns.value * nc.value].reshape(ns.value, nc.value)

    def debug_detect(self, fused: bool, n_slots: int, cap: int = 2048, s_slot: int = None):
        """diagnostics (roam_engine_debug_detect): candidate lists of n_slots detections from the two-kernel form or the fused kernel;
        returns (counts (n,), rc (n, cap) uint32, val (n, cap) float64[, S (W, W) float64 of slot s_slot])"""
        n = np.zeros(n_slots, np.int32)
        rc = np.zeros((n_slots, cap), np.uint32)
        val = np.zeros((n_slots, cap), np.float64)
        W = 2 * (self.cfg.clip // 2)
        S = np.zeros((W, W), np.float64) if s_slot is not None else None
        self.ctx.check(self.lib.roam_engine_debug_detect(self.ctx.h, int(bool(fused)), int(n_slots), int(cap), _ffi._ptr(rc), _ffi._ptr(val),
                                                         _ffi._ptr(n), int(s_slot or 0), _ffi._ptr(S) if S is not None else None))
        return (n, rc, val) if S is None else (n, rc, val, S)

    def time_kernel(self, name: str, reps: int = 20):
        ms = C.c_float(0)
        by = C.c_double(0)
        self.ctx.check(self.lib.roam_engine_time_kernel(self.ctx.h, name.encode(), int(reps), C.byref(ms), C.byref(by)))
        return float(ms.value), float(by.value)
