"""ctypes binding of libroam_hip.so (include/roam_abi.h).  No torch, no CPU fallback: if
the HIP library or a gfx950 device is missing every compute call raises RoamError."""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ROAM_LIB") or os.path.join(_HERE, "csrc", "libroam_hip.so")     # ROAM_LIB: an A/B build (profiles/build_variant.py)

ROAM_OK, ROAM_E_ARG, ROAM_E_HIP, ROAM_E_CAPACITY, ROAM_E_NODEVICE, ROAM_E_STATE = 0, -1, -2, -3, -4, -5
MAX_FEATURES = 1024
STEP_NEW_SEQUENCE = 0x40000000      # roam_abi.h ROAM_STEP_NEW_SEQUENCE: OR into a lane's scan index


class RoamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libroam_hip error {code}: {msg}")
        self.code = code


class EngineCfg(C.Structure):
    _fields_ = [("lanes", C.c_int32), ("rows", C.c_int32), ("stride", C.c_int32), ("payload_off", C.c_int32),
                ("clip", C.c_int32), ("pool_scans", C.c_int32), ("peaks_cap", C.c_int32),
                ("reject_outliers", C.c_int32), ("motion_distortion", C.c_int32),
                ("clique_node_limit", C.c_int64), ("sigma5", C.c_double * 5), ("retrack_on_device", C.c_int32),
                ("retrack_slots", C.c_int32), ("keyframe_trans_m", C.c_double), ("keyframe_rot_rad", C.c_double)]


class LaneResult(C.Structure):
    _fields_ = [("pose", C.c_double * 3), ("velocity", C.c_double * 3), ("kabsch_R", C.c_double * 4),
                ("kabsch_h", C.c_double * 2), ("n_tracked", C.c_int32), ("n_good", C.c_int32),
                ("n_inliers", C.c_int32), ("n_peaks", C.c_int32), ("lm_nfev", C.c_int32),
                ("lm_info", C.c_int32), ("flags", C.c_int32), ("n_after_retrack", C.c_int32)]


class KeyframeHdr(C.Structure):
    _fields_ = [("pose", C.c_double * 3), ("velocity", C.c_double * 3), ("n_features", C.c_int32), ("n_peaks", C.c_int32),
                ("scan", C.c_int32), ("lane", C.c_int32)]


COMM_ID_BYTES = 128
_P = C.POINTER
_vp = C.c_void_p
_SIGS = {
    "roam_create": (C.c_int32, [C.c_int32, _P(_vp)]),
    "roam_destroy": (C.c_int32, [_vp]),
    "roam_last_error": (C.c_char_p, [_vp]),
    "roam_version": (C.c_char_p, []),
    "roam_device_info": (C.c_int32, [_vp, C.c_char_p, C.c_int32, _P(C.c_int32), _P(C.c_int64), C.c_char_p, C.c_int32]),
    "roam_synchronize": (C.c_int32, [_vp]),
    "roam_host_alloc": (C.c_int32, [_vp, C.c_int64, _P(_vp)]),
    "roam_host_free": (C.c_int32, [_vp, _vp]),
    "roam_png_decode_gray8": (C.c_int32, [_vp, C.c_int64, _vp, C.c_int64, C.c_int64, _P(C.c_int32), _P(C.c_int32)]),
    "roam_png_decode_file": (C.c_int32, [C.c_char_p, _vp, C.c_int64, C.c_int64, _P(C.c_int32), _P(C.c_int32)]),
    "roam_png_pool_create": (C.c_int32, [C.c_int32, _P(_vp)]),
    "roam_png_pool_submit": (C.c_int32, [_vp, C.c_char_p, _vp, C.c_int64, C.c_int64, C.c_int64]),
    "roam_png_pool_wait": (C.c_int32, [_vp, C.c_int64, _P(C.c_int32), _P(C.c_int32)]),
    "roam_png_pool_destroy": (C.c_int32, [_vp]),
    "roam_peaks_polar_f32": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32, _vp, C.c_int64, _P(C.c_int64)]),
    "roam_peaks_record_u8": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _vp, C.c_int64, _P(C.c_int64)]),
    "roam_polar_to_cart_f32": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32, _vp, _vp]),
    "roam_polar_to_cart_record_u8": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _vp, _vp]),
    "roam_klt_track_u8": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp, _vp]),
    "roam_klt_track_f32": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp, _vp]),
    "roam_pyr_down_u8": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32, _vp]),
    "roam_reject_outliers": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_double, C.c_int64, _vp, _P(C.c_int32), _P(C.c_int32), _vp]),
    "roam_time_reject_outliers": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_double, C.c_int64, C.c_int32, _P(C.c_float), _P(C.c_float),
                                               _P(C.c_int32), _P(C.c_int32)]),
    "roam_kabsch2d": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, _vp]),
    "roam_mds_solve": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_int32, _vp, _vp, C.c_double, _vp, _P(C.c_int32), _P(C.c_int32), _vp, _vp]),
    "roam_mds_undistort": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_double, _vp, _vp]),
    "roam_ssc": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, _vp, _P(C.c_int32)]),
    "roam_doh_maxima": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32, _vp, C.c_int32, C.c_double, _vp, _vp, C.c_int32, _P(C.c_int32)]),
    "roam_engine_create": (C.c_int32, [_vp, _P(EngineCfg)]),
    "roam_engine_destroy": (C.c_int32, [_vp]),
    "roam_engine_upload_scan": (C.c_int32, [_vp, C.c_int32, _vp]),
    "roam_engine_upload_scans_async": (C.c_int32, [_vp, C.c_int32, C.c_int32, _vp, C.c_int64]),
    "roam_engine_fence": (C.c_int32, [_vp]),
    "roam_engine_copy_scan": (C.c_int32, [_vp, C.c_int32, C.c_int32]),
    "roam_engine_init_lane": (C.c_int32, [_vp, C.c_int32, C.c_int32, _vp, C.c_int32, _vp]),
    "roam_engine_step": (C.c_int32, [_vp, _vp]),
    "roam_engine_results": (C.c_int32, [_vp, _P(LaneResult), C.c_int32]),
    "roam_engine_step_results": (C.c_int32, [_vp, C.c_int64, _P(LaneResult), C.c_int32]),
    "roam_engine_steps_enqueued": (C.c_int32, [_vp, _P(C.c_int64)]),
    "roam_engine_set_retrack": (C.c_int32, [_vp, C.c_int32]),
    "roam_engine_init_lane_detect": (C.c_int32, [_vp, C.c_int32, C.c_int32, _vp]),
    "roam_engine_init_lanes_detect": (C.c_int32, [_vp, C.c_int32, C.c_int32, _vp, _vp]),
    "roam_engine_lane_features": (C.c_int32, [_vp, C.c_int32, _vp, C.c_int32, _P(C.c_int32)]),
    "roam_engine_lane_peaks": (C.c_int32, [_vp, C.c_int32, _vp, C.c_int64, _P(C.c_int64)]),
    "roam_engine_doh_maxima": (C.c_int32, [_vp, C.c_int32, _vp, C.c_int32, C.c_double, _vp, _vp, C.c_int32, _P(C.c_int32)]),
    "roam_engine_lane_image": (C.c_int32, [_vp, C.c_int32, C.c_int32, _vp, C.c_int64]),
    "roam_engine_set_features": (C.c_int32, [_vp, C.c_int32, _vp, C.c_int32]),
    "roam_engine_kernel_avg": (C.c_int32, [_vp, C.c_char_p, C.c_int32, _P(C.c_float), _P(C.c_int32)]),
    "roam_engine_kernel_chunk_ms": (C.c_int32, [_vp, C.c_char_p, C.c_int32, _P(C.c_float), C.c_int32, _P(C.c_int32), _P(C.c_int32)]),
    "roam_engine_detect_chunk": (C.c_int32, [_vp, _P(C.c_int32)]),
    "roam_engine_map_reserve": (C.c_int32, [_vp, C.c_int32]),
    "roam_engine_map_count": (C.c_int32, [_vp, C.c_int32, _P(C.c_int32)]),
    "roam_engine_map_get": (C.c_int32, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_int32, _P(C.c_int32), _P(C.c_int32)]),
    "roam_engine_stage_times": (C.c_int32, [_vp, _vp, _P(C.c_char_p), C.c_int32, _P(C.c_int32)]),
    "roam_engine_set_stage_events": (C.c_int32, [_vp, C.c_int32]),
    "roam_engine_time_kernel": (C.c_int32, [_vp, C.c_char_p, C.c_int32, _P(C.c_float), _P(C.c_double)]),
    "roam_engine_debug_detect": (C.c_int32, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_int32, _vp]),
    "roam_fmt_rotation": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "roam_prune_blobs": (C.c_int32, [_vp, C.c_int32, C.c_double, _vp]),
    "roam_argsort_np122": (C.c_int32, [_vp, C.c_int32, _vp]),
    "roam_comm_available": (C.c_int32, []),
    "roam_comm_unique_id": (C.c_int32, [_vp]),
    "roam_comm_init": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32]),
    "roam_comm_destroy": (C.c_int32, [_vp]),
    "roam_comm_info": (C.c_int32, [_vp, _P(C.c_int32), _P(C.c_int32)]),
    "roam_comm_allreduce_f64": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32]),
    "roam_comm_barrier": (C.c_int32, [_vp]),
    "roam_bcast_keyframe": (C.c_int32, [_vp, C.c_int32, C.c_int32, _P(KeyframeHdr), _vp, C.c_int32, _vp, C.c_int64]),
    "roam_keyframe_exchange": (C.c_int32, [_vp, C.c_int32]),
    "roam_remote_map_reserve": (C.c_int32, [_vp, C.c_int32]),
    "roam_remote_map_count": (C.c_int32, [_vp, _P(C.c_int64), _P(C.c_int32)]),
    "roam_debug_keyframe_append": (C.c_int32, [_vp, _vp, C.c_int32, _P(C.c_int64), _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "roam_remote_map_get": (C.c_int32, [_vp, C.c_int32, _P(KeyframeHdr), _P(C.c_int32), _vp, C.c_int32, _vp, C.c_int64]),
}
ABI_SYMBOLS = tuple(_SIGS)

_lib = None


def load_library():
    """dlopen libroam_hip.so and declare every signature of include/roam_abi.h."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RoamError(ROAM_E_NODEVICE, f"{LIB_PATH} not built - run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(LIB_PATH)
        partial = bool(os.environ.get("ROAM_LIB_PARTIAL"))      # a host-only build (profiles/asan_cpu.sh): the symbols it has
        for name, (res, args) in _SIGS.items():
            if partial and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


class Context:
    """One GPU + one HIP stream (roam_ctx).  Not thread-safe; make one per thread/GPU."""

    def __init__(self, device_id: int = 0):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.roam_create(int(device_id), C.byref(h))
        if rc != ROAM_OK:
            raise RoamError(rc, "roam_create failed: no usable MI355X/gfx950 device (the product path has no CPU fallback)")
        self.h = h
        self.device_id = device_id

    def close(self):
        if getattr(self, "h", None):
            self.lib.roam_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, ok=(ROAM_OK,)):
        if rc not in ok:
            raise RoamError(rc, self.lib.roam_last_error(self.h).decode(errors="replace"))
        return rc

    def device_info(self):
        name = C.create_string_buffer(256)
        arch = C.create_string_buffer(64)
        cu, mem = C.c_int32(0), C.c_int64(0)
        self.check(self.lib.roam_device_info(self.h, name, 256, C.byref(cu), C.byref(mem), arch, 64))
        return dict(name=name.value.decode(), arch=arch.value.decode(), cu_count=cu.value, hbm_bytes=mem.value)

    def host_alloc(self, shape, dtype=np.uint8):
        """numpy array backed by pinned host memory (freed with host_free)"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = _vp()
        self.check(self.lib.roam_host_alloc(self.h, n, C.byref(p)))
        buf = (C.c_uint8 * n).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self.check(self.lib.roam_host_free(self.h, p))

    # ---- stage API -------------------------------------------------------------------
    def peaks_polar_f32(self, polar):
        img = np.ascontiguousarray(polar, np.float32)
        rows, cols = img.shape
        cap = rows * ((cols + 1) // 2)
        out = np.empty((cap, 2), np.int32)
        n = C.c_int64(0)
        self.check(self.lib.roam_peaks_polar_f32(self.h, _ptr(img), rows, cols, _ptr(out), cap, C.byref(n)))
        return out[:n.value]

    def peaks_record_u8(self, rec, payload_off=11, clip=2025):
        rec = np.ascontiguousarray(rec, np.uint8)
        rows, stride = rec.shape
        cap = rows * ((clip + 1) // 2)
        out = np.empty((cap, 2), np.int32)
        n = C.c_int64(0)
        self.check(self.lib.roam_peaks_record_u8(self.h, _ptr(rec), rows, stride, payload_off, clip, _ptr(out), cap, C.byref(n)))
        return out[:n.value]

    def polar_to_cart_f32(self, polar, want_f32=True, want_u8=False):
        img = np.ascontiguousarray(polar, np.float32)
        rows, cols = img.shape
        W = 2 * (cols // 2)
        f = np.empty((W, W), np.float32) if want_f32 else None
        u = np.empty((W, W), np.uint8) if want_u8 else None
        self.check(self.lib.roam_polar_to_cart_f32(self.h, _ptr(img), rows, cols, _ptr(f), _ptr(u)))
        return f, u

    def polar_to_cart_record_u8(self, rec, payload_off=11, clip=2025, want_f32=False, want_u8=True):
        rec = np.ascontiguousarray(rec, np.uint8)
        rows, stride = rec.shape
        W = 2 * (clip // 2)
        f = np.empty((W, W), np.float32) if want_f32 else None
        u = np.empty((W, W), np.uint8) if want_u8 else None
        self.check(self.lib.roam_polar_to_cart_record_u8(self.h, _ptr(rec), rows, stride, payload_off, clip, _ptr(f), _ptr(u)))
        return f, u

    def pyr_down_u8(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self.check(self.lib.roam_pyr_down_u8(self.h, _ptr(img), w, h, _ptr(out)))
        return out

    def klt_track(self, prev_img, next_img, pts):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        K = pts.shape[0]
        nxt = np.zeros((K, 2), np.float32)
        st = np.zeros((K,), np.uint8)
        err = np.zeros((K,), np.float32)
        h, w = prev_img.shape
        assert next_img.shape == prev_img.shape
        if prev_img.dtype == np.uint8:
            a, b = np.ascontiguousarray(prev_img), np.ascontiguousarray(next_img, np.uint8)
            fn = self.lib.roam_klt_track_u8
        else:
            a, b = np.ascontiguousarray(prev_img, np.float32), np.ascontiguousarray(next_img, np.float32)
            fn = self.lib.roam_klt_track_f32
        self.check(fn(self.h, _ptr(a), _ptr(b), w, h, _ptr(pts), K, _ptr(nxt), _ptr(st), _ptr(err)))
        return nxt, st.reshape(-1, 1), err.reshape(-1, 1)

    def reject_outliers(self, prev, new, thr_px, node_limit=0, want_adj=False):
        prev = np.ascontiguousarray(prev, np.float32).reshape(-1, 2)
        new = np.ascontiguousarray(new, np.float32).reshape(-1, 2)
        K = prev.shape[0]
        mask = np.zeros(K, np.uint8)
        n_in, flags = C.c_int32(0), C.c_int32(0)
        adj = np.zeros((K, max(1, (K + 63) // 64)), np.uint64) if want_adj else None
        self.check(self.lib.roam_reject_outliers(self.h, _ptr(prev), _ptr(new), K, float(thr_px), int(node_limit),
                                                 _ptr(mask), C.byref(n_in), C.byref(flags), _ptr(adj)))
        return mask.astype(bool), n_in.value, flags.value, adj

    def time_reject_outliers(self, prev, new, thr_px, copies=4096, reps=3, node_limit=0):
        """(graph ms, clique ms, inliers, proven) per launch of `copies` replicas of one correspondence set"""
        p = np.ascontiguousarray(prev, np.float32).reshape(-1, 2)
        n = np.ascontiguousarray(new, np.float32).reshape(-1, 2)
        g, q, ni, pr = C.c_float(0), C.c_float(0), C.c_int32(0), C.c_int32(0)
        self.check(self.lib.roam_time_reject_outliers(self.h, _ptr(p), _ptr(n), p.shape[0], int(copies), float(thr_px), int(node_limit), int(reps),
                                                      C.byref(g), C.byref(q), C.byref(ni), C.byref(pr)))
        return float(g.value), float(q.value), ni.value, bool(pr.value)

    def kabsch2d(self, src, tgt):
        s = np.ascontiguousarray(src, np.float64).reshape(-1, 2)
        t = np.ascontiguousarray(tgt, np.float64).reshape(-1, 2)
        R = np.empty((2, 2), np.float64)
        h = np.empty((2, 1), np.float64)
        self.check(self.lib.roam_kabsch2d(self.h, _ptr(s), _ptr(t), s.shape[0], _ptr(R), _ptr(h)))
        return R, h

    def mds_solve(self, T_wj0, p_w, p_jt, T_init, sigma5, period=0.25, want_debug=False):
        T0 = np.ascontiguousarray(T_wj0, np.float64)
        Ti = np.ascontiguousarray(T_init, np.float64)
        pw = np.ascontiguousarray(p_w[:, :2], np.float64)
        pj = np.ascontiguousarray(p_jt[:, :2], np.float64)
        sg = np.ascontiguousarray(sigma5, np.float64)
        N = pw.shape[0]
        out = np.empty(6)
        nfev, info = C.c_int32(0), C.c_int32(0)
        x0 = np.empty(6) if want_debug else None
        r0 = np.empty(2 * N + 3) if want_debug else None
        self.check(self.lib.roam_mds_solve(self.h, _ptr(T0), _ptr(pw), _ptr(pj), N, _ptr(Ti), _ptr(sg), float(period),
                                           _ptr(out), C.byref(nfev), C.byref(info), _ptr(x0), _ptr(r0)))
        return out, nfev.value, info.value, x0, r0

    def mds_undistort(self, v, pts, period=0.25):
        v = np.ascontiguousarray(v, np.float64)
        p = np.ascontiguousarray(pts[:, :2], np.float64)
        N = p.shape[0]
        xy = np.empty((N, 2))
        dT = np.empty(N)
        self.check(self.lib.roam_mds_undistort(self.h, _ptr(v), _ptr(p), N, float(period), _ptr(xy), _ptr(dT)))
        return xy, dT

    def ssc(self, kp, num_ret, tol, cols, rows):
        kp = np.ascontiguousarray(kp, np.float64)
        B = kp.shape[0]
        sel = np.empty(max(B, 1), np.int32)
        n = C.c_int32(0)
        self.check(self.lib.roam_ssc(self.h, _ptr(kp), B, int(num_ret), float(tol), int(cols), int(rows), _ptr(sel), C.byref(n)))
        return sel[:n.value]

    def fmt_rotation(self, src_polar, tgt_polar, clip_px=1012, downsample=10):
        """FMT.getRotationUsingFMT -> (angle rad, scale, response)"""
        a = np.ascontiguousarray(src_polar, np.float32)
        b = np.ascontiguousarray(tgt_polar, np.float32)
        assert a.shape == b.shape, "Images need to have the same shape!"
        ang, sc, rs = C.c_double(0), C.c_double(0), C.c_double(0)
        self.check(self.lib.roam_fmt_rotation(self.h, _ptr(a), _ptr(b), a.shape[0], a.shape[1], int(clip_px), int(downsample),
                                              C.byref(ang), C.byref(sc), C.byref(rs)))
        return ang.value, sc.value, rs.value

    def doh_maxima(self, img, sigmas, threshold, cap=1 << 18):
        """-> (rcs (n,3) int32 [row, col, sigma_index] in C order, values (n,) f64)"""
        img = np.ascontiguousarray(img, np.float32)
        h, w = img.shape
        sig = np.ascontiguousarray(sigmas, np.float64)
        rcs = np.empty((cap, 3), np.int32)
        val = np.empty(cap, np.float64)
        n = C.c_int32(0)
        self.check(self.lib.roam_doh_maxima(self.h, _ptr(img), w, h, _ptr(sig), len(sig), float(threshold), _ptr(rcs),
                                            _ptr(val), cap, C.byref(n)))
        return rcs[:n.value], val[:n.value]


_default = {}
_lock = threading.Lock()


def default_context(device_id: int = None) -> Context:
    """Process-wide context per device (lazily created); device from ROAM_DEVICE / LOCAL_RANK."""
    if device_id is None:
        device_id = int(os.environ.get("ROAM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _lock:
        if device_id not in _default:
            _default[device_id] = Context(device_id)
        return _default[device_id]
