"""Oxford record decode + polar->Cartesian warp with the reference's names
(reference parseData.py:9-53,100-135).  The warp runs on the MI355X (warp.hip)."""
import os
import time

import numpy as np

from . import _ffi

RANGE_RESOLUTION_M = 0.0432                                   # parseData.py:9
DOWNSAMPLE_FACTOR = 2                                         # parseData.py:10
RANGE_RESOLUTION_CART_M = RANGE_RESOLUTION_M * DOWNSAMPLE_FACTOR   # parseData.py:13
MAX_RANGE_CLIP_DEFAULT = 87.5                                 # parseData.py:14


def extractDataFromRadarImage(polarImgData: np.ndarray, maxRangeClipM: float = MAX_RANGE_CLIP_DEFAULT):
    """Split a (400, 3779) u8 record into (range_azimuth_data f32, azimuths, range_resolution,
    azimuth_resolution, valid, timestamps) exactly like parseData.py:17-53.  (The engine never
    materialises the f32 image: its kernels read the u8 payload directly.)"""
    encoder_size = 5600
    timestamps = polarImgData[:, :8].copy().view(np.int64)
    azimuths = (polarImgData[:, 8:10].copy().view(np.uint16) / float(encoder_size) * 2 * np.pi).astype(np.float32)
    valid = polarImgData[:, 10:11] == 255
    data = polarImgData[:, 11:].astype(np.float32) / 255.
    if maxRangeClipM > 0:
        data = data[:, :int(maxRangeClipM / RANGE_RESOLUTION_M)]
    return data, azimuths, RANGE_RESOLUTION_M, azimuths[1] - azimuths[0], valid, timestamps


def convertPolarImageToCartesian(imgPolar: np.ndarray, logPolarMode: bool = False,
                                 downsampleFactor: int = DOWNSAMPLE_FACTOR,
                                 changeGlobalRangeResolution: bool = False) -> np.ndarray:
    """(rows, cols) f32 polar -> (2R, 2R) f32 Cartesian, R = cols // 2 (parseData.py:100-135)."""
    if logPolarMode or downsampleFactor != 2:
        raise NotImplementedError("only the reference's live configuration (linear, downsampleFactor=2) is built")
    cart, _ = _ffi.default_context().polar_to_cart_f32(imgPolar, want_f32=True, want_u8=False)
    return cart


def getRadarImgPaths(dataPath: str, timestampPath: str):
    """list of '<dataPath>/<stamp>.png' for every line of radar.timestamps (parseData.py:208-226; like the
    reference, the string valid flag is truthy for every line)"""
    import os
    imgPathArr = []
    with open(timestampPath, "r") as f:
        for line in f.readlines():
            stamp, valid = line.strip().split(" ")
            if valid:
                imgPathArr.append(os.path.join(dataPath, stamp + ".png"))
    return imgPathArr


def readRadarRecord(imgPath: str) -> np.ndarray:
    """(400, 3779) u8 Oxford record from a PNG (the reference uses cv2.imread(..., IMREAD_GRAYSCALE), parseData.py:178).  The data
    set's format - 8-bit greyscale, non-interlaced - is decoded by the library (roam_png_decode_file: zlib inflate + un-filter, host
    code); any other PNG goes through Pillow's conversion to 8-bit grey, as cv2 would convert it.  PNG inflate is host work (8f-f2)."""
    import ctypes as C
    from . import _ffi
    if not os.path.exists(imgPath):
        raise FileNotFoundError(imgPath)
    lib = _ffi.load_library()
    rows, cols = C.c_int32(0), C.c_int32(0)
    rc = lib.roam_png_decode_file(os.fsencode(imgPath), None, 0, 0, C.byref(rows), C.byref(cols))     # header only: the size
    if rc == _ffi.ROAM_E_CAPACITY and rows.value > 0 and cols.value > 0:
        out = np.empty((rows.value, cols.value), np.uint8)
        rc = lib.roam_png_decode_file(os.fsencode(imgPath), out.ctypes.data_as(C.c_void_p), out.size, 0, None, None)
        if rc == _ffi.ROAM_OK:
            return out
    from PIL import Image
    return np.array(Image.open(imgPath).convert("L"), dtype=np.uint8)


def prefetchRadarRecords(imgPaths, workers: int = 0, depth: int = 0):
    """the records of `imgPaths`, IN ORDER, decoded ahead of the consumer by a pool of host threads (8f-f2: the reference decodes
    frame k with cv2.imread inside its loop, parseData.py:160-226 / RawROAMSystem.py:162-165; a 1.5 MB Oxford PNG inflates in
    10-20 ms, one thread feeds 50-100 frames/s, the engine takes 700-1100 pairs/s of one sequence).  zlib inflate and Pillow's
    decoder release the GIL, so threads scale; at most `depth` decoded frames (1.5 MB each) wait for the consumer.
    workers = 0: min(32, cores / 2); depth = 0: 3 x workers.  A frame that fails to decode raises when ITS turn comes."""
    import os
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    paths = list(imgPaths)
    workers = workers if workers > 0 else max(1, min(32, (os.cpu_count() or 2) // 2))
    depth = depth if depth > 0 else 3 * workers
    if workers == 1:
        for p in paths:
            yield readRadarRecord(p)
        return
    pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="roam-png")
    try:
        pending, nxt = deque(), 0
        while nxt < len(paths) or pending:
            while nxt < len(paths) and len(pending) < depth:
                pending.append(pool.submit(readRadarRecord, paths[nxt]))
                nxt += 1
            yield pending.popleft().result()
    finally:
        pool.shutdown(wait=True, cancel_futures=True)


class NativeRecordReader:
    """PNG files -> Oxford records, decoded by the library's pool of host THREADS (roam_png_pool_*: zlib inflate + PNG un-filter in C,
    8f-f2) straight into the slots of a ring - pinned memory (roam_host_alloc) when a Context is given, so that a record is uploaded
    from where it was decoded: no Pillow, no GIL, no copy between processes.  records() yields the frames IN ORDER as views of their
    slots; the view of frame j stays untouched until the consumer asks for frame j + hold + 1 (an asynchronous upload from the slot has
    `hold` frames to finish).  workers = 0: min(32, cores / 2); depth (slots, 1.5 MB each) = 0: hold + 2 x workers.  Context manager, or close()."""

    def __init__(self, workers: int = 0, depth: int = 0, rec_shape=(400, 3779), ctx=None, hold: int = 1):
        import ctypes as C
        from . import _ffi
        self._C, self._ffi = C, _ffi
        self.lib = _ffi.load_library()
        self.workers = workers if workers > 0 else max(1, min(32, (os.cpu_count() or 2) // 2))
        self.hold = max(1, int(hold))
        self.depth = depth if depth > 0 else self.hold + 2 * self.workers
        if self.depth <= self.hold:
            raise ValueError("depth must exceed hold")
        self.rec_shape = tuple(int(v) for v in rec_shape)
        self.rec_bytes = int(np.prod(self.rec_shape))
        self.ctx = ctx
        self.pinned = ctx is not None
        self.ring = ctx.host_alloc((self.depth,) + self.rec_shape) if ctx is not None else np.empty((self.depth,) + self.rec_shape, np.uint8)
        h = C.c_void_p()
        rc = self.lib.roam_png_pool_create(self.workers, C.byref(h))
        if rc != _ffi.ROAM_OK:
            raise _ffi.RoamError(rc, "roam_png_pool_create")
        self._pool = h
        self._ticket = 0
        self.wait_s = 0.0                                           # seconds the consumer has spent blocked on a frame that was not ready

    def records(self, imgPaths):
        C = self._C
        paths = [os.fsencode(p) for p in imgPaths]
        n = len(paths)
        base, nxt, want = self._ticket, 0, 0
        self._ticket += n
        rows, cols = C.c_int32(0), C.c_int32(0)
        try:
            while want < n:
                # frame i lives in slot i % depth: it may be submitted once frame i - depth is `hold` yields behind the consumer
                while nxt < n and nxt < want + self.depth - self.hold:
                    slot = self.ring[nxt % self.depth]
                    rc = self.lib.roam_png_pool_submit(self._pool, paths[nxt], slot.ctypes.data_as(C.c_void_p), self.rec_bytes, self.rec_shape[1], base + nxt)
                    if rc != self._ffi.ROAM_OK:
                        raise self._ffi.RoamError(rc, "roam_png_pool_submit")
                    nxt += 1
                t0 = time.perf_counter()
                rc = self.lib.roam_png_pool_wait(self._pool, base + want, C.byref(rows), C.byref(cols))
                self.wait_s += time.perf_counter() - t0
                want += 1
                if rc != self._ffi.ROAM_OK:
                    p = os.fsdecode(paths[want - 1])
                    if not os.path.exists(p):
                        raise FileNotFoundError(p)
                    raise RuntimeError(f"{p}: not an 8-bit greyscale PNG of at most {self.rec_shape} (status {rc}, {rows.value} x {cols.value})")
                if (rows.value, cols.value) == self.rec_shape:
                    yield self.ring[(want - 1) % self.depth]
                else:                                           # a smaller image: the rows lie rec_shape[1] bytes apart in the slot
                    yield self.ring[(want - 1) % self.depth][:rows.value, :cols.value]
        finally:
            for i in range(want, nxt):                          # nothing of this iteration is left in flight when it ends, however it ends
                self.lib.roam_png_pool_wait(self._pool, base + i, None, None)

    def close(self):
        if self._pool is not None:
            self.lib.roam_png_pool_destroy(self._pool)
            self._pool = None
            if self.ctx is not None:
                self.ctx.host_free(self.ring)
            self.ring = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _decode_worker(shm_name, rec_bytes, tasks, done):
    """body of one process of RecordDecodePool: decode the PNGs it is handed into their slots of the shared ring"""
    from multiprocessing import shared_memory
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        buf = np.ndarray((shm.size,), np.uint8, buffer=shm.buf)
        buf[::4096] |= np.uint8(0)                               # map the whole ring into this process now (24 k soft page faults per
        done.put((-1, -1, None, None, None))                     # worker otherwise spread over its first frames); then: ready
        while True:
            job = tasks.get()
            if job is None:
                break
            idx, path, slot = job
            try:
                rec = readRadarRecord(path)
                if rec.size > rec_bytes:
                    raise ValueError(f"{path}: {rec.shape} does not fit a {rec_bytes}-byte slot")
                buf[slot * rec_bytes:slot * rec_bytes + rec.size] = rec.ravel()
                done.put((idx, slot, rec.shape, None, None))
            except BaseException as ex:                          # noqa: BLE001 - reported at the frame's turn
                done.put((idx, slot, None, type(ex).__name__, str(ex)))
        del buf
    finally:
        shm.close()


class RecordDecodePool:
    """PNG inflate on a pool of host PROCESSES (8f-f2).  Threads stop scaling at ~1 300 frames/s on a 256-core host - what Pillow does
    around its decoder holds the GIL for ~0.8 ms per frame, and the thread that feeds the pinned ring queues for it too - which is enough
    for the single-sequence driver with motion distortion (700 pairs/s) and not without (1 200).  Processes decode into the slots of a
    ring in shared memory; the consumer gets the frames IN ORDER as views of those slots (valid until it asks for the next one).
    workers = 0: min(32, cores / 2); depth (slots, 1.5 MB each) = 0: 2 x workers.  Use as a context manager, or close()."""

    def __init__(self, workers: int = 0, depth: int = 0, rec_bytes: int = 400 * 3779):
        import multiprocessing as mp
        import os
        from multiprocessing import shared_memory
        self.workers = workers if workers > 0 else max(1, min(32, (os.cpu_count() or 2) // 2))
        self.depth = depth if depth > 0 else 2 * self.workers
        self.rec_bytes = int(rec_bytes)
        self._shm = shared_memory.SharedMemory(create=True, size=self.depth * self.rec_bytes)
        self._buf = np.ndarray((self._shm.size,), np.uint8, buffer=self._shm.buf)
        ctx = mp.get_context("spawn")                          # fresh interpreters: nothing of the parent's GPU state is inherited
        self._tasks, self._done = ctx.Queue(), ctx.Queue()
        self._outstanding = 0
        self._procs = [ctx.Process(target=_decode_worker, args=(self._shm.name, self.rec_bytes, self._tasks, self._done), daemon=True)
                       for _ in range(self.workers)]
        for p in self._procs:
            p.start()
        self._outstanding = self.workers                         # one "ready" message each: the ring is mapped, the imports are done
        try:
            self._drain()
        except BaseException:
            self.close()
            raise

    def records(self, imgPaths):
        paths = list(imgPaths)
        self._drain()                                           # (an earlier iteration that was abandoned half way)
        free = list(range(self.depth))
        ready, nxt, want = {}, 0, 0
        try:
            while want < len(paths):
                while nxt < len(paths) and free:
                    self._tasks.put((nxt, paths[nxt], free.pop()))
                    self._outstanding += 1
                    nxt += 1
                while want not in ready:
                    idx, slot, shape, exn, msg = self._get_done()
                    self._outstanding -= 1
                    ready[idx] = (slot, shape, exn, msg)
                slot, shape, exn, msg = ready.pop(want)
                want += 1
                if exn is not None:
                    raise (FileNotFoundError(msg) if exn == "FileNotFoundError" else RuntimeError(f"{exn}: {msg}"))
                n = int(np.prod(shape))
                yield self._buf[slot * self.rec_bytes:slot * self.rec_bytes + n].reshape(shape)
                free.append(slot)                               # the consumer is back: it has copied the frame
        finally:
            self._drain()                                       # nothing of this iteration is left in flight when it ends, however it ends

    def _get_done(self):
        import queue
        while True:
            try:
                return self._done.get(timeout=1.0)
            except queue.Empty:
                dead = [p.exitcode for p in self._procs if not p.is_alive()]
                if dead:                                        # (e.g. spawned from a __main__ that is not an importable file)
                    self._outstanding = 0
                    raise RuntimeError(f"RecordDecodePool: {len(dead)} of {self.workers} decode processes died (exit codes {dead[:4]})")

    def _drain(self):
        while self._outstanding > 0:
            self._get_done()
            self._outstanding -= 1

    def close(self):
        if self._shm is None:
            return
        for _ in self._procs:
            self._tasks.put(None)
        for p in self._procs:
            p.join(timeout=5)
            if p.is_alive():
                p.kill()                                        # (its own child, by handle)
        self._buf = None
        self._shm.close()
        self._shm.unlink()
        self._shm = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def getPolarImageFromImgPaths(imgPathArr, index: int) -> np.ndarray:
    polar, _, _, _, _, _ = extractDataFromRadarImage(readRadarRecord(imgPathArr[index]))
    return polar
