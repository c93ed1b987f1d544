"""Oxford record decode + polar->Cartesian warp with the reference's names
(reference parseData.py:9-53,100-135).  The warp runs on the MI355X (warp.hip)."""
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
    """(400, 3779) u8 Oxford record from a PNG (the reference uses cv2.imread(..., IMREAD_GRAYSCALE),
    parseData.py:178; Pillow decodes the same 8-bit greyscale PNG).  PNG inflate is host work (§8f-f2)."""
    from PIL import Image
    return np.array(Image.open(imgPath).convert("L"), dtype=np.uint8)


def prefetchRadarRecords(imgPaths, workers: int = 0, depth: int = 0):
    """the records of `imgPaths`, IN ORDER, decoded ahead of the consumer by a pool of host threads (8f-f2: the reference decodes
    frame k with cv2.imread inside its loop, parseData.py:160-226 / RawROAMSystem.py:162-165; a 1.5 MB Oxford PNG inflates in
    10-20 ms, one thread feeds 50-100 frames/s, the engine takes 700-1100 pairs/s of one sequence).  zlib inflate and Pillow's
    decoder release the GIL, so threads scale; at most `depth` decoded frames (1.5 MB each) wait for the consumer.
    workers = 0: min(16, cores / 2); depth = 0: 3 x workers.  A frame that fails to decode raises when ITS turn comes."""
    import os
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    paths = list(imgPaths)
    workers = workers if workers > 0 else max(1, min(16, (os.cpu_count() or 2) // 2))
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


def getPolarImageFromImgPaths(imgPathArr, index: int) -> np.ndarray:
    polar, _, _, _, _, _ = extractDataFromRadarImage(readRadarRecord(imgPathArr[index]))
    return polar
