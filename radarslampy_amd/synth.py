"""Seeded synthetic Oxford-format radar sequences (host-side input generator).

A record is what parseData.extractDataFromRadarImage (reference parseData.py:17-53) decodes:
400 azimuth rows x (8 B int64 us timestamp | 2 B uint16 encoder | 1 B valid | 3768 B power).
The world is a set of static point reflectors (+ optional movers) seen from an SE(2)
trajectory; each reflector is rendered as a Gaussian blob in (azimuth, range) on top of
range-correlated speckle whose statistics follow the real `data/tiny` scans (mean ~11/255).
Conventions match the reference: Cartesian pixel = 1012 + metres/0.0864 with x right / y
down, azimuth row = atan2(y, x) * 400 / 2pi, range bin = r / 0.0432 m, pose T_wj maps
sensor-frame metres to world metres (RawROAMSystem.py:194-201), optional constant-velocity
intra-scan distortion with the time offsets of motionDistortion.py:107-153.
This is input data generation, not part of the measured path."""
import numpy as np

ROWS, NBINS, META = 400, 3768, 11
STRIDE = META + NBINS
RANGE_RES = 0.0432
M_PER_PX = 0.0864
CENTER = 1012.0
CLIP = 2025


def se2(x, y, th):
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, -s, x], [s, c, y], [0, 0, 1.0]])


class World:
    def __init__(self, seed: int, n_static: int = 320, n_movers: int = 0, extent_m: float = 85.0):
        rng = np.random.default_rng(seed)
        self.static = rng.uniform(-extent_m, extent_m, size=(n_static, 2)) + rng.uniform(-5, 5, size=2)
        self.amp = rng.uniform(70, 136, size=n_static + n_movers)
        self.movers = rng.uniform(-extent_m * 0.6, extent_m * 0.6, size=(n_movers, 2))
        self.mover_vel = rng.uniform(-6, 6, size=(n_movers, 2))       # m/s
        self.seed = seed


def trajectory(n_frames: int, seed: int, speed_m=(0.8, 1.6), yaw=(-0.03, 0.03), pose0=(0.0, 0.0, 0.0)):
    """poses (n,3) [x,y,th] world; per-frame deltas like full_seq_1 (mean ~1 m/frame)."""
    rng = np.random.default_rng(seed + 7919)
    T = se2(*pose0)
    poses = [np.array(pose0, float)]
    for _ in range(n_frames - 1):
        d = se2(rng.uniform(*speed_m), rng.uniform(-0.05, 0.05), rng.uniform(*yaw))
        T = T @ d
        poses.append(np.array([T[0, 2], T[1, 2], np.arctan2(T[1, 0], T[0, 0])]))
    return np.array(poses)


def _speckle(rng, rows, cols, mean=9.0):
    from scipy.ndimage import uniform_filter1d
    n = rng.exponential(mean, size=(rows, cols)).astype(np.float32)
    n = uniform_filter1d(n, size=5, axis=1, mode="nearest")          # range-correlated like real scans
    n = uniform_filter1d(n, size=2, axis=0, mode="wrap")
    return n


def render_record(world: World, pose, t_index: int = 0, velocity=None, seed: int = 0, timestamp_us: int = 1547131046353776,
                  scintillation: float = 0.0):
    """One 400 x 3779 u8 record of `world` seen from `pose`.  velocity=(vx,vy,vth) [m/s,rad/s]
    switches the intra-scan motion distortion on.  scintillation in (0, 1]: every reflector's amplitude is scaled by an
    independent factor in [1 - scintillation, 1] per frame (radar returns fluctuate from scan to scan; on the real `tiny`
    scans half of the tracked features fail the LK error gate every frame)."""
    rng = np.random.default_rng((world.seed * 1000003 + t_index * 7 + seed) & 0x7fffffff)
    img = _speckle(rng, ROWS, NBINS)
    fade = np.ones(len(world.amp))
    if scintillation > 0:
        fade = np.random.default_rng((world.seed * 7919 + t_index * 104729 + 13) & 0x7fffffff).uniform(1.0 - scintillation, 1.0, size=len(world.amp))
    Tinv = np.linalg.inv(se2(*pose))
    pts = world.static
    if len(world.movers):
        pts = np.vstack((pts, world.movers + world.mover_vel * (0.25 * t_index)))
    p = (Tinv @ np.column_stack((pts, np.ones(len(pts)))).T).T[:, :2]
    if velocity is not None:
        v = np.asarray(velocity, float)
        dT = 0.25 * np.arctan2(-p[:, 1], -p[:, 0]) / (2 * np.pi)
        out = np.empty_like(p)
        for i in range(len(p)):                                       # p_obs = SE2(v dT)^-1 p_ref
            out[i] = (np.linalg.inv(se2(*(v * dT[i]))) @ np.array([p[i, 0], p[i, 1], 1.0]))[:2]
        p = out
    r = np.hypot(p[:, 0], p[:, 1]) / RANGE_RES
    az = (np.arctan2(p[:, 1], p[:, 0]) % (2 * np.pi)) * ROWS / (2 * np.pi)
    rows_idx = np.arange(ROWS)[:, None]
    for ri, ai, A in zip(r, az, world.amp * fade):
        if ri < 20 or ri > NBINS - 20:
            continue
        sig_a = max(0.9, 2.2 * ROWS / (2 * np.pi * max(ri * 0.5, 1.0)) * 4.0)   # ~constant metric extent
        sig_r = 3.0
        a0, a1 = int(np.floor(ai - 4 * sig_a)), int(np.ceil(ai + 4 * sig_a))
        r0, r1 = max(0, int(ri - 12)), min(NBINS, int(ri + 13))
        aa = np.arange(a0, a1 + 1)
        ga = np.exp(-0.5 * ((aa - ai) / sig_a) ** 2)
        gr = np.exp(-0.5 * ((np.arange(r0, r1) - ri) / sig_r) ** 2)
        img[np.ix_(aa % ROWS, np.arange(r0, r1))] += A * ga[:, None] * gr[None, :]
    rec = np.zeros((ROWS, STRIDE), np.uint8)
    ts = (timestamp_us + t_index * 250000 + (np.arange(ROWS) * 250000) // ROWS).astype(np.int64)
    rec[:, :8] = ts.view(np.uint8).reshape(ROWS, 8)
    enc = (13 + 14 * np.arange(ROWS)).astype(np.uint16)
    rec[:, 8:10] = enc.view(np.uint8).reshape(ROWS, 2)
    rec[:, 10] = 255
    rec[:, META:] = np.clip(np.floor(img), 0, 255).astype(np.uint8)
    return rec


def reflector_pixels(world: World, pose, margin_px: float = 40.0):
    """Cartesian pixel coordinates [x,y] (f32) of the static reflectors visible from `pose`
    inside the 2024^2 image (used as the initial feature set when a test/bench does not run
    the blob detector)."""
    Tinv = np.linalg.inv(se2(*pose))
    p = (Tinv @ np.column_stack((world.static, np.ones(len(world.static)))).T).T[:, :2]
    px = p / M_PER_PX + CENTER
    keep = (px[:, 0] > margin_px) & (px[:, 0] < 2024 - margin_px) & (px[:, 1] > margin_px) & (px[:, 1] < 2024 - margin_px)
    keep &= np.hypot(p[:, 0], p[:, 1]) > 3.0
    return px[keep].astype(np.float32)


def make_sequence(seed: int, n_frames: int, n_static: int = 320, n_movers: int = 0, distortion: bool = False,
                  scintillation: float = 0.0):
    """-> (records list of (400,3779) u8, poses (n,3), initial features (K,2) f32)."""
    world = World(seed, n_static, n_movers)
    poses = trajectory(n_frames, seed)
    recs = []
    for t in range(n_frames):
        vel = None
        if distortion and t > 0:
            d = np.linalg.inv(se2(*poses[t - 1])) @ se2(*poses[t])
            vel = np.array([d[0, 2], d[1, 2], np.arctan2(d[1, 0], d[0, 0])]) / 0.25
        recs.append(render_record(world, poses[t], t, vel, scintillation=scintillation))
    return recs, poses, reflector_pixels(world, poses[0])


# ---------------------------------------------------------------------------------------------------- long sequences
class StreamWorld:
    """Unbounded reflector world for long trajectories (BASELINE configs 3 / 4: the 8 866-frame, 9 km full_seq_1 path):
    reflectors are generated per TILE x TILE metre tile from a hash of (seed, tile), so every pose sees the same ~density *
    170^2 reflectors around it however far the path goes, and two visits of one place see the same reflectors."""
    TILE = 50.0

    def __init__(self, seed: int, per_tile: int = 40, mover_fraction: float = 0.0):
        self.seed, self.per_tile, self.mover_fraction = seed, per_tile, mover_fraction
        self._cache = {}

    def tile(self, ix: int, iy: int):
        key = (ix, iy)
        if key not in self._cache:
            rng = np.random.default_rng([self.seed & 0x7fffffff, ix & 0xffffffff, iy & 0xffffffff])
            xy = (np.array([ix, iy]) + rng.random((self.per_tile, 2))) * self.TILE
            amp = rng.uniform(70, 136, size=self.per_tile)
            vel = rng.uniform(-6, 6, size=(self.per_tile, 2)) * (rng.random((self.per_tile, 1)) < self.mover_fraction)
            if len(self._cache) > 4096:
                self._cache.clear()
            self._cache[key] = (xy, amp, vel)
        return self._cache[key]

    def around(self, x: float, y: float, radius_m: float = 95.0):
        """(positions (n,2) at t = 0, amplitudes (n,), velocities (n,2) m/s) of the tiles within radius of (x, y)"""
        i0, i1 = int(np.floor((x - radius_m) / self.TILE)), int(np.floor((x + radius_m) / self.TILE))
        j0, j1 = int(np.floor((y - radius_m) / self.TILE)), int(np.floor((y + radius_m) / self.TILE))
        parts = [self.tile(i, j) for i in range(i0, i1 + 1) for j in range(j0, j1 + 1)]
        return np.vstack([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), np.vstack([p[2] for p in parts])


class _LocalWorld:
    """adapter: the reflectors around one pose in the shape render_record expects"""

    def __init__(self, seed, static, amp):
        self.seed, self.static, self.amp = seed, static, amp
        self.movers, self.mover_vel = np.zeros((0, 2)), np.zeros((0, 2))


def render_stream_record(world: StreamWorld, pose, t_index: int, velocity=None, scintillation: float = 0.0):
    """one record of a StreamWorld seen from `pose` at frame t_index (movers advance 0.25 s per frame)"""
    xy, amp, vel = world.around(pose[0], pose[1])
    return render_record(_LocalWorld(world.seed, xy + vel * (0.25 * t_index), amp), pose, t_index, velocity, scintillation=scintillation)


def poses_from_deltas(deltas, pose0=(0.0, 0.0, 0.0)):
    """ground-truth style body-frame motions (n, 3) [dx, dy, dth] -> (n + 1, 3) poses, pose0 first"""
    T = se2(*pose0)
    out = [np.array(pose0, float)]
    for dx, dy, dth in np.asarray(deltas, float):
        T = T @ se2(dx, dy, dth)
        out.append(np.array([T[0, 2], T[1, 2], np.arctan2(T[1, 0], T[0, 0])]))
    return np.array(out)


def _render_job(job):
    seed, per_tile, mover_fraction, pose, t, vel, scint = job
    return render_stream_record(StreamWorld(seed, per_tile, mover_fraction), pose, t, vel, scint)


def stream_jobs(world: StreamWorld, poses, distortion: bool = False, scintillation: float = 0.0):
    """picklable render jobs for frames 0..len(poses)-1 (use with multiprocessing.Pool.imap(_render_job, jobs))"""
    jobs = []
    for t, pose in enumerate(poses):
        vel = None
        if distortion and t > 0:
            d = np.linalg.inv(se2(*poses[t - 1])) @ se2(*pose)
            vel = np.array([d[0, 2], d[1, 2], np.arctan2(d[1, 0], d[0, 0])]) / 0.25
        jobs.append((world.seed, world.per_tile, world.mover_fraction, np.asarray(pose, float), t, vel, scintillation))
    return jobs


def png_bytes_gray8(img: np.ndarray, filter_type: int = 1, level: int = 1, rle: bool = True, idat_bytes: int = 8192) -> bytes:
    """an 8-bit greyscale PNG of `img` the way the Oxford Radar RobotCar files are written (checked on the reference's data/tiny scans: Sub
    filter on every scanline, zlib stream header 78 01, IDAT chunks of 8 192 bytes - OpenCV's cv2.imwrite defaults: compression level 1,
    Z_RLE).  filter_type 0 (None) / 1 (Sub) / 2 (Up) for every line; used by bench.py --png and the tests, no part of the product path."""
    import struct
    import zlib
    a = np.ascontiguousarray(img, np.uint8)
    h, w = a.shape
    if filter_type == 0:
        body = a
    elif filter_type == 1:
        body = a.copy()
        body[:, 1:] -= a[:, :-1]
    elif filter_type == 2:
        body = a.copy()
        body[1:] -= a[:-1]
    else:
        raise ValueError("filter_type 0, 1 or 2")
    raw = np.empty((h, w + 1), np.uint8)
    raw[:, 0] = filter_type
    raw[:, 1:] = body
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, zlib.Z_RLE if rle else zlib.Z_DEFAULT_STRATEGY)
    z = co.compress(raw.tobytes()) + co.flush()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0))
    for i in range(0, len(z), idat_bytes):
        out += chunk(b"IDAT", z[i:i + idat_bytes])
    return out + chunk(b"IEND", b"")
